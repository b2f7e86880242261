// Small bandwidth-bound kernels around the generator's conv stack.
#include "common.h"

namespace dissc {

// x[b, c, t] = [ dict[code[b,t]] (E) | f0[b,t] (1) | spkr_emb[spkr[b]] (E) ]
// Replaces nn.Embedding x2 + _upsample + torch.cat, reference sr/models.py:189,207-215.
__global__ void embed_concat_kernel(const int64_t* __restrict__ code, const float* __restrict__ f0,
                                    const int64_t* __restrict__ spkr,
                                    const float* __restrict__ dict_w,
                                    const float* __restrict__ spkr_w,
                                    const int32_t* __restrict__ lengths, int T, int E, int has_f0,
                                    int has_spkr, int n_codes, int n_spk, float* __restrict__ x,
                                    int ldx, int C, const ZeroSpans zs) {
  const int b = blockIdx.z;
  const int c = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  // the forward's first kernel also zeroes the guard floats in front of the chains' scratch buffers (three 32-byte
  // hipMemsetAsync launches before: 3 x 9 us of a 2.1 ms single-utterance forward)
  if (blockIdx.x == 0 && c == 0 && b == 0 && threadIdx.x < ZERO_TAIL)
    for (int j = 0; j < zs.n; ++j) zs.p[j][threadIdx.x] = 0.f;
  const int len = lengths ? lengths[b] : T;
  if (t >= len) return;
  float v;
  if (c < E) {
    long long id = code[(size_t)b * T + t];
    id = id < 0 ? 0 : (id >= n_codes ? n_codes - 1 : id);  // out-of-range ids are clamped, never read OOB
    v = dict_w[(size_t)id * E + c];
  } else if (has_f0 && c == E) {
    v = f0[(size_t)b * T + t];
  } else {
    long long id = spkr[b];
    id = id < 0 ? 0 : (id >= n_spk ? n_spk - 1 : id);
    v = spkr_w[(size_t)id * E + (c - E - (has_f0 ? 1 : 0))];
  }
  x[((size_t)b * C + c) * ldx + t] = v;
}

void launch_embed_concat(const int64_t* code, const float* f0, const int64_t* spkr,
                         const float* dict_w, const float* spkr_w, const int32_t* lengths, int B,
                         int T, int E, int has_f0, int has_spkr, int n_codes, int n_spk, float* x,
                         int ldx, const ZeroSpans& zs, hipStream_t stream) {
  const int C = E + (has_f0 ? 1 : 0) + (has_spkr ? E : 0);
  dim3 grid((T + 127) / 128, C, B);
  hipLaunchKernelGGL(embed_concat_kernel, grid, dim3(128), 0, stream, code, f0, spkr, dict_w,
                     spkr_w, lengths, T, E, has_f0, has_spkr, n_codes, n_spk, x, ldx, C, zs);
}

// wav[b, t] = tanh(bias + sum_{ci,j} w[ci,j] * lrelu(x[b,ci,t+j-KS/2], slope)), 0 beyond length.
// Replaces F.leaky_relu -> conv_post -> tanh, reference sr/models.py:110-112.
constexpr int POST_TILE = 512;
__global__ void __launch_bounds__(256) conv_post_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias,
                                                        const int32_t* __restrict__ lengths,
                                                        int len_mul, int C, int KS, int L, int ldx,
                                                        long long x_bstride, float slope,
                                                        float* __restrict__ wav, int ldw) {
  extern __shared__ float xs[];  // [C][POST_TILE + KS - 1]
  const int b = blockIdx.y;
  const int len = lengths ? lengths[b] * len_mul : L;
  const int t0 = blockIdx.x * POST_TILE;
  float* wb = wav + (size_t)b * ldw;
  if (t0 >= len) {
    for (int i = threadIdx.x; i < POST_TILE; i += 256)
      if (t0 + i < L) wb[t0 + i] = 0.f;
    return;
  }
  const int halo = KS / 2;
  const int W = POST_TILE + KS - 1;
  const float* xb = x + (size_t)b * x_bstride;
  for (int e = threadIdx.x; e < C * W; e += 256) {
    const int ci = e / W, u = e - ci * W;
    const int t = t0 - halo + u;
    float v = 0.f;
    if (t >= 0 && t < len) {
      v = xb[(size_t)ci * ldx + t];
      v = v > 0.f ? v : v * slope;
    }
    xs[e] = v;
  }
  __syncthreads();
  const float bz = bias[0];
#pragma unroll
  for (int i = 0; i < POST_TILE / 256; ++i) {
    const int u = threadIdx.x + i * 256;
    const int t = t0 + u;
    if (t >= L) continue;
    float s = bz;
    for (int ci = 0; ci < C; ++ci)
      for (int j = 0; j < KS; ++j) s = fmaf(w[ci * KS + j], xs[ci * W + u + j], s);
    wb[t] = t < len ? tanhf(s) : 0.f;
  }
}

// The same tail for KS == 7, HBM-bound form: one thread = 4 consecutive samples, the 12-sample
// window of every channel comes straight from global memory as three aligned 16-byte loads (neighbouring
// threads share them through L1/L2), no LDS, no barrier.  Accumulation order as above (channels outer,
// taps inner) -> identical bits.
__global__ void __launch_bounds__(256) conv_post7_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias,
                                                         const int32_t* __restrict__ lengths, int len_mul, int C,
                                                         int L, int ldx, long long x_bstride, float slope,
                                                         float* __restrict__ wav, int ldw) {
  const int b = blockIdx.y;
  const int t = 4 * (blockIdx.x * 256 + threadIdx.x);
  if (t >= L) return;
  const int len = lengths ? lengths[b] * len_mul : L;
  float* wb = wav + (size_t)b * ldw + t;
  if (t >= len) {
    for (int e = 0; e < 4 && t + e < L; ++e) wb[e] = 0.f;
    return;
  }
  const float* xb = x + (size_t)b * x_bstride;
  const float bz = bias[0];
  float s[4] = {bz, bz, bz, bz};
  for (int ci = 0; ci < C; ++ci) {
    const float* row = xb + (size_t)ci * ldx;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4 q0 = t >= 4 ? *reinterpret_cast<const f32x4*>(row + t - 4) : zero;
    const f32x4 q1 = *reinterpret_cast<const f32x4*>(row + t);
    const f32x4 q2 = t + 8 <= ldx ? *reinterpret_cast<const f32x4*>(row + t + 4) : zero;
    float v[12];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = q0[e]; v[4 + e] = q1[e]; v[8 + e] = q2[e]; }
#pragma unroll
    for (int e = 0; e < 12; ++e) {
      const int pos = t - 4 + e;
      const float u = v[e] > 0.f ? v[e] : v[e] * slope;
      v[e] = (pos >= 0 && pos < len) ? u : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const float wj = w[ci * 7 + j];
#pragma unroll
      for (int e = 0; e < 4; ++e) s[e] = fmaf(wj, v[e + j + 1], s[e]);
    }
  }
  if (t + 4 <= L && ((ldw & 3) == 0)) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (t + e < len) ? tanhf(s[e]) : 0.f;
    *reinterpret_cast<f32x4*>(wb) = o;
  } else {
    for (int e = 0; e < 4 && t + e < L; ++e) wb[e] = (t + e < len) ? tanhf(s[e]) : 0.f;
  }
}

void launch_conv_post(const float* x, const float* w, const float* bias, const int32_t* lengths,
                      int len_mul, int B, int C, int KS, int L, int ldx, long long x_bstride,
                      float slope, float* wav, int ldw, hipStream_t stream) {
  if (KS == 7 && (ldx & 3) == 0 && (x_bstride & 3) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)wav & 15) == 0) {
    dim3 grid((L + 1023) / 1024, B);
    hipLaunchKernelGGL(conv_post7_kernel, grid, dim3(256), 0, stream, x, w, bias, lengths, len_mul, C, L, ldx,
                       x_bstride, slope, wav, ldw);
    return;
  }
  dim3 grid((L + POST_TILE - 1) / POST_TILE, B);
  const size_t lds = (size_t)C * (POST_TILE + KS - 1) * sizeof(float);
  hipLaunchKernelGGL(conv_post_kernel, grid, dim3(256), lds, stream, x, w, bias, lengths, len_mul,
                     C, KS, L, ldx, x_bstride, slope, wav, ldw);
}

// (y*32768) -> int16 with C truncation + two's-complement wrap -> float -> / max|.|
// Replaces reference sr/inference.py:73-75 and librosa.util.normalize at :206/:250.
__global__ void __launch_bounds__(1024) wav_postprocess_kernel(float* __restrict__ wav,
                                                               const int32_t* __restrict__ n_samples,
                                                               int ld) {
  __shared__ float red[16];
  const int b = blockIdx.x;
  const int n = n_samples[b];
  float* w = wav + (size_t)b * ld;
  float peak = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) {
    int v = (int)truncf(w[i] * 32768.0f);
    v = ((v + 32768) & 65535) - 32768;
    const float f = (float)v;
    w[i] = f;
    peak = fmaxf(peak, fabsf(f));
  }
  for (int off = 32; off > 0; off >>= 1) peak = fmaxf(peak, __shfl_xor(peak, off));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = peak;
  __syncthreads();
  peak = red[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) peak = fmaxf(peak, red[i]);
  if (peak < 1.17549435e-38f) return;  // librosa: norms below tiny are left as is
  for (int i = threadIdx.x; i < n; i += 1024) w[i] = __fdiv_rn(w[i], peak);
}

void launch_wav_postprocess(float* wav, const int32_t* n_samples, int B, int ld,
                            hipStream_t stream) {
  hipLaunchKernelGGL(wav_postprocess_kernel, dim3(B), dim3(1024), 0, stream, wav, n_samples, ld);
}

}  // namespace dissc

// ---- diagnostics: sustained fp32 MFMA rate of this GPU at its real clocks -------------
namespace dissc {
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) mfma_peak32_kernel(float* out, int iters) {
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7] + acc[i][15];
  if (s == 12345.678f) out[0] = s;
}

__global__ void __launch_bounds__(256) mfma_peak_kernel(float* out, int iters) {
  f32x4 acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f + 1.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678f) out[0] = s;
}
}  // namespace dissc

namespace dissc {
__global__ void erf_check_kernel(const float* __restrict__ x, float* __restrict__ y, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = erf_1ulp(x[i]);
}
}  // namespace dissc

// Diagnostics: y[i] = the device erf used inside the GELU epilogues (device pointers)
extern "C" int dissc_erf_check(const float* x, float* y, int n, void* stream) {
  if (!x || !y || n < 0) {
    dissc::set_error("dissc_erf_check: bad argument");
    return DISSC_EINVAL;
  }
  if (n == 0) return DISSC_OK;
  hipLaunchKernelGGL(dissc::erf_check_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, y, n);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

extern "C" int dissc_mfma_peak(int iters, float* tflops) {
  using namespace dissc;
  if (!tflops || iters == 0) return DISSC_EINVAL;
  float* d = nullptr;
  DISSC_HIP_CHECK(hipMalloc((void**)&d, 16));
  hipEvent_t e0, e1;
  DISSC_HIP_CHECK(hipEventCreate(&e0));
  DISSC_HIP_CHECK(hipEventCreate(&e1));
  const bool use32 = iters < 0;  // negative iters: the 32x32x2 form (same FLOPs per iteration)
  if (use32) iters = -iters;
  const int blocks = 256 * 4;  // 4 blocks x 4 waves per CU
  if (use32) hipLaunchKernelGGL(mfma_peak32_kernel, dim3(blocks), dim3(256), 0, nullptr, d, 100);
  else hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, nullptr, d, 100);
  DISSC_HIP_CHECK(hipEventRecord(e0, nullptr));
  if (use32) hipLaunchKernelGGL(mfma_peak32_kernel, dim3(blocks), dim3(256), 0, nullptr, d, iters);
  else hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, nullptr, d, iters);
  DISSC_HIP_CHECK(hipEventRecord(e1, nullptr));
  DISSC_HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  DISSC_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double flops = 2.0 * 16 * 16 * 4 * 8.0 * iters * 4.0 * blocks;  // per MFMA x 8 x iters x waves
  *tflops = (float)(flops / (ms * 1e-3) / 1e12);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(d);
  return DISSC_OK;
}

// ---- diagnostics: do the fp32 VALU pipe and the MFMA pipe overlap on this part? ----------------
namespace dissc {
__global__ void __launch_bounds__(256) valu_peak_kernel(float* out, int iters) {
  float a[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 1e-3f + i;
  const float m = 1.0001f, c = 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = fmaf(a[i], m, c);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i];
  if (s == 12345.678f) out[0] = s;
}
}  // namespace dissc

// ms[0] = MFMA kernel alone, ms[1] = VALU kernel alone, ms[2] = both concurrently (two streams)
extern "C" int dissc_pipe_overlap(int mfma_iters, int valu_iters, float* ms) {
  using namespace dissc;
  if (!ms || mfma_iters <= 0 || valu_iters <= 0) return DISSC_EINVAL;
  float* d = nullptr;
  DISSC_HIP_CHECK(hipMalloc((void**)&d, 16));
  hipStream_t s1, s2;
  DISSC_HIP_CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  DISSC_HIP_CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1, e2;
  DISSC_HIP_CHECK(hipEventCreate(&e0));
  DISSC_HIP_CHECK(hipEventCreate(&e1));
  DISSC_HIP_CHECK(hipEventCreate(&e2));
  const int blocks = 256 * 2;  // 2 blocks x 4 waves per CU per kernel: both kernels fit side by side
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(mfma_peak32_kernel, dim3(blocks), dim3(256), 0, s1, d, 10);
    hipLaunchKernelGGL(valu_peak_kernel, dim3(blocks), dim3(256), 0, s2, d, 10);
    DISSC_HIP_CHECK(hipDeviceSynchronize());
    DISSC_HIP_CHECK(hipEventRecord(e0, s1));
    DISSC_HIP_CHECK(hipStreamWaitEvent(s2, e0, 0));
    if (mode == 0 || mode == 2) hipLaunchKernelGGL(mfma_peak32_kernel, dim3(blocks), dim3(256), 0, s1, d, mfma_iters);
    if (mode == 1 || mode == 2) hipLaunchKernelGGL(valu_peak_kernel, dim3(blocks), dim3(256), 0, s2, d, valu_iters);
    DISSC_HIP_CHECK(hipEventRecord(e2, s2));
    DISSC_HIP_CHECK(hipStreamWaitEvent(s1, e2, 0));
    DISSC_HIP_CHECK(hipEventRecord(e1, s1));
    DISSC_HIP_CHECK(hipEventSynchronize(e1));
    DISSC_HIP_CHECK(hipEventElapsedTime(&ms[mode], e0, e1));
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2);
  (void)hipStreamDestroy(s1); (void)hipStreamDestroy(s2);
  (void)hipFree(d);
  return DISSC_OK;
}
