// Micro-benchmark, companion of mfma_mix.hip: would an 8-WAVE workgroup (one wave per evaluation point of an 8-point
// Toom-Cook form such as F(6,3); 2 waves per SIMD, up to 256 registers each) with a LARGER wave tile sustain more of the
// fp32 matrix pipe than conv_wino_kernel's 12 waves x (64 x 64)?  Per "tap" (4 k-steps = 8 channels) a wave with an
// (MI x NI) tile of 32 x 32 blocks issues 4 MI NI MFMAs and needs MI global 16-byte A-fragment loads (one tap ahead) and
// 4 NI ds_read_b32 B-fragment values:  MI = 4, NI = 2: 32 MFMAs, 4 A, 8 B;  MI = 2, NI = 4: 32 MFMAs, 2 A, 16 B.
// Build + run: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/mfma_mix8 mfma_mix8.hip && /tmp/mfma_mix8
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NWV, int MI, int NI, int MODE>
__global__ void __launch_bounds__(64 * NWV, NWV / 4) mix8_kernel(const float* __restrict__ wslab, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 16 * 1024; i += 64 * NWV) lds[i] = 1e-3f * (i & 255);
  __syncthreads();
  f32x16 acc[MI][NI];
  for (int a = 0; a < MI; ++a)
    for (int b = 0; b < NI; ++b)
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  // every workgroup streams the same NWV slabs (L2 hits, like the kernel's weights)
  const f32x4* wp = reinterpret_cast<const f32x4*>(wslab) + (size_t)wave * 64 * 64 * MI + lane;
  const float* vt = lds + wave * 8 * 112 + (lane & 31) + (lane >> 5) * 112;
  f32x4 av[MI], avn[MI];
  for (int m = 0; m < MI; ++m) av[m] = f32x4{1.f, 2.f, 3.f, 4.f};
  float bk[4][NI];
  for (int s = 0; s < 4; ++s)
    for (int n = 0; n < NI; ++n) bk[s][n] = 1.f + s + n;
  for (int it = 0; it < iters; ++it) {
    const int j = it & 3;
    if (MODE & 2) {
#pragma unroll
      for (int m = 0; m < MI; ++m) avn[m] = wp[(size_t)((it & 31) + 32 * m) * 64];
    } else {
#pragma unroll
      for (int m = 0; m < MI; ++m) avn[m] = av[m];
    }
    float bn[4][NI];
    if (MODE & 1) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int n = 0; n < NI; ++n) bn[s][n] = vt[s * 2 * 112 + j + (n & 1) * 32 + (n >> 1) * 7];
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int n = 0; n < NI; ++n) bn[s][n] = bk[s][n];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int m = 0; m < MI; ++m)
#pragma unroll
        for (int n = 0; n < NI; ++n)
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][s], bk[s][n], acc[m][n], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < MI; ++m) av[m] = avn[m];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int n = 0; n < NI; ++n) bk[s][n] = bn[s][n];
  }
  float sum = 0.f;
  for (int a = 0; a < MI; ++a)
    for (int b = 0; b < NI; ++b)
      for (int e = 0; e < 16; ++e) sum += acc[a][b][e];
  if (sum == 12345.678f) out[0] = sum;
}

template <int NWV, int MI, int NI, int MODE>
static void run(const float* w, float* out, int iters, const char* what) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&mix8_kernel<NWV, MI, NI, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize,
                      160 * 1024);
  const int blocks = 256 * 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((mix8_kernel<NWV, MI, NI, MODE>), dim3(blocks), dim3(64 * NWV), 100 * 1024, 0, w, out, 64);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((mix8_kernel<NWV, MI, NI, MODE>), dim3(blocks), dim3(64 * NWV), 100 * 1024, 0, w, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * 32 * 32 * 2 * 4.0 * MI * NI * iters * NWV * blocks;
  printf("%2d waves, tile %d x %d blocks, mode %d  %-40s %8.3f ms  %6.1f TFLOP/s\n", NWV, MI, NI, MODE, what, ms,
         flops / (ms * 1e-3) / 1e12);
}

int main() {
  float *w = nullptr, *out = nullptr;
  const size_t nw = (size_t)16 * 64 * 64 * 4 * 4;
  hipMalloc((void**)&w, nw * sizeof(float));
  hipMalloc((void**)&out, 16);
  hipMemset(w, 0, nw * sizeof(float));
  const int iters = 2048;
  for (int rep = 0; rep < 2; ++rep) {
    run<12, 2, 2, 0>(w, out, iters, "MFMA only (today's shape)");
    run<12, 2, 2, 3>(w, out, iters, "A global + B b32 (today's tap loop)");
    run<8, 4, 2, 0>(w, out, iters, "MFMA only");
    run<8, 4, 2, 1>(w, out, iters, "+ B b32");
    run<8, 4, 2, 2>(w, out, iters, "+ A global");
    run<8, 4, 2, 3>(w, out, iters, "A global + B b32");
    run<8, 2, 4, 0>(w, out, iters, "MFMA only");
    run<8, 2, 4, 3>(w, out, iters, "A global + B b32");
    run<8, 2, 2, 3>(w, out, iters, "A global + B b32 (2 per SIMD, small tile)");
    run<16, 2, 2, 3>(w, out, iters, "A global + B b32 (4 per SIMD)");
  }
  return 0;
}
