// dissc_hubert_*: HuBERT-base unit encoder (CNN feature extractor -> projection -> positional
// conv -> N post-LN transformer layers -> k-means assignment) on one MI355X.
//
// Replaces what data/encode.py:21-22,32 reaches through textless' SpeechEncoder: fairseq
// HubertModel.extract_features(output_layer=6) and the k-means quantiser's predict()
// (third-party code that is absent from the reference tree; algorithm restated in
// oracle/hubert_ref.py, pinned to HF HubertModel + sklearn KMeans).
//
// Everything stays channels-first [B][C][ld] fp32 with time contiguous, so every dense layer
// (strided feature convs, 1x1 projections, QKV/out/FFN linears, the grouped k=128 positional
// conv, the centroid dot products) is the same fp32-MFMA implicit GEMM as the vocoder
// (conv_mfma_kernel) with GELU / residual fused in its epilogue.  Attention is exact:
// S = Q^T K (batched MFMA GEMM) -> row softmax -> O = V P^T (batched MFMA GEMM).
// Ragged batches are per-utterance exact: GroupNorm statistics over valid frames only, zero
// padding of the positional conv at each utterance's end, keys masked to T_b.
#include <string.h>

#include <map>
#include <string>

#include "common.h"

namespace dissc {

constexpr int NCONV = 7;
__constant__ int c_k[NCONV] = {10, 3, 3, 3, 3, 2, 2};
__constant__ int c_s[NCONV] = {5, 2, 2, 2, 2, 2, 2};

// lens[l][b] = frames after conv layer l (0 when the utterance is too short)
__global__ void hubert_lengths_kernel(const int32_t* __restrict__ n_samples, int B, int n_default,
                                      int32_t* __restrict__ lens) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int n = n_samples ? n_samples[b] : n_default;
  n = n < 0 ? 0 : (n > n_default ? n_default : n);  // never read past the row (host validates and raises)
  for (int l = 0; l < NCONV; ++l) {
    n = n >= c_k[l] ? (n - c_k[l]) / c_s[l] + 1 : 0;
    lens[l * B + b] = n;
  }
}

// ---- conv0 (1 -> 512, k10, s5, no bias) + GroupNorm(512 groups) + GELU -----------------------
constexpr int C0_TILE = 1024;  // outputs per block: 256 threads x 4
constexpr int C0_CH = 512;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// pass 1: GroupNorm statistics WITHOUT evaluating the 512 channels.  conv0 has one input channel, so
// channel c's output is x_c[t] = sum_j w[c][j] * s[5t + j] and its moments are linear / quadratic forms of
// the waveform's own lag moments:
//     sum_t x_c     = sum_j   w_cj       * S_j ,   S_j  = sum_t s[5t+j]                (10 numbers)
//     sum_t x_c^2   = sum_jk  w_cj w_ck  * R_jk,   R_jk = sum_t s[5t+j] * s[5t+k]      (55 numbers, j <= k)
// One light pass over the 1-channel waveform (65 fp64 partial sums per 1024-frame tile, fixed reduction
// order -> deterministic), then 512 tiny quadratic forms per utterance.  All in double: more accurate than
// summing fp32 outputs, and 512x less arithmetic than a pass that evaluates the channels.
constexpr int C0_NM = 65;  // 10 + 55

__global__ void __launch_bounds__(256) conv0_moments_kernel(const float* __restrict__ wav, int ldw,
                                                            const int32_t* __restrict__ len0, int ntile,
                                                            double* __restrict__ part) {
  __shared__ double red[4][C0_NM];
  const int b = blockIdx.y, tile = blockIdx.x;
  const int T0 = len0[b];
  const int t = tile * C0_TILE + threadIdx.x * 4;
  const float* wb = wav + (size_t)b * ldw;
  double acc[C0_NM];
#pragma unroll
  for (int i = 0; i < C0_NM; ++i) acc[i] = 0.0;
  float s[25];
  const bool any = t < T0;
#pragma unroll
  for (int i = 0; i < 25; ++i) s[i] = (any && (5 * t + i) < 5 * (T0 - 1) + 10) ? wb[5 * t + i] : 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (t + e < T0) {
      int idx = 10;
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        const double sj = (double)s[5 * e + j];
        acc[j] += sj;
#pragma unroll
        for (int k = j; k < 10; ++k) { acc[idx] = fma(sj, (double)s[5 * e + k], acc[idx]); ++idx; }
      }
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < C0_NM; ++i) {
    double v = acc[i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if (lane == 0) red[wv][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < C0_NM)
    part[((size_t)b * ntile + tile) * C0_NM + threadIdx.x] =
        ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// tile partials (fixed order) -> mean / rstd per (b, c)
__global__ void __launch_bounds__(128) conv0_finalize_kernel(const double* __restrict__ part,
                                                             const float* __restrict__ w,
                                                             const int32_t* __restrict__ len0, int ntile, float eps,
                                                             float* __restrict__ stats) {
  __shared__ double M[C0_NM];
  const int b = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int T0 = len0[b];
  const int nt = (T0 + C0_TILE - 1) / C0_TILE;
  if (threadIdx.x < C0_NM) {
    double v = 0.0;
    for (int i = 0; i < nt; ++i) v += part[((size_t)b * ntile + i) * C0_NM + threadIdx.x];
    M[threadIdx.x] = v;
  }
  __syncthreads();
  if (c >= C0_CH) return;
  double wj[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) wj[j] = (double)w[c * 10 + j];
  double s = 0.0, q = 0.0;
  int idx = 10;
#pragma unroll
  for (int j = 0; j < 10; ++j) {
    s += wj[j] * M[j];
#pragma unroll
    for (int k = j; k < 10; ++k) {
      const double term = wj[j] * wj[k] * M[idx++];
      q += (k == j) ? term : 2.0 * term;
    }
  }
  const double n = T0 > 0 ? (double)T0 : 1.0;
  const double mean = s / n;
  double var = q / n - mean * mean;
  if (var < 0) var = 0;
  stats[((size_t)b * C0_CH + c) * 2] = (float)mean;
  stats[((size_t)b * C0_CH + c) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// pass 2: recompute conv0, normalise, affine, GELU, store [B][512][ld0]
__global__ void __launch_bounds__(256) conv0_apply_kernel(const float* __restrict__ wav, int ldw,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          const float* __restrict__ stats,
                                                          const int32_t* __restrict__ len0,
                                                          float* __restrict__ out, int ld0) {
  const int b = blockIdx.y;
  const int T0 = len0[b];
  const int t = blockIdx.x * C0_TILE + threadIdx.x * 4;
  if (t >= T0) return;
  const float* wb = wav + (size_t)b * ldw;
  float s[25];
#pragma unroll
  for (int i = 0; i < 25; ++i) s[i] = (5 * t + i) < 5 * (T0 - 1) + 10 ? wb[5 * t + i] : 0.f;
  const float* st = stats + (size_t)b * C0_CH * 2;
  float* ob = out + (size_t)b * C0_CH * ld0 + t;
  for (int c = 0; c < C0_CH; ++c) {
    const float mean = st[c * 2], rstd = st[c * 2 + 1], ga = gamma[c], be = beta[c];
    f32x4 y;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x = 0.f;
#pragma unroll
      for (int j = 0; j < 10; ++j) x = fmaf(w[c * 10 + j], s[5 * e + j], x);
      x = (x - mean) * rstd * ga + be;
      y[e] = 0.5f * x * (1.f + erf_1ulp(x * 0.70710678118654752440f));
    }
    if (t + 3 < T0) {
      *reinterpret_cast<f32x4*>(ob + (size_t)c * ld0) = y;
    } else {
      for (int e = 0; e < 4 && t + e < T0; ++e) ob[(size_t)c * ld0 + e] = y[e];
    }
  }
}

// ---- LayerNorm over channels of a channels-first tensor (optionally of x + r) ----------------
// One block = 64 time steps x all C channels; 16 waves each keep C/16 channels of their 64
// columns in registers (single pass over HBM), partial moments are combined through LDS.
constexpr int LN_WAVES = 16;
constexpr int LN_MAXC = 48;  // channels per wave held in registers (C <= 768)
__global__ void __launch_bounds__(64 * LN_WAVES) ln_cf_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ r,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              const int32_t* __restrict__ lens, int C,
                                                              int ld, float eps, float* __restrict__ y) {
  __shared__ float red[LN_WAVES][64];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + lane;
  const bool ok = t < lens[b];
  const int cpw = C / LN_WAVES;  // host guarantees C % 16 == 0 and cpw <= LN_MAXC
  const size_t base = (size_t)b * C * ld + (size_t)(wv * cpw) * ld + t;
  float v[LN_MAXC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i) {
    v[i] = 0.f;
    if (i < cpw && ok) {
      v[i] = x[base + (size_t)i * ld];
      if (r) v[i] += r[base + (size_t)i * ld];
    }
    s += v[i];
  }
  red[wv][lane] = s;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < LN_WAVES; ++w) tot += red[w][lane];
  const float mean = tot / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i)
    if (i < cpw) {
      const float d = v[i] - mean;
      q = fmaf(d, d, q);
    }
  __syncthreads();
  red[wv][lane] = q;
  __syncthreads();
  tot = 0.f;
#pragma unroll
  for (int w = 0; w < LN_WAVES; ++w) tot += red[w][lane];
  const float rstd = 1.f / sqrtf(tot / (float)C + eps);
  if (!ok) return;
#pragma unroll
  for (int i = 0; i < LN_MAXC; ++i)
    if (i < cpw) {
      const int c = wv * cpw + i;
      y[base + (size_t)i * ld] = (v[i] - mean) * rstd * gamma[c] + beta[c];
    }
}

// ---- batched fp32 MFMA GEMM for attention: C[z] = alpha * A[z] (MxK) * B[z] (KxN) ------------
struct BGemmArgs {
  const float* A; const float* B; float* C;
  long long a_bs, a_hs, b_bs, b_hs, c_bs, c_hs;  // per-utterance / per-head offsets (floats)
  long long sam, sak, sbk, sbn, scm;             // element strides; C is n-contiguous
  const int32_t* lens;                           // T_b
  int m_is_t, n_is_t, k_is_t, fixed;             // dims: T_b where the flag is set, else `fixed`
  int H;
  float alpha;
};
constexpr int BG_LD = 80;  // LDS row stride: 80 % 32 == 16 -> conflict-free fragment reads

__global__ void __launch_bounds__(256) bgemm_kernel(const BGemmArgs a) {
  __shared__ float As[16 * BG_LD];
  __shared__ float Bs[16 * BG_LD];
  const int z = blockIdx.z, b = z / a.H, h = z - b * a.H;
  const int T = a.lens[b];
  const int M = a.m_is_t ? T : a.fixed, N = a.n_is_t ? T : a.fixed, K = a.k_is_t ? T : a.fixed;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  if (m0 >= M || n0 >= N) return;
  const float* A = a.A + b * a.a_bs + h * a.a_hs;
  const float* B = a.B + b * a.b_bs + h * a.b_hs;
  float* C = a.C + b * a.c_bs + h * a.c_hs;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, g = lane >> 4;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool a_mfast = a.sam == 1, b_nfast = a.sbn == 1;
  for (int k0 = 0; k0 < K; k0 += 16) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256;
      int kk, mm;
      if (a_mfast) { kk = e >> 6; mm = e & 63; } else { mm = e >> 4; kk = e & 15; }
      float v = 0.f;
      if (m0 + mm < M && k0 + kk < K) v = A[(long long)(m0 + mm) * a.sam + (long long)(k0 + kk) * a.sak];
      As[kk * BG_LD + mm] = v;
      int nn;
      if (b_nfast) { kk = e >> 6; nn = e & 63; } else { nn = e >> 4; kk = e & 15; }
      v = 0.f;
      if (n0 + nn < N && k0 + kk < K) v = B[(long long)(k0 + kk) * a.sbk + (long long)(n0 + nn) * a.sbn];
      Bs[kk * BG_LD + nn] = v;
    }
    __syncthreads();
#pragma unroll
    for (int cq = 0; cq < 4; ++cq) {
      float av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        av[i] = As[(g + 4 * cq) * BG_LD + wm * 32 + i * 16 + l15];
        bv[i] = Bs[(g + 4 * cq) * BG_LD + wn * 32 + i * 16 + l15];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wm * 32 + i * 16 + 4 * g + r;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 32 + j * 16 + l15;
        if (n < N) C[(long long)m * a.scm + n] = a.alpha * acc[i][j][r];
      }
    }
}

// in-place softmax over the first T_b entries of every row of S[b][h][i][:]
__global__ void __launch_bounds__(256) softmax_rows_kernel(float* __restrict__ S,
                                                           const int32_t* __restrict__ lens, int H,
                                                           int Tmax, int ldS) {
  const int b = blockIdx.z;
  const int T = lens[b];
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);  // one wave per row
  const int h = blockIdx.y;
  if (row >= T) return;
  float* p = S + (((size_t)b * H + h) * Tmax + row) * ldS;
  const int lane = threadIdx.x & 63;
  float mx = -INFINITY;
  for (int j = lane; j < T; j += 64) mx = fmaxf(mx, p[j]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  float sum = 0.f;
  for (int j = lane; j < T; j += 64) {
    const float e = expf(p[j] - mx);
    p[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  for (int j = lane; j < T; j += 64) p[j] = __fdiv_rn(p[j], sum);
}

// ---- fused exact attention: attn.hip (a translation unit of its own: it is compiled with -amdgpu-mfma-vgpr-form) ----------------
int launch_attn_fused(const float* qkv, const int32_t* lens, int D, int hd, int ld, float* out, int T, int H, int B, hipStream_t st);

// ---- k-means assignment (the quantiser's predict(): reference data/encode.py:21-22 via textless KMeansQuantizer -> sklearn) -----
// units[t] = the FIRST k minimising  s_k = cnorm[k] - 2 <x_t, c_k>   (sklearn's dense predict: ||c||^2 - 2 x.c, first minimum).
// The integer step is SPECIFIED to the bit, so that the CPU oracle can restate it exactly (oracle/host_ref.c:
// oracle_kmeans_assign_f32) and identical inputs give identical indices, exact ties included:
//     <x, c_k> = the fp32 fma chain  acc = fmaf(x[d], c_k[d], acc)  over d = 0 .. D-1 in order, from acc = 0;
//     s_k = cnorm[k] - 2 acc (one rounding: 2 acc is exact);  k scanned upwards with a strict '<' from best = +inf, unit 0:
//     a NaN score never wins, a row of NaNs gets unit 0.
// An MFMA GEMM cannot promise that (its k-blocking and internal summation order are the hardware's); the products are 0.08 % of
// the encoder's FLOPs, so they run on the vector ALU: one lane per frame, the centroids staged through LDS 64 dimensions at
// a time and read back as broadcasts, KQ accumulators per lane, the workgroup's four waves splitting the centroids.
// x element (t, d) at x[t * st + d * sd]: channels-first features (st = 1, sd = ld: coalesced) or [T][D] rows (st = D, sd = 1).
constexpr int KM_DC = 64;   // dimensions per LDS chunk
constexpr int KM_TH = 256;  // 4 waves x 64 frames

template <int KQ>
__global__ void __launch_bounds__(KM_TH) kmeans_assign_kernel(const float* __restrict__ x, long long x_bstride, long long st,
                                                              long long sd, const float* __restrict__ centers,
                                                              const float* __restrict__ cnorm, const int32_t* __restrict__ lens,
                                                              int T, int K, int D, int64_t* __restrict__ units) {
  constexpr int KG = 4 * KQ;             // centroids per pass
  constexpr int LR = KM_DC + 4;          // LDS row stride (16-byte aligned rows)
  constexpr int NS = (KG * (KM_DC / 4) + KM_TH - 1) / KM_TH;  // float4 staging slots per thread
  __shared__ __attribute__((aligned(16))) float cs[KG * LR];
  __shared__ float rs[4][64];
  __shared__ int rk[4][64];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int t = blockIdx.x * 64 + lane;
  const int tlen = lens ? lens[b] : T;
  const bool live = t < T && t < tlen;
  const float* xt = x + (size_t)b * x_bstride + (size_t)(t < T ? t : T - 1) * st;
  float best = INFINITY;
  int arg = 0;
  const int nchunk = (D + KM_DC - 1) / KM_DC;
  for (int k0 = 0; k0 < K; k0 += KG) {
    float acc[KQ];
#pragma unroll
    for (int j = 0; j < KQ; ++j) acc[j] = 0.f;
    // chunk of [KG centroids][64 dimensions] -> registers -> LDS, the next chunk's loads in flight behind this one's arithmetic
    f32x4 sv[NS];
    auto stage_load = [&](int c) {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int e = threadIdx.x + i * KM_TH;
        const int row = e / (KM_DC / 4), q = e - row * (KM_DC / 4);
        const int k = k0 + row, d = c * KM_DC + 4 * q;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < KG && k < K) {
          const float* cp = centers + (size_t)k * D + d;
          if (d + 4 <= D && (D & 3) == 0) {
            v = *reinterpret_cast<const f32x4*>(cp);
          } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = d + u < D ? cp[u] : 0.f;
          }
        }
        sv[i] = v;
      }
    };
    auto stage_store = [&]() {
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int e = threadIdx.x + i * KM_TH;
        const int row = e / (KM_DC / 4), q = e - row * (KM_DC / 4);
        if (row < KG) *reinterpret_cast<f32x4*>(cs + row * LR + 4 * q) = sv[i];
      }
    };
    stage_load(0);
    for (int c = 0; c < nchunk; ++c) {
      __syncthreads();  // the previous chunk has been read
      stage_store();
      __syncthreads();
      if (c + 1 < nchunk) stage_load(c + 1);
      const int d0 = c * KM_DC;
      const float* cw = cs + wave * KQ * LR;
#pragma unroll 2
      for (int dd = 0; dd < KM_DC; dd += 4) {
        float xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) xv[u] = d0 + dd + u < D ? xt[(size_t)(d0 + dd + u) * sd] : 0.f;  // (beyond D: c = 0 too, +0 * +0)
#pragma unroll
        for (int j = 0; j < KQ; ++j) {
          const f32x4 cv = *reinterpret_cast<const f32x4*>(cw + j * LR + dd);  // same address in every lane: a broadcast
          acc[j] = fmaf(xv[0], cv[0], acc[j]);
          acc[j] = fmaf(xv[1], cv[1], acc[j]);
          acc[j] = fmaf(xv[2], cv[2], acc[j]);
          acc[j] = fmaf(xv[3], cv[3], acc[j]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < KQ; ++j) {
      const int k = k0 + wave * KQ + j;
      if (k < K) {
        const float sc = cnorm[k] - 2.f * acc[j];
        if (sc < best) {
          best = sc;
          arg = k;
        }
      }
    }
  }
  // the four waves' candidates, lowest score first, lowest index among equal scores
  rs[wave][lane] = best;
  rk[wave][lane] = arg;
  __syncthreads();
  if (wave == 0 && t < T) {
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const float s2 = rs[w][lane];
      const int k2 = rk[w][lane];
      if (s2 < best || (s2 == best && k2 < arg && s2 < INFINITY)) {
        best = s2;
        arg = k2;
      }
    }
    units[(size_t)b * T + t] = live ? arg : 0;
  }
}

// cnorm[k] = the fp32 fma chain s = fmaf(c[d], c[d], s) over d in order (same spec as the scores; oracle_kmeans_cnorm_f32)
__global__ void kmeans_cnorm_kernel(const float* __restrict__ centers, int K, int D, float* __restrict__ cnorm) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  float s = 0.f;
  for (int d = 0; d < D; ++d) s = fmaf(centers[(size_t)k * D + d], centers[(size_t)k * D + d], s);
  cnorm[k] = s;
}

// x as described above; B utterances (grid y); units [B][T] (0 at and beyond an utterance's length)
int launch_kmeans_assign(const float* x, long long x_bstride, long long st, long long sd, const float* centers, const float* cnorm,
                         const int32_t* lens, int B, int T, int K, int D, int64_t* units, hipStream_t stream) {
  if (B <= 0 || T <= 0 || K <= 0 || D <= 0) {
    set_error("kmeans_assign: bad shape (B %d, T %d, K %d, D %d)", B, T, K, D);
    return DISSC_EINVAL;
  }
  const int npass = (K + 127) / 128;
  const int kq = ((K + npass - 1) / npass + 3) / 4;  // centroids per wave and pass
  const dim3 grid((T + 63) / 64, B), block(KM_TH);
  if (kq <= 13) hipLaunchKernelGGL(kmeans_assign_kernel<13>, grid, block, 0, stream, x, x_bstride, st, sd, centers, cnorm, lens, T, K, D, units);
  else if (kq <= 25) hipLaunchKernelGGL(kmeans_assign_kernel<25>, grid, block, 0, stream, x, x_bstride, st, sd, centers, cnorm, lens, T, K, D, units);
  else hipLaunchKernelGGL(kmeans_assign_kernel<32>, grid, block, 0, stream, x, x_bstride, st, sd, centers, cnorm, lens, T, K, D, units);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

}  // namespace dissc

using namespace dissc;

namespace dissc {
// option "attn_fused" (Options::attn_fused, default 1): "attn_fused" option: 0 = S=QK^T -> softmax -> PV through HBM (3 kernels)
// option "hubert_split" (Options::hubert_split, default 1): "hubert_split" option: batches of >= 16 utterances run as 2-4 parts on streams of their own (0 never, 1
                         // unless the batch fills whole workgroup rounds by itself, N >= 2: always N parts)
}

struct dissc_hubert {
  Options opt;  // this handle's snapshot of the tuning options (common.h)
  int n_layers = 6, H = 12, D = 768, F = 3072, CF = 512, K = 0;
  float* w0 = nullptr;   // conv0 [512][10]
  float* gn_g = nullptr; // GroupNorm affine
  float* gn_b = nullptr;
  std::vector<DevConv> fconv;  // feature convs 1..6 (stride 2, GELU)
  std::vector<DevS2tc> ftc;    // the k = 3 ones in polyphase Toom-Cook form ("enc_tc" option; wpack == nullptr: direct)
  float *ln0_g = nullptr, *ln0_b = nullptr;  // feature LayerNorm(512)
  DevConv proj, pos;
  float* centers = nullptr;  // [K][768] k-means centroids
  float *eln_g = nullptr, *eln_b = nullptr;  // encoder.layer_norm
  struct Layer {
    DevConv qkv, out, fc1, fc2;
    float *ln1_g = nullptr, *ln1_b = nullptr, *ln2_g = nullptr, *ln2_b = nullptr;
  };
  std::vector<Layer> layers;
  float* cnorm = nullptr;
  static constexpr int MAX_PARTS = 4;
  hipStream_t side[MAX_PARTS - 1] = {};  // the other parts of a split batch (dissc_hubert_forward)
  hipEvent_t ev_fork = nullptr, ev_join[MAX_PARTS - 1] = {};
  ~dissc_hubert() {
    for (auto st : side)
      if (st) (void)hipStreamDestroy(st);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    for (auto e : ev_join)
      if (e) (void)hipEventDestroy(e);
    for (float* p : {w0, gn_g, gn_b, ln0_g, ln0_b, eln_g, eln_b, cnorm, centers})
      if (p) (void)hipFree(p);
    for (auto& c : fconv) free_conv(c);
    for (auto& c : ftc) free_s2tc(c);
    free_conv(proj); free_conv(pos);
    for (auto& l : layers) {
      free_conv(l.qkv); free_conv(l.out); free_conv(l.fc1); free_conv(l.fc2);
      for (float* p : {l.ln1_g, l.ln1_b, l.ln2_g, l.ln2_b})
        if (p) (void)hipFree(p);
    }
  }
};

static inline size_t rup(size_t x, size_t m) { return (x + m - 1) / m * m; }

static int frames_of(int n) {
  static const int k[NCONV] = {10, 3, 3, 3, 3, 2, 2}, s[NCONV] = {5, 2, 2, 2, 2, 2, 2};
  for (int l = 0; l < NCONV; ++l) n = n >= k[l] ? (n - k[l]) / s[l] + 1 : 0;
  return n;
}
static int frames_after(int n, int upto) {
  static const int k[NCONV] = {10, 3, 3, 3, 3, 2, 2}, s[NCONV] = {5, 2, 2, 2, 2, 2, 2};
  for (int l = 0; l <= upto; ++l) n = n >= k[l] ? (n - k[l]) / s[l] + 1 : 0;
  return n;
}

extern "C" {

int dissc_hubert_frames(int n_samples) { return frames_of(n_samples); }

int dissc_hubert_create(int n_layers, const DisscTensor* weights, size_t n_weights,
                        const float* centers, int n_centers, dissc_hubert_t* out) {
  if (!weights || !out || n_layers < 1 || n_layers > 12) {
    set_error("dissc_hubert_create: bad argument");
    return DISSC_EINVAL;
  }
  std::map<std::string, const DisscTensor*> by;
  for (size_t i = 0; i < n_weights; ++i) by[weights[i].name] = &weights[i];
  dissc_hubert* m = new dissc_hubert();
  m->opt = g_defaults;  // frozen here
  OptScope opt_scope(&m->opt);
  m->n_layers = n_layers;
  int rc = DISSC_OK;
  auto fail = [&](int code) {
    delete m;
    return code;
  };
  auto get = [&](const std::string& name, size_t numel, const float** p) -> int {
    auto it = by.find(name);
    if (it == by.end()) {
      set_error("dissc_hubert_create: missing tensor '%s'", name.c_str());
      return DISSC_ENOTFOUND;
    }
    size_t n = 1;
    for (int d = 0; d < it->second->ndim; ++d) n *= (size_t)it->second->shape[d];
    if (n != numel) {
      set_error("dissc_hubert_create: tensor '%s' has %zu elements, expected %zu", name.c_str(), n, numel);
      return DISSC_EINVAL;
    }
    *p = it->second->data;
    return DISSC_OK;
  };
  auto up = [&](const std::string& name, size_t numel, float** d) -> int {
    const float* p;
    int r = get(name, numel, &p);
    if (r) return r;
    return upload(std::vector<float>(p, p + numel), d);
  };
  const int D = m->D, F = m->F, CF = m->CF;
  const float *w, *b;
  if ((rc = up("feature_extractor.conv_layers.0.0.weight", (size_t)CF * 10, &m->w0))) return fail(rc);
  if ((rc = up("feature_extractor.conv_layers.0.2.weight", CF, &m->gn_g))) return fail(rc);
  if ((rc = up("feature_extractor.conv_layers.0.2.bias", CF, &m->gn_b))) return fail(rc);
  static const int ks[NCONV] = {10, 3, 3, 3, 3, 2, 2};
  m->fconv.resize(NCONV - 1);
  m->ftc.resize(NCONV - 1);
  for (int l = 1; l < NCONV; ++l) {
    char name[96];
    snprintf(name, sizeof(name), "feature_extractor.conv_layers.%d.0.weight", l);
    if ((rc = get(name, (size_t)CF * CF * ks[l], &w))) return fail(rc);
    if (opts().enc_tc && s2tc_supported(CF, CF, ks[l], 2)) {  // k = 3: polyphase Toom-Cook form (conv_s2tc.hip)
      if ((rc = make_s2tc(w, nullptr, CF, CF, m->ftc[l - 1]))) return fail(rc);
      m->ftc[l - 1].act = 1;
      continue;
    }
    if ((rc = make_conv(w, nullptr, CF, CF, ks[l], 1, m->fconv[l - 1], 1, 2, 0))) return fail(rc);
    m->fconv[l - 1].act = 1;
  }
  if ((rc = up("layer_norm.weight", CF, &m->ln0_g)) || (rc = up("layer_norm.bias", CF, &m->ln0_b)))
    return fail(rc);
  if ((rc = get("post_extract_proj.weight", (size_t)D * CF, &w)) || (rc = get("post_extract_proj.bias", D, &b)))
    return fail(rc);
  if ((rc = make_conv(w, b, D, CF, 1, 1, m->proj, 1, 1, 0))) return fail(rc);
  // positional conv: weight already folded (weight_norm dim=2) by the caller: [768][48][128]
  if ((rc = get("encoder.pos_conv.0.weight", (size_t)D * 48 * 128, &w)) ||
      (rc = get("encoder.pos_conv.0.bias", D, &b)))
    return fail(rc);
  if ((rc = make_conv(w, b, D, D, 128, 1, m->pos, 16, 1, 64))) return fail(rc);
  m->pos.act = 1;
  if ((rc = up("encoder.layer_norm.weight", D, &m->eln_g)) || (rc = up("encoder.layer_norm.bias", D, &m->eln_b)))
    return fail(rc);
  m->layers.resize(n_layers);
  for (int i = 0; i < n_layers; ++i) {
    char pre[64];
    snprintf(pre, sizeof(pre), "encoder.layers.%d.", i);
    std::string p(pre);
    auto& L = m->layers[i];
    const float *wq, *wk, *wv, *bq, *bk, *bv;
    if ((rc = get(p + "self_attn.q_proj.weight", (size_t)D * D, &wq)) || (rc = get(p + "self_attn.q_proj.bias", D, &bq)) ||
        (rc = get(p + "self_attn.k_proj.weight", (size_t)D * D, &wk)) || (rc = get(p + "self_attn.k_proj.bias", D, &bk)) ||
        (rc = get(p + "self_attn.v_proj.weight", (size_t)D * D, &wv)) || (rc = get(p + "self_attn.v_proj.bias", D, &bv)))
      return fail(rc);
    std::vector<float> wqkv((size_t)3 * D * D), bqkv(3 * D);
    const float scale = 1.f / sqrtf((float)(D / m->H));  // 0.125: exact in fp32
    for (size_t e = 0; e < (size_t)D * D; ++e) wqkv[e] = wq[e] * scale;
    memcpy(wqkv.data() + (size_t)D * D, wk, (size_t)D * D * 4);
    memcpy(wqkv.data() + (size_t)2 * D * D, wv, (size_t)D * D * 4);
    for (int e = 0; e < D; ++e) bqkv[e] = bq[e] * scale;
    memcpy(bqkv.data() + D, bk, D * 4);
    memcpy(bqkv.data() + 2 * D, bv, D * 4);
    if ((rc = make_conv(wqkv.data(), bqkv.data(), 3 * D, D, 1, 1, L.qkv, 1, 1, 0))) return fail(rc);
    if ((rc = get(p + "self_attn.out_proj.weight", (size_t)D * D, &w)) || (rc = get(p + "self_attn.out_proj.bias", D, &b)))
      return fail(rc);
    if ((rc = make_conv(w, b, D, D, 1, 1, L.out, 1, 1, 0))) return fail(rc);
    if ((rc = get(p + "fc1.weight", (size_t)F * D, &w)) || (rc = get(p + "fc1.bias", F, &b))) return fail(rc);
    if ((rc = make_conv(w, b, F, D, 1, 1, L.fc1, 1, 1, 0))) return fail(rc);
    L.fc1.act = 1;
    if ((rc = get(p + "fc2.weight", (size_t)D * F, &w)) || (rc = get(p + "fc2.bias", D, &b))) return fail(rc);
    if ((rc = make_conv(w, b, D, F, 1, 1, L.fc2, 1, 1, 0))) return fail(rc);
    if ((rc = up(p + "self_attn_layer_norm.weight", D, &L.ln1_g)) || (rc = up(p + "self_attn_layer_norm.bias", D, &L.ln1_b)) ||
        (rc = up(p + "final_layer_norm.weight", D, &L.ln2_g)) || (rc = up(p + "final_layer_norm.bias", D, &L.ln2_b)))
      return fail(rc);
  }
  if (centers && n_centers > 0) {
    m->K = n_centers;
    if ((rc = upload(std::vector<float>(centers, centers + (size_t)n_centers * D), &m->centers))) return fail(rc);
    std::vector<float> cn(n_centers);
    for (int k = 0; k < n_centers; ++k) {  // the fma chain of the assignment's specification (kmeans_assign_kernel)
      float s = 0.f;
      for (int d = 0; d < D; ++d) s = fmaf(centers[(size_t)k * D + d], centers[(size_t)k * D + d], s);
      cn[k] = s;
    }
    if ((rc = upload(cn, &m->cnorm))) return fail(rc);
  }
  *out = m;
  return DISSC_OK;
}

void dissc_hubert_destroy(dissc_hubert_t m) { delete m; }

// The quantiser's predict() on its own (the call dissc_hubert_forward makes on its own features): dense [T][D] rows.
int dissc_kmeans_assign(const float* dense, const float* centers, const float* cnorm, int T, int K, int D, int64_t* units,
                        void* stream_) {
  if (!dense || !centers || !units || T <= 0 || K <= 0 || D <= 0) {
    set_error("dissc_kmeans_assign: bad argument");
    return DISSC_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream_;
  float* cn = nullptr;
  if (!cnorm) {  // (test convenience; allocates and synchronises)
    DISSC_HIP_CHECK(hipMalloc((void**)&cn, (size_t)K * 4));
    hipLaunchKernelGGL(kmeans_cnorm_kernel, dim3((K + 63) / 64), dim3(64), 0, st, centers, K, D, cn);
  }
  int rc = launch_kmeans_assign(dense, 0, D, 1, centers, cnorm ? cnorm : cn, nullptr, 1, T, K, D, units, st);
  if (cn) {
    hipError_t e = hipStreamSynchronize(st);
    (void)hipFree(cn);
    if (!rc && e != hipSuccess) {
      set_error("dissc_kmeans_assign: %s", hipGetErrorString(e));
      rc = DISSC_EHIP;
    }
  }
  return rc;
}

// workspace carve-up (floats unless noted)
struct HubertWs {
  int32_t* lens;     // [7][B]
  double* part;      // [B][ntile][65] lag moments of the waveform
  float* stats;      // [B][512][2]
  float* f[2];       // ping-pong feature buffers, each B*512*ld0 floats
  float *x, *y, *t1; // [B][768][ldT]
  float* qkv;        // [B][2304][ldT]
  float* ffn;        // [B][3072][ldT]
  float* S;          // [B][H][T][ldS]
  size_t bytes;
};

static HubertWs carve(const dissc_hubert* m, int B, int Nmax, void* base_) {
  HubertWs w;
  char* p = (char*)rup((size_t)base_, 256);
  char* p0 = p;
  const int T0 = frames_after(Nmax, 0), T = frames_of(Nmax);
  const size_t ld0 = rup(T0 > 0 ? T0 : 1, 4), ldT = rup(T > 0 ? T : 1, 4);
  const int ntile = (T0 + C0_TILE - 1) / C0_TILE + 1;
  auto take = [&](size_t bytes) {
    char* r = p;
    p += rup(bytes, 256);
    return r;
  };
  w.lens = (int32_t*)take((size_t)NCONV * B * 4);
  w.part = (double*)take((size_t)B * ntile * C0_NM * 8);
  w.stats = (float*)take((size_t)B * C0_CH * 2 * 4);
  w.f[0] = (float*)take((size_t)B * 512 * ld0 * 4);
  const size_t ld1 = rup(frames_after(Nmax, 1) > 0 ? frames_after(Nmax, 1) : 1, 4);
  w.f[1] = (float*)take((size_t)B * 512 * ld1 * 4);
  w.x = (float*)take((size_t)B * m->D * ldT * 4);
  w.y = (float*)take((size_t)B * m->D * ldT * 4);
  w.t1 = (float*)take((size_t)B * m->D * ldT * 4);
  w.qkv = (float*)take((size_t)B * 3 * m->D * ldT * 4);
  w.ffn = (float*)take((size_t)B * m->F * ldT * 4);
  const bool need_s = !(opts().attn_fused && m->D / m->H == 64);  // the fused attention keeps S on chip
  w.S = (float*)take(need_s ? (size_t)B * m->H * (size_t)(T > 0 ? T : 1) * ldT * 4 : 256);
  w.bytes = (size_t)(p - p0) + 256;
  return w;
}

constexpr int HUBERT_SPLIT_MIN_B = 16;

size_t dissc_hubert_workspace_bytes(dissc_hubert_t m, int B, int Nmax) {
  if (!m || B <= 0 || Nmax <= 0) return 0;
  OptScope opt_scope(&m->opt);
  size_t whole = carve(m, B, Nmax, nullptr).bytes;
  if (B >= HUBERT_SPLIT_MIN_B)  // the parts of a split batch side by side (whatever the option says when the forward runs)
    for (int np = 2; np <= dissc_hubert::MAX_PARTS; ++np) {
      const size_t parts = np * rup(carve(m, (B + np - 1) / np, Nmax, nullptr).bytes, 256) + 256;
      if (parts > whole) whole = parts;
    }
  return whole;
}

static int hubert_forward_part(dissc_hubert_t m, const float* wav, const int32_t* n_samples, int B, int Nmax,
                               float* dense_out, int64_t* units_out, void* workspace, hipStream_t st);

int dissc_hubert_forward(dissc_hubert_t m, const float* wav, const int32_t* n_samples, int B, int Nmax,
                         float* dense_out, int64_t* units_out, void* workspace, size_t ws_bytes,
                         void* stream_) {
  if (!m || !wav || !workspace || B <= 0 || Nmax <= 0 || (units_out && m->K <= 0)) {
    set_error("dissc_hubert_forward: bad argument");
    return DISSC_EINVAL;
  }
  OptScope opt_scope(&m->opt);  // the forward reads this handle's options only
  if (ws_bytes < dissc_hubert_workspace_bytes(m, B, Nmax)) {
    set_error("dissc_hubert_forward: workspace %zu < %zu bytes", ws_bytes,
              dissc_hubert_workspace_bytes(m, B, Nmax));
    return DISSC_ENOMEM;
  }
  hipStream_t st = (hipStream_t)stream_;
  // "hubert_split": 0 never, 2 always, 1 (default) unless the whole batch already fills whole rounds: with rows of T frames
  // the linears launch B ceil(T / 64) column tiles per M tile, and when that is a multiple of the CU count (32 x 10 s: 8 x 32 =
  // 256) every round is full and splitting only costs (28.0 -> 28.9 ms).  The lengths live on the device, so the rows decide.
  bool split = opts().hubert_split != 0 && B >= HUBERT_SPLIT_MIN_B;
  if (split && opts().hubert_split == 1) {
    static int n_cu_of[64] = {0};  // per device (a process may hold models on several GPUs)
    int dev = 0;
    DISSC_HIP_CHECK(hipGetDevice(&dev));
    int& n_cu = n_cu_of[dev & 63];
    if (!n_cu) {
      hipDeviceProp_t prop;
      DISSC_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
      n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int tiles = B * ((frames_of(Nmax) + 63) / 64);
    if (tiles % n_cu == 0) split = false;
  }
  if (!split) return hubert_forward_part(m, wav, n_samples, B, Nmax, dense_out, units_out, workspace, st);
  // Parts on streams of their own.  The layers of one batch depend on each other, so every launch ends in a partly filled
  // round of workgroups that nothing covers -- and only special shapes avoid it (32 x 10 s: 8 column tiles x 32 = exactly one
  // workgroup per CU and M tile; the same audio in rows of 10.5 s, or ragged, costs 10-15 % more).  Independent parts
  // fill each other's tails: ragged 32.0 -> 30.2 ms per 320 s, the lucky shape 28.0 -> 28.8 (tools/encode_ragged.py).
  // Utterances are independent (per-utterance GroupNorm statistics, masked attention), so the units do not change.
  // parts of >= 8 utterances, at most 4 (32 ragged utterances of 8-12 s: 32.0 ms whole, 31.0 / 30.6 / 30.2 in 2 / 3 / 4 parts)
  int nparts = opts().hubert_split >= 2 ? opts().hubert_split : B / 8;
  if (nparts < 2) nparts = 2;
  if (nparts > dissc_hubert::MAX_PARTS) nparts = dissc_hubert::MAX_PARTS;
  if (!m->ev_fork) DISSC_HIP_CHECK(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
  for (int i = 0; i + 1 < nparts; ++i)
    if (!m->side[i]) {
      DISSC_HIP_CHECK(hipStreamCreateWithFlags(&m->side[i], hipStreamNonBlocking));
      DISSC_HIP_CHECK(hipEventCreateWithFlags(&m->ev_join[i], hipEventDisableTiming));
    }
  const int Bp = (B + nparts - 1) / nparts;
  const int T = frames_of(Nmax);
  const size_t ldT = rup(T > 0 ? T : 1, 4);
  const size_t part_ws = rup(carve(m, Bp, Nmax, nullptr).bytes, 256);
  DISSC_HIP_CHECK(hipEventRecord(m->ev_fork, st));
  // From here on side streams may hold work: an error must not return before `st` waits for every side stream that was
  // forked (the caller may free or reuse the workspace as soon as `st` is done).  The side streams and events belong to
  // the model: ONE forward of a model at a time (a model is single-stream, include/dissc_hip.h).
  int rc = DISSC_OK;
  auto note = [&](hipError_t e, const char* what) {
    if (e != hipSuccess && !rc) {
      set_error("dissc_hubert_forward: %s: %s", what, hipGetErrorString(e));
      rc = DISSC_EHIP;
    }
    return e == hipSuccess;
  };
  for (int i = 0; i < nparts; ++i) {
    const int b0 = i * Bp, bn = (b0 + Bp <= B ? Bp : B - b0);
    if (bn <= 0) break;
    hipStream_t si = i == 0 ? st : m->side[i - 1];
    if (i > 0 && !note(hipStreamWaitEvent(si, m->ev_fork, 0), "fork")) continue;  // nothing launched on this stream
    if (!rc) {
      const int r = hubert_forward_part(m, wav + (size_t)b0 * Nmax, n_samples ? n_samples + b0 : nullptr, bn, Nmax,
                                        dense_out ? dense_out + (size_t)b0 * m->D * ldT : nullptr,
                                        units_out ? units_out + (size_t)b0 * T : nullptr, (char*)workspace + i * part_ws, si);
      if (r && !rc) rc = r;
    }
    if (i > 0) {  // join even after an error: kernels already on `si` must finish before `st` moves on
      if (note(hipEventRecord(m->ev_join[i - 1], si), "join record"))
        note(hipStreamWaitEvent(st, m->ev_join[i - 1], 0), "join wait");
      else
        (void)hipStreamSynchronize(si);
    }
  }
  return rc;
}

static int hubert_forward_part(dissc_hubert_t m, const float* wav, const int32_t* n_samples, int B, int Nmax,
                               float* dense_out, int64_t* units_out, void* workspace, hipStream_t st) {
  HubertWs w = carve(m, B, Nmax, workspace);
  const int T0 = frames_after(Nmax, 0), T = frames_of(Nmax);
  if (T <= 0) {
    set_error("dissc_hubert_forward: %d samples give no frame (need >= 400)", Nmax);
    return DISSC_EINVAL;
  }
  const int ld0 = (int)rup(T0, 4), ldT = (int)rup(T, 4);
  const int ntile = (T0 + C0_TILE - 1) / C0_TILE + 1;
  const int D = m->D, H = m->H, hd = D / H;
  int rc;
  hipLaunchKernelGGL(hubert_lengths_kernel, dim3((B + 63) / 64), dim3(64), 0, st, n_samples, B, Nmax, w.lens);
  // conv0 + GroupNorm + GELU
  dim3 g0((T0 + C0_TILE - 1) / C0_TILE, B);
  hipLaunchKernelGGL(conv0_moments_kernel, g0, dim3(256), 0, st, wav, Nmax, w.lens, ntile, w.part);
  hipLaunchKernelGGL(conv0_finalize_kernel, dim3(C0_CH / 128, B), dim3(128), 0, st, w.part, m->w0, w.lens, ntile,
                     1e-5f, w.stats);
  hipLaunchKernelGGL(conv0_apply_kernel, g0, dim3(256), 0, st, wav, Nmax, m->w0, m->gn_g, m->gn_b, w.stats,
                     w.lens, w.f[0], ld0);
  // strided feature convs (GELU fused), ping-pong
  int cur = 0, ld_in = ld0, n_in = Nmax;
  n_in = T0;
  for (int l = 1; l < NCONV; ++l) {
    const int n_out = frames_after(Nmax, l);
    const int ld_out = (int)rup(n_out > 0 ? n_out : 1, 4);
    ConvIO io;
    io.lengths_in = w.lens + (size_t)(l - 1) * B;
    io.lengths_out = w.lens + (size_t)l * B;
    if (m->ftc[l - 1].wpack) {
      if ((rc = run_s2tc(m->ftc[l - 1], w.f[cur], w.f[cur ^ 1], io.lengths_in, io.lengths_out, n_in, n_out, B, ld_in, ld_out,
                         n_out, st)))
        return rc;
    } else if ((rc = run_conv_ex(m->fconv[l - 1], w.f[cur], w.f[cur ^ 1], nullptr, io, B, 512, ld_in, ld_out, n_out,
                                 1.0f, EPI_STORE, st)))
      return rc;
    cur ^= 1;
    ld_in = ld_out;
    n_in = n_out;
  }
  (void)n_in;
  const int32_t* lensT = w.lens + (size_t)(NCONV - 1) * B;
  ConvIO ioT;
  ioT.lengths_in = lensT;
  dim3 gln((T + 63) / 64, B);
  // LayerNorm(512) -> Linear(512,768)
  float* fn = w.f[cur ^ 1];  // free buffer, [B][512][ldT] fits (ldT <= ld of that buffer)
  hipLaunchKernelGGL(ln_cf_kernel, gln, dim3(64 * LN_WAVES), 0, st, w.f[cur], (const float*)nullptr, m->ln0_g, m->ln0_b,
                     lensT, 512, ldT, 1e-5f, fn);
  if ((rc = run_conv_ex(m->proj, fn, w.x, nullptr, ioT, B, 512, ldT, ldT, T, 1.0f, EPI_STORE, st))) return rc;
  // x = LN(x + GELU(pos_conv(x)))
  if ((rc = run_conv_ex(m->pos, w.x, w.y, w.x, ioT, B, D, ldT, ldT, T, 1.0f, EPI_RES, st))) return rc;
  hipLaunchKernelGGL(ln_cf_kernel, gln, dim3(64 * LN_WAVES), 0, st, w.y, (const float*)nullptr, m->eln_g, m->eln_b, lensT,
                     D, ldT, 1e-5f, w.x);
  for (int i = 0; i < m->n_layers; ++i) {
    auto& L = m->layers[i];
    if ((rc = run_conv_ex(L.qkv, w.x, w.qkv, nullptr, ioT, B, D, ldT, ldT, T, 1.0f, EPI_STORE, st))) return rc;
    if (opts().attn_fused && hd == 64) {
      if ((rc = launch_attn_fused(w.qkv, lensT, D, hd, ldT, w.t1, T, H, B, st))) return rc;
    } else {
    BGemmArgs a;
      // S[b,h][i][j] = sum_d Q[d][i] K[d][j]      (Q already scaled by 1/sqrt(hd))
      a.A = w.qkv; a.B = w.qkv + (size_t)D * ldT; a.C = w.S;
      a.a_bs = (long long)3 * D * ldT; a.a_hs = (long long)hd * ldT; a.b_bs = a.a_bs; a.b_hs = a.a_hs;
      a.c_bs = (long long)H * T * ldT; a.c_hs = (long long)T * ldT;
      a.sam = 1; a.sak = ldT; a.sbk = ldT; a.sbn = 1; a.scm = ldT;
      a.lens = lensT; a.m_is_t = 1; a.n_is_t = 1; a.k_is_t = 0; a.fixed = hd; a.H = H; a.alpha = 1.f;
      dim3 gs((T + 63) / 64, (T + 63) / 64, B * H);
      hipLaunchKernelGGL(bgemm_kernel, gs, dim3(256), 0, st, a);
      hipLaunchKernelGGL(softmax_rows_kernel, dim3((T + 3) / 4, H, B), dim3(256), 0, st, w.S, lensT, H, T, ldT);
      // O[h*hd + d][i] = sum_j V[d][j] P[i][j]  -> t1 (channels-first)
      a.A = w.qkv + (size_t)2 * D * ldT; a.B = w.S; a.C = w.t1;
      a.a_bs = (long long)3 * D * ldT; a.a_hs = (long long)hd * ldT;
      a.b_bs = (long long)H * T * ldT; a.b_hs = (long long)T * ldT;
      a.c_bs = (long long)D * ldT; a.c_hs = (long long)hd * ldT;
      a.sam = ldT; a.sak = 1; a.sbk = 1; a.sbn = ldT; a.scm = ldT;
      a.m_is_t = 0; a.n_is_t = 1; a.k_is_t = 1; a.fixed = hd;
      dim3 go((T + 63) / 64, (hd + 63) / 64, B * H);
      hipLaunchKernelGGL(bgemm_kernel, go, dim3(256), 0, st, a);
    }
    // x = LN(x + out_proj(O))
    if ((rc = run_conv_ex(L.out, w.t1, w.y, w.x, ioT, B, D, ldT, ldT, T, 1.0f, EPI_RES, st))) return rc;
    hipLaunchKernelGGL(ln_cf_kernel, gln, dim3(64 * LN_WAVES), 0, st, w.y, (const float*)nullptr, L.ln1_g, L.ln1_b, lensT, D,
                       ldT, 1e-5f, w.x);
    // x = LN(x + fc2(GELU(fc1(x))))
    if ((rc = run_conv_ex(L.fc1, w.x, w.ffn, nullptr, ioT, B, D, ldT, ldT, T, 1.0f, EPI_STORE, st))) return rc;
    if ((rc = run_conv_ex(L.fc2, w.ffn, w.y, w.x, ioT, B, m->F, ldT, ldT, T, 1.0f, EPI_RES, st))) return rc;
    hipLaunchKernelGGL(ln_cf_kernel, gln, dim3(64 * LN_WAVES), 0, st, w.y, (const float*)nullptr, L.ln2_g, L.ln2_b, lensT, D,
                       ldT, 1e-5f, w.x);
  }
  if (dense_out)  // [B][768][ldT] channels-first
    DISSC_HIP_CHECK(hipMemcpyAsync(dense_out, w.x, (size_t)B * D * ldT * 4, hipMemcpyDeviceToDevice, st));
  if (units_out) {
    if ((rc = launch_kmeans_assign(w.x, (long long)D * ldT, 1, ldT, m->centers, m->cnorm, lensT, B, T, m->K, D, units_out, st)))
      return rc;
  }
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

}  // extern "C"

// ---- stand-alone test / gate surface of the stride-2 feature convs (moved here from conv_s2tc.hip in round 6) ----
namespace dissc {
__global__ void s2_out_lengths_kernel(const int32_t* __restrict__ lin, int B, int k, int32_t* __restrict__ lout) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) lout[b] = lin[b] >= k ? (lin[b] - k) / 2 + 1 : 0;
}
}  // namespace dissc

extern "C" {

// Stand-alone stride-2 VALID conv (tests / gates): x f32 [B,Cin,ldx] -> y f32 [B,Cout,ldo], output length (len - k) / 2 + 1,
// w HOST [Cout,Cin,k], optional exact-erf GELU.  form 0 = the direct implicit GEMM (conv_mfma32.hip), 1 = the polyphase Toom-Cook form
// (k = 3 only; experimental/csrc/conv_s2tc.hip: DISSC_EXPERIMENTAL=1 builds, an error otherwise).  Synchronous.
int dissc_conv1d_s2(const float* x, const float* w_host, const float* bias_host, float* y, const int32_t* lengths_in, int B,
                    int Cin, int Cout, int k, int ldx, int ldo, int Lmax_in, int act, int form, void* stream_) {
  if (!x || !w_host || !y || B <= 0 || Lmax_in < k || (form != 0 && form != 1)) {
    set_error("dissc_conv1d_s2: bad argument");
    return DISSC_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream_;
  const int Lmax_out = (Lmax_in - k) / 2 + 1;
  int32_t* lout = nullptr;
  if (lengths_in) {
    DISSC_HIP_CHECK(hipMalloc((void**)&lout, (size_t)B * 4));
    hipLaunchKernelGGL(s2_out_lengths_kernel, dim3((B + 63) / 64), dim3(64), 0, st, lengths_in, B, k, lout);
  }
  int rc;
  if (form == 1) {
    DevS2tc dc;
    if (k != 3) {
      set_error("dissc_conv1d_s2: the polyphase form is k = 3 only");
      rc = DISSC_EINVAL;
    } else if (!(rc = make_s2tc(w_host, bias_host, Cout, Cin, dc))) {
      dc.act = act;
      rc = run_s2tc(dc, x, y, lengths_in, lout, Lmax_in, Lmax_out, B, ldx, ldo, Lmax_out, st);
    }
    hipError_t e = hipStreamSynchronize(st);
    free_s2tc(dc);
    if (!rc && e != hipSuccess) {
      set_error("dissc_conv1d_s2: %s", hipGetErrorString(e));
      rc = DISSC_EHIP;
    }
  } else {
    DevConv dc;
    if (!(rc = make_conv(w_host, bias_host, Cout, Cin, k, 1, dc, 1, 2, 0))) {
      dc.act = act;
      ConvIO io;
      io.lengths_in = lengths_in; io.lengths_out = lout; io.len_default = Lmax_in; io.olen_default = Lmax_out;
      rc = run_conv_ex(dc, x, y, nullptr, io, B, Cin, ldx, ldo, Lmax_out, 1.0f, EPI_STORE, st);
    }
    hipError_t e = hipStreamSynchronize(st);
    free_conv(dc);
    if (!rc && e != hipSuccess) {
      set_error("dissc_conv1d_s2: %s", hipGetErrorString(e));
      rc = DISSC_EHIP;
    }
  }
  if (lout) (void)hipFree(lout);
  return rc;
}

// Diagnostics: average ms of `iters` launches of one stride-2, k = 3 conv (C -> C channels, L input samples per utterance,
// GELU fused) on synthetic data; form as above.
int dissc_conv_s2_bench(int B, int C, int L, int form, int iters, float* ms_out) {
  if (!ms_out || B <= 0 || L < 3 || iters <= 0 || (form != 0 && form != 1)) {
    set_error("dissc_conv_s2_bench: bad argument");
    return DISSC_EINVAL;
  }
  std::vector<float> w((size_t)C * C * 3);
  uint32_t s = 12345u;
  for (auto& v : w) {
    s = s * 1664525u + 1013904223u;
    v = ((s >> 8) / 16777216.0f - 0.5f) * 0.1f;
  }
  const int Lo = (L - 3) / 2 + 1;
  const int ldx = (L + 3) / 4 * 4, ldo = (Lo + 3) / 4 * 4;
  const size_t nx = (size_t)B * C * ldx, no = (size_t)B * C * ldo;
  std::vector<float> hx(nx);
  for (auto& v : hx) {
    s = s * 1664525u + 1013904223u;
    v = ((s >> 8) / 16777216.0f - 0.3f) * 2.f;
  }
  float *x = nullptr, *y = nullptr;
  DISSC_HIP_CHECK(hipMalloc((void**)&x, nx * 4));
  DISSC_HIP_CHECK(hipMalloc((void**)&y, no * 4));
  DISSC_HIP_CHECK(hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice));
  DevS2tc tc;
  DevConv dc;
  int rc = form == 1 ? make_s2tc(w.data(), nullptr, C, C, tc) : make_conv(w.data(), nullptr, C, C, 3, 1, dc, 1, 2, 0);
  tc.act = 1;
  dc.act = 1;
  ConvIO io;
  io.len_default = L; io.olen_default = Lo;
  float* tlbuf = nullptr;  // diagnostics (DISSC_TIMELINE): a kernel's timeline stamps, handed over as the accumulator pointer
  const char* tlp = getenv("DISSC_TIMELINE");
  if (tlp) {
    DISSC_HIP_CHECK(hipMalloc((void**)&tlbuf, (size_t)4096 * 64));
    DISSC_HIP_CHECK(hipMemset(tlbuf, 0, (size_t)4096 * 64));
  }
  auto once = [&]() {
    return form == 1 ? run_s2tc(tc, x, y, nullptr, nullptr, L, Lo, B, ldx, ldo, Lo, nullptr)
                     : run_conv_ex(dc, x, y, nullptr, tlbuf, io, B, C, ldx, ldo, Lo, 1.0f, EPI_STORE, 1.f, nullptr);
  };
  hipEvent_t e0, e1;
  DISSC_HIP_CHECK(hipEventCreate(&e0));
  DISSC_HIP_CHECK(hipEventCreate(&e1));
  for (int it = 0; it < 2 && !rc; ++it) rc = once();
  DISSC_HIP_CHECK(hipEventRecord(e0, nullptr));
  for (int it = 0; it < iters && !rc; ++it) rc = once();
  DISSC_HIP_CHECK(hipEventRecord(e1, nullptr));
  hipError_t e = hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *ms_out = ms / iters;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (tlbuf) {
    std::vector<char> hb((size_t)4096 * 64);
    if (hipMemcpy(hb.data(), tlbuf, hb.size(), hipMemcpyDeviceToHost) == hipSuccess)
      if (FILE* f = fopen(tlp, "wb")) {
        fwrite(hb.data(), 1, hb.size(), f);
        fclose(f);
      }
    (void)hipFree(tlbuf);
  }
  (void)hipFree(x); (void)hipFree(y);
  free_s2tc(tc);
  free_conv(dc);
  if (rc) return rc;
  DISSC_HIP_CHECK(e);
  return DISSC_OK;
}

}  // extern "C"
