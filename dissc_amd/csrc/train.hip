// Training step of DISSC's length / pitch predictors on the MI355X.
// Replaces one iteration of the loops at reference train_len_predictor.py:57-68 and train_f0_predictor.py:58-66:
// model.train() forward (token-embedding masking, PositionalEncoding dropout, BatchNorm with batch statistics),
// LenSumLoss (loss/len_loss.py:16-30) or PitchLoss (loss/pitch_loss.py:6-27), backward, torch.optim.Adam.step().
//
// The three models are one description: a chain of Conv1d(k=3, "same") [+ BatchNorm1d] [+ LeakyReLU(0.01)] layers
// over [tok_emb | spk_emb (+pe)] and one or two scalar heads.  Wide convolutions run forward AND backward-data on the
// fp32 matrix cores through conv_mfma32_kernel (backward-data = the same conv with the transposed, tap-flipped
// weights); both packed copies are rebuilt ON THE DEVICE from the master weights after every Adam step.  Weight
// gradients are per-utterance partial correlations reduced in a fixed order; BatchNorm statistics / gradients, the
// losses and the embedding gradients are fixed-order reductions too, so a step is deterministic.  Random masks are
// INPUTS (the caller draws them): the reference uses the CUDA generator, which cannot be reproduced anyway.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>

#include "common.h"

namespace dissc {

constexpr float T_SLOPE = 0.01f;
constexpr float T_BN_EPS = 1e-5f;
constexpr float T_BN_MOM = 0.1f;

// x0[b][c][t]: c < E: keep[b,t] * tok[seq[b,t]][c]; c >= E: (spk[spk_id[b]][c-E] + pe[t][c-E]) * pe_mult[b][t][c-E]
__global__ void train_embed_kernel(const int64_t* __restrict__ seq, const int64_t* __restrict__ spk,
                                   const float* __restrict__ keep, const float* __restrict__ pe_mult,
                                   const float* __restrict__ tok, const float* __restrict__ spe,
                                   const float* __restrict__ pe, int L, int E, int n_tok, int n_spk,
                                   float* __restrict__ x, int ld) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= L) return;
  float v;
  if (c < E) {
    long long id = seq[(size_t)b * L + t];
    id = id < 0 ? 0 : (id >= n_tok ? n_tok - 1 : id);
    v = tok[(size_t)id * E + c];
    if (keep) v *= keep[(size_t)b * L + t];
  } else {
    long long id = spk[b];
    id = id < 0 ? 0 : (id >= n_spk ? n_spk - 1 : id);
    v = spe[(size_t)id * E + (c - E)];
    if (pe) v += pe[(size_t)t * E + (c - E)];
    if (pe_mult) v *= pe_mult[((size_t)b * L + t) * E + (c - E)];
  }
  x[((size_t)b * 2 * E + c) * ld + t] = v;
}

__device__ __forceinline__ double block_sum_d(double v, double* red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if (lane == 0) red[wv] = v;
  __syncthreads();
  double s = 0.0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
  return s;
}

// per channel: batch mean / biased variance over (b, t < L); running statistics updated in place
__global__ void __launch_bounds__(256) train_bn_stats_kernel(const float* __restrict__ z, int B, int C, int L, int ld,
                                                             float* __restrict__ mean, float* __restrict__ invstd,
                                                             float* __restrict__ run_mean, float* __restrict__ run_var) {
  __shared__ double red[4];
  const int c = blockIdx.x;
  double s = 0.0;
  for (int b = 0; b < B; ++b)
    for (int t = threadIdx.x; t < L; t += 256) s += z[((size_t)b * C + c) * ld + t];
  const double n = (double)B * L;
  const double m = block_sum_d(s, red) / n;
  double q = 0.0;
  for (int b = 0; b < B; ++b)
    for (int t = threadIdx.x; t < L; t += 256) {
      const double d = z[((size_t)b * C + c) * ld + t] - m;
      q += d * d;
    }
  const double var = block_sum_d(q, red) / n;
  if (threadIdx.x == 0) {
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)T_BN_EPS));
    run_mean[c] = (1.f - T_BN_MOM) * run_mean[c] + T_BN_MOM * (float)m;
    run_var[c] = (1.f - T_BN_MOM) * run_var[c] + T_BN_MOM * (float)(var * n / (n > 1.0 ? n - 1.0 : 1.0));
  }
}

// a = [leaky]( bn ? gamma * (z - mean) * invstd + beta : z )
__global__ void train_act_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                 const float* __restrict__ invstd, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, int C, int L, int ld, int bn, int leaky,
                                 float* __restrict__ a) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= L) return;
  const size_t i = ((size_t)b * C + c) * ld + t;
  float v = z[i];
  if (bn) v = (v - mean[c]) * invstd[c] * gamma[c] + beta[c];
  if (leaky) v = v > 0.f ? v : v * T_SLOPE;
  a[i] = v;
}

// per channel: s1 = sum dy, s2 = sum dy * xhat with dy = da * leaky'(a)   (dgamma = s2, dbeta = s1)
__global__ void __launch_bounds__(256) train_bn_bwd_reduce_kernel(const float* __restrict__ da, const float* __restrict__ a,
                                                                  const float* __restrict__ z, const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd, int B, int C, int L,
                                                                  int ld, int leaky, float* __restrict__ dgamma,
                                                                  float* __restrict__ dbeta) {
  __shared__ double red[4];
  const int c = blockIdx.x;
  double s1 = 0.0, s2 = 0.0;
  const float m = mean[c], is = invstd[c];
  for (int b = 0; b < B; ++b)
    for (int t = threadIdx.x; t < L; t += 256) {
      const size_t i = ((size_t)b * C + c) * ld + t;
      float dy = da[i];
      if (leaky && !(a[i] > 0.f)) dy *= T_SLOPE;
      s1 += dy;
      s2 += (double)dy * (double)((z[i] - m) * is);
    }
  s1 = block_sum_d(s1, red);
  s2 = block_sum_d(s2, red);
  if (threadIdx.x == 0) {
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
  }
}

// dz = bn ? gamma * invstd * (dy - s1/N - xhat * s2/N) : dy,   dy = da * leaky'(a)
__global__ void train_bn_bwd_apply_kernel(const float* __restrict__ da, const float* __restrict__ a,
                                          const float* __restrict__ z, const float* __restrict__ mean,
                                          const float* __restrict__ invstd, const float* __restrict__ gamma,
                                          const float* __restrict__ dgamma, const float* __restrict__ dbeta, int B,
                                          int C, int L, int ld, int bn, int leaky, float* __restrict__ dz) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= L) return;
  const size_t i = ((size_t)b * C + c) * ld + t;
  float dy = da[i];
  if (leaky && !(a[i] > 0.f)) dy *= T_SLOPE;
  if (bn) {
    const float n = (float)B * (float)L;
    const float xh = (z[i] - mean[c]) * invstd[c];
    dy = gamma[c] * invstd[c] * (dy - dbeta[c] / n - xh * dgamma[c] / n);
  }
  dz[i] = dy;
}

// per-utterance partial weight gradient: part[b][co][ci][j] = sum_t dz[b][co][t] * a_in[b][ci][t + j - pad]
constexpr int WG_T = 64;
__global__ void __launch_bounds__(256) train_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ ain,
                                                          int Cout, int Cin, int K, int L, int ld_o, int ld_i,
                                                          float* __restrict__ part) {
  __shared__ float ds[16][WG_T];
  __shared__ float as[16][WG_T + 4];
  const int b = blockIdx.z, co0 = blockIdx.x * 16, ci0 = blockIdx.y * 16;
  const int col = threadIdx.x >> 4, cil = threadIdx.x & 15;
  const int pad = (K - 1) / 2;
  float acc[3] = {0.f, 0.f, 0.f};
  for (int t0 = 0; t0 < L; t0 += WG_T) {
    __syncthreads();
    for (int e = threadIdx.x; e < 16 * WG_T; e += 256) {
      const int r = e / WG_T, u = e - r * WG_T;
      const int t = t0 + u;
      ds[r][u] = (t < L && co0 + r < Cout) ? dz[((size_t)b * Cout + co0 + r) * ld_o + t] : 0.f;
    }
    for (int e = threadIdx.x; e < 16 * (WG_T + 2); e += 256) {
      const int r = e / (WG_T + 2), u = e - r * (WG_T + 2);
      const int t = t0 + u - pad;
      as[r][u] = (t >= 0 && t < L && ci0 + r < Cin) ? ain[((size_t)b * Cin + ci0 + r) * ld_i + t] : 0.f;
    }
    __syncthreads();
    for (int u = 0; u < WG_T; ++u) {
      const float d = ds[col][u];
      for (int j = 0; j < K && j < 3; ++j) acc[j] = fmaf(d, as[cil][u + j], acc[j]);
    }
  }
  if (co0 + col < Cout && ci0 + cil < Cin)
    for (int j = 0; j < K && j < 3; ++j)
      part[(((size_t)b * Cout + co0 + col) * Cin + ci0 + cil) * K + j] = acc[j];
}

// dw[i] = sum_b part[b][i] (fixed order)
// The same product on the matrix cores (k = 3 layers with >= 32 output rows): per utterance a [Cout x L] x [L x 3 Cin]
// GEMM whose reduction axis is time.  One workgroup = 32 output rows x (4 waves x 32 input channels); a k-step of
// v_mfma_f32_32x32x2 is two time positions, the three taps are three accumulators fed from the same LDS window
// shifted by one.  Rows are padded to an odd stride so that the fragment reads (32 rows x 1 column) are conflict-free.
constexpr int WGM_T = 64;             // time positions per LDS chunk
constexpr int WGM_AW = WGM_T + 8;     // a_in window: t0 - 4 .. t0 + 67 (16-byte aligned loads)
constexpr int WGM_LDD = WGM_T + 1;    // odd row strides
constexpr int WGM_LDA = WGM_AW + 1;
constexpr int WGM_SPLIT = 2;          // time halves per utterance (one partial each)
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) train_wgrad_mfma_kernel(const float* __restrict__ dz, const float* __restrict__ ain,
                                                               int Cout, int Cin, int L, int ld_o, int ld_i,
                                                               float* __restrict__ part) {
  __shared__ float ds[32 * WGM_LDD];
  __shared__ float as[128 * WGM_LDA];
  const int b = blockIdx.z / WGM_SPLIT, half = blockIdx.z % WGM_SPLIT;
  const int co0 = blockIdx.x * 32, ci0 = blockIdx.y * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const bool live = ci0 + wave * 32 < Cin;  // wave-uniform
  const int span = ((L + WGM_SPLIT - 1) / WGM_SPLIT + WGM_T - 1) / WGM_T * WGM_T;
  const int tbeg = half * span, tend = (tbeg + span < L) ? tbeg + span : L;
  f32x16 acc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  const float* dzb = dz + ((size_t)b * Cout + co0) * ld_o;
  const float* ab = ain + ((size_t)b * Cin + ci0) * ld_i;
  constexpr int ND = 32 * (WGM_T / 4) / 256;             // float4 slots per thread: dz tile (2)
  constexpr int NA = 128 * (WGM_AW / 4) / 256;           //                          a_in tile (9)
  f32x4 rd[ND], ra[NA];
  auto load = [&](int t0) {
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int e = tid + i * 256, r = e / (WGM_T / 4), v = e - r * (WGM_T / 4), t = t0 + 4 * v;
      rd[i] = (t < ld_o && co0 + r < Cout) ? *reinterpret_cast<const f32x4*>(dzb + (size_t)r * ld_o + t)
                                           : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = tid + i * 256, r = e / (WGM_AW / 4), v = e - r * (WGM_AW / 4), t = t0 - 4 + 4 * v;
      ra[i] = (t >= 0 && t < ld_i && ci0 + r < Cin) ? *reinterpret_cast<const f32x4*>(ab + (size_t)r * ld_i + t)
                                                    : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto store = [&](int t0) {
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      const int e = tid + i * 256, r = e / (WGM_T / 4), v = e - r * (WGM_T / 4), t = t0 + 4 * v;
#pragma unroll
      for (int q = 0; q < 4; ++q) ds[r * WGM_LDD + 4 * v + q] = (t + q < tend) ? rd[i][q] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int e = tid + i * 256, r = e / (WGM_AW / 4), v = e - r * (WGM_AW / 4), t = t0 - 4 + 4 * v;
#pragma unroll
      for (int q = 0; q < 4; ++q) as[r * WGM_LDA + 4 * v + q] = (t + q < L) ? ra[i][q] : 0.f;  // (t < 0 loaded as 0)
    }
  };
  if (tbeg < tend) load(tbeg);
  for (int t0 = tbeg; t0 < tend; t0 += WGM_T) {
    __syncthreads();
    store(t0);
    __syncthreads();
    if (t0 + WGM_T < tend) load(t0 + WGM_T);  // in flight behind this chunk's MFMAs
    if (live) {
      const float* ap = ds + l31 * WGM_LDD + h;                      // A: dz[co = l31][t = t0 + 2 s + h]
      const float* bp = as + (wave * 32 + l31) * WGM_LDA + 3 + h;    // B: a_in[ci = l31][t + tap - 1]
#pragma unroll 4
      for (int sI = 0; sI < WGM_T / 2; ++sI) {
        const float av = ap[2 * sI];
        const float b0 = bp[2 * sI], b1 = bp[2 * sI + 1], b2 = bp[2 * sI + 2];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b2, acc[2], 0, 0, 0);
      }
    }
  }
  if (!live) return;
  const int ci = ci0 + wave * 32 + l31;
  if (ci >= Cin) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = co0 + (r & 3) + 8 * (r >> 2) + 4 * h;
    if (co < Cout) {
      float* o = part + (((size_t)blockIdx.z * Cout + co) * Cin + ci) * 3;
      o[0] = acc[0][r];
      o[1] = acc[1][r];
      o[2] = acc[2][r];
    }
  }
}

__global__ void train_reduce_b_kernel(const float* __restrict__ part, int B, size_t n, float* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  int b = 0;
  for (; b + 8 <= B; b += 8) {  // eight loads in flight, summed in the fixed order b = 0, 1, 2, ...
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = part[(size_t)(b + q) * n + i];
#pragma unroll
    for (int q = 0; q < 8; ++q) s += v[q];
  }
  for (; b < B; ++b) s += part[(size_t)b * n + i];
  out[i] = s;
}

// db[c] = sum_{b,t} dz[b][c][t]
__global__ void __launch_bounds__(256) train_bias_grad_kernel(const float* __restrict__ dz, int B, int C, int L, int ld,
                                                              float* __restrict__ db) {
  __shared__ double red[4];
  const int c = blockIdx.x;
  double s = 0.0;
  for (int b = 0; b < B; ++b)
    for (int t = threadIdx.x; t < L; t += 256) s += dz[((size_t)b * C + c) * ld + t];
  s = block_sum_d(s, red);
  if (threadIdx.x == 0) db[c] = (float)s;
}

// scalar head: out[b][t] = bias + sum_{ci,j} w[ci*K+j] * a[b][ci][t + j - pad]
// 64 positions x 4 channel quarters per workgroup; the quarters are summed in a fixed order.
__global__ void __launch_bounds__(256) train_head_fwd_kernel(const float* __restrict__ a, const float* __restrict__ w,
                                                             const float* __restrict__ bias, int C, int K, int L, int ld,
                                                             float* __restrict__ out, int ldo) {
  __shared__ float red[4][64];
  const int b = blockIdx.y;
  const int u = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + u;
  const int pad = (K - 1) / 2;
  const int cq = (C + 3) / 4, c0 = g * cq, c1 = (c0 + cq < C) ? c0 + cq : C;
  float s = 0.f;
  if (t < L)
    for (int ci = c0; ci < c1; ++ci) {
      const float* row = a + ((size_t)b * C + ci) * ld;
      for (int j = 0; j < K; ++j) {
        const int tt = t + j - pad;
        if (tt >= 0 && tt < L) s = fmaf(w[ci * K + j], row[tt], s);
      }
    }
  red[g][u] = s;
  __syncthreads();
  if (g == 0 && t < L) out[(size_t)b * ldo + t] = ((red[0][u] + red[1][u]) + (red[2][u] + red[3][u])) + bias[0];
}

// da[b][ci][t] (+)= sum_j w[ci*K+j] * dout[b][t - j + pad]
__global__ void train_head_bwd_data_kernel(const float* __restrict__ dout, const float* __restrict__ w, int C, int K,
                                           int L, int ld, int ldo, int accumulate, float* __restrict__ da) {
  const int b = blockIdx.z, ci = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= L) return;
  const int pad = (K - 1) / 2;
  float s = 0.f;
  for (int j = 0; j < K; ++j) {
    const int tt = t - j + pad;
    if (tt >= 0 && tt < L) s = fmaf(w[ci * K + j], dout[(size_t)b * ldo + tt], s);
  }
  const size_t i = ((size_t)b * C + ci) * ld + t;
  da[i] = accumulate ? da[i] + s : s;
}

// dw[ci*K+j] = sum_{b,t} dout[b][t] * a[b][ci][t + j - pad]; block C: db = sum dout
__global__ void __launch_bounds__(256) train_head_wgrad_kernel(const float* __restrict__ dout, const float* __restrict__ a,
                                                               int B, int C, int K, int L, int ld, int ldo,
                                                               float* __restrict__ dw, float* __restrict__ db) {
  __shared__ double red[4];
  const int ci = blockIdx.x;
  const int pad = (K - 1) / 2;
  if (ci == C) {
    double s = 0.0;
    for (int b = 0; b < B; ++b)
      for (int t = threadIdx.x; t < L; t += 256) s += dout[(size_t)b * ldo + t];
    s = block_sum_d(s, red);
    if (threadIdx.x == 0) db[0] = (float)s;
    return;
  }
  for (int j = 0; j < K; ++j) {
    double s = 0.0;
    for (int b = 0; b < B; ++b)
      for (int t = threadIdx.x; t < L; t += 256) {
        const int tt = t + j - pad;
        if (tt >= 0 && tt < L) s += (double)dout[(size_t)b * ldo + t] * (double)a[((size_t)b * C + ci) * ld + tt];
      }
    s = block_sum_d(s, red);
    if (threadIdx.x == 0) dw[ci * K + j] = (float)s;
  }
}

// LenSumLoss: masked squared error + 0.5 * squared sums over groups of four without padding; one block per row
__global__ void __launch_bounds__(256) train_len_loss_kernel(const float* __restrict__ out, const float* __restrict__ tgt,
                                                             int L, int ldo, float pad, float nmean, float nstd,
                                                             float* __restrict__ dout, double* __restrict__ loss_part) {
  __shared__ double red[4];
  const int b = blockIdx.x;
  double acc = 0.0;
  const int ngrp = L / 4;
  for (int t = threadIdx.x; t < L; t += 256) {
    const float tg = tgt[(size_t)b * L + t];
    const float diff = (out[(size_t)b * ldo + t] * nstd + nmean) - tg;
    float g = 0.f;
    if (tg != pad) {
      acc += (double)diff * diff;
      g = 2.f * diff;
    }
    const int grp = t >> 2;
    if (grp < ngrp) {
      bool anypad = false;
      float s = 0.f;
      for (int e = 0; e < 4; ++e) {
        const float te = tgt[(size_t)b * L + 4 * grp + e];
        anypad |= (te == pad);
        s += (out[(size_t)b * ldo + 4 * grp + e] * nstd + nmean) - te;
      }
      if (!anypad) {
        g += s;  // d/dp of 0.5 * s^2
        if ((t & 3) == 0) acc += 0.5 * (double)s * s;
      }
    }
    dout[(size_t)b * ldo + t] = g * nstd;
  }
  acc = block_sum_d(acc, red);
  if (threadIdx.x == 0) loss_part[b] = acc;
}

// PitchLoss: 100 * masked BCE-with-logits(cls, gts != 0) + masked, voiced-only L1 between de-normalised values
__global__ void __launch_bounds__(256) train_pitch_loss_kernel(const float* __restrict__ cls, const float* __restrict__ reg,
                                                               const float* __restrict__ tgt, const int64_t* __restrict__ spk,
                                                               const float* __restrict__ id2mean, const float* __restrict__ id2std,
                                                               int n_stats, int L, int ldo, float pad,
                                                               float* __restrict__ dcls, float* __restrict__ dreg,
                                                               double* __restrict__ loss_part) {
  __shared__ double red[4];
  const int b = blockIdx.x;
  long long s = spk[b];
  s = s < 0 ? 0 : (s >= n_stats ? n_stats - 1 : s);
  const float mean = id2mean[s], sd = id2std[s];
  double acc = 0.0;
  for (int t = threadIdx.x; t < L; t += 256) {
    const float g = tgt[(size_t)b * L + t];
    const float x = cls[(size_t)b * ldo + t], r = reg[(size_t)b * ldo + t];
    float gc = 0.f, gr = 0.f;
    if (g != pad) {
      const float y = g != 0.f ? 1.f : 0.f;
      const float bce = fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
      acc += 100.0 * (double)bce;
      gc = 100.f * (1.f / (1.f + expf(-x)) - y);
      if (g != 0.f) {
        const float d = (mean + sd * r) - (mean + sd * g);
        acc += (double)fabsf(d);
        gr = sd * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
      }
    }
    dcls[(size_t)b * ldo + t] = gc;
    dreg[(size_t)b * ldo + t] = gr;
  }
  acc = block_sum_d(acc, red);
  if (threadIdx.x == 0) loss_part[b] = acc;
}

__global__ void train_loss_final_kernel(const double* __restrict__ part, int B, float* __restrict__ loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int b = 0; b < B; ++b) s += part[b];
    loss[0] = (float)s;
  }
}

// token embedding gradient, one block per vocabulary row (fixed summation order); padding row: zero
__global__ void __launch_bounds__(256) train_tok_grad_kernel(const float* __restrict__ dx0, const int64_t* __restrict__ seq,
                                                             const float* __restrict__ keep, int B, int L, int ld, int E,
                                                             int pad_row, float* __restrict__ dtok) {
  // one workgroup per token row: the threads scan the batch position-parallel (coalesced), each keeps its own 32
  // channel sums, then a fixed-order tree over the 256 threads -- deterministic, no atomics
  __shared__ float red[256][33];
  const int v = blockIdx.x, tid = threadIdx.x;
  float acc[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) acc[c] = 0.f;
  if (v != pad_row)
    for (int p = tid; p < B * L; p += 256) {
      if (seq[p] != v) continue;
      const int b = p / L, t = p - b * L;
      const float k = keep ? keep[p] : 1.f;
      const float* src = dx0 + (size_t)b * 2 * E * ld + t;
#pragma unroll
      for (int c = 0; c < 32; ++c) acc[c] += k * src[(size_t)c * ld];
    }
#pragma unroll
  for (int c = 0; c < 32; ++c) red[tid][c] = acc[c];
  __syncthreads();
  const int c = tid & 31, part = tid >> 5;  // 8 partial sums of 32 rows each per channel
  float s = 0.f;
  for (int i = 0; i < 32; ++i) s += red[part * 32 + i][c];
  __syncthreads();
  red[part][c] = s;
  __syncthreads();
  if (tid < 32) {
    float tot = 0.f;
    for (int i = 0; i < 8; ++i) tot += red[i][tid];
    dtok[(size_t)v * E + tid] = tot;
  }
}

// speaker embedding gradient, one block per speaker row
__global__ void __launch_bounds__(256) train_spk_grad_kernel(const float* __restrict__ dx0, const int64_t* __restrict__ spk,
                                                             const float* __restrict__ pe_mult, int B, int L, int ld,
                                                             int E, int pad_row, float* __restrict__ dspk) {
  __shared__ float red[8][32];
  const int v = blockIdx.x;
  const int c = threadIdx.x & 31, ln = threadIdx.x >> 5;
  float s = 0.f;
  if (v != pad_row)
    for (int b = 0; b < B; ++b) {
      if (spk[b] != v) continue;
      for (int t = ln; t < L; t += 8) {
        const float m = pe_mult ? pe_mult[((size_t)b * L + t) * E + c] : 1.f;
        s += m * dx0[((size_t)b * 2 * E + E + c) * ld + t];
      }
    }
  red[ln][c] = s;
  __syncthreads();
  if (ln == 0) {
    float tot = 0.f;
    for (int i = 0; i < 8; ++i) tot += red[i][c];
    dspk[(size_t)v * E + c] = tot;
  }
}

// torch.optim.Adam (amsgrad False, weight_decay 0) over the flat trainable region
__global__ void train_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                  float* __restrict__ v, size_t n, float b1, float b2, float eps, float step_size,
                                  float bc2_sqrt) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  const float mi = m[i] + (1.f - b1) * (gi - m[i]);  // exp_avg.lerp_(grad, 1 - beta1)
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = p[i] - step_size * (mi / denom);
}

// master weights [Cout][Cin][K] -> A-fragment order of conv_mfma32_kernel (pack_conv_weights32); transposed = the
// backward-data conv: rows = input channels, columns = output channels, taps flipped
__global__ void train_repack32_kernel(const float* __restrict__ w, int Cout, int Cin, int K, int nsub, int nchunk,
                                      int transposed, float* __restrict__ packed) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)nsub * nchunk * K * 2 * 64 * 4;
  if (idx >= total) return;
  const int e = idx & 3, lane = (idx >> 2) & 63, hf = (idx >> 8) & 1;
  size_t r = idx >> 9;
  const int j = r % K;
  r /= K;
  const int c = r % nchunk, ms = r / nchunk;
  const int row = ms * 32 + (lane & 31);
  const int col = c * 16 + 2 * (4 * hf + e) + (lane >> 5);
  float v = 0.f;
  if (!transposed) {
    if (row < Cout && col < Cin) v = w[((size_t)row * Cin + col) * K + j];
  } else {
    if (row < Cin && col < Cout) v = w[((size_t)col * Cin + row) * K + (K - 1 - j)];
  }
  packed[idx] = v;
}

}  // namespace dissc

using namespace dissc;

namespace {

struct TParam {
  std::string name;
  size_t off = 0, numel = 0;
  bool trainable = true;
};

struct TLayer {
  std::string conv, bn;
  int cin = 0, cout = 0, k = 3;
  bool has_bn = false, leaky = true;
  int in = -1;               // producing layer (-1: the embedding)
  int w = -1, b = -1, g = -1, be = -1, rm = -1, rv = -1;  // parameter indices
  DevConv fwd, bwd;          // wide layers only
  // workspace views (set per step)
  float *z = nullptr, *a = nullptr, *da = nullptr, *dz = nullptr, *mean = nullptr, *invstd = nullptr;
};

}  // namespace

struct dissc_trainer {
  Options opt;  // this handle's snapshot of the tuning options (common.h)
  int kind = 0, E = 32, n_tok = 0, n_spk = 0, pe_len = 0;
  std::vector<TParam> params;      // trainable ones first (flat region [0, n_train))
  std::map<std::string, int> index;
  size_t n_train = 0, n_total = 0;
  float *P = nullptr, *G = nullptr, *M = nullptr, *V = nullptr;
  std::vector<TLayer> layers;      // trunk, then the heads' hidden layers
  std::vector<int> heads;          // scalar heads: layer indices (cout == 1)
  int tok = -1, spe = -1, pe = -1;
  float nmean = 0.f, nstd = 1.f;
  float *id2mean = nullptr, *id2std = nullptr;
  int n_stats = 0;
  long long step = 0;
  ~dissc_trainer() {
    for (float* p : {P, G, M, V, id2mean, id2std})
      if (p) (void)hipFree(p);
    for (auto& l : layers) {
      free_conv(l.fwd);
      free_conv(l.bwd);
    }
  }
};

static inline size_t t_rup(size_t x, size_t m) { return (x + m - 1) / m * m; }

static void t_add_layer(dissc_trainer* t, const char* conv, const char* bn, int cin, int cout, int k, bool leaky, int in) {
  TLayer l;
  l.conv = conv;
  l.bn = bn ? bn : "";
  l.cin = cin; l.cout = cout; l.k = k; l.has_bn = bn != nullptr; l.leaky = leaky; l.in = in;
  t->layers.push_back(l);
}

extern "C" {

int dissc_train_create(int kind, const DisscTensor* tensors, size_t n, dissc_trainer_t* out) {
  if (kind < 0 || kind > 2 || !tensors || !out) {
    set_error("dissc_train_create: bad argument");
    return DISSC_EINVAL;
  }
  dissc_trainer* t = new dissc_trainer();
  t->opt = g_defaults;  // frozen here
  OptScope opt_scope(&t->opt);
  t->kind = kind;
  auto fail = [&](int rc) { delete t; return rc; };
  // ---- model description (reference model/len_predictor.py:13-33, model/pitch_predictor.py:52-70,117-143) ----
  char nm[32], bnm[32];
  if (kind == 0) {
    t_add_layer(t, "cnn1", "bn1", 64, 128, 3, true, -1);
    for (int i = 1; i <= 6; ++i) {
      snprintf(nm, sizeof(nm), "cnn1%d", i);
      snprintf(bnm, sizeof(bnm), "bn1%d", i);
      t_add_layer(t, nm, bnm, 128, 128, 3, true, i - 1);
    }
    t_add_layer(t, "cnn2", nullptr, 128, 1, 3, false, 6);
    t->heads = {7};
  } else {
    const bool base = kind == 2;
    t_add_layer(t, "cnn1", base ? "bn1" : nullptr, 64, 128, 3, true, -1);
    for (int i = 1; i <= 7; ++i) {
      snprintf(nm, sizeof(nm), "cnn1%d", i);
      snprintf(bnm, sizeof(bnm), "bn1%d", i);
      t_add_layer(t, nm, base ? bnm : nullptr, 128, 128, 3, true, i - 1);
    }
    t_add_layer(t, "cnn2", base ? nullptr : "bn2", 128, 128, 3, true, 7);          // 8
    t_add_layer(t, "cnn_class1", base ? "bn_c1" : nullptr, 128, 128, 3, true, 8);  // 9
    t_add_layer(t, "cnn_reg1", base ? "bn_r1" : nullptr, 128, 128, 3, true, 8);    // 10
    t_add_layer(t, "cnn_class2", nullptr, 128, 1, 1, false, 9);                    // 11
    t_add_layer(t, "cnn_reg2", nullptr, 128, 1, 1, false, 10);                     // 12
    t->heads = {11, 12};
  }
  // ---- parameters: trainable first ----
  std::map<std::string, const DisscTensor*> byname;
  for (size_t i = 0; i < n; ++i) byname[tensors[i].name] = &tensors[i];
  auto numel_of = [](const DisscTensor* d) {
    size_t k = 1;
    for (int i = 0; i < d->ndim; ++i) k *= (size_t)d->shape[i];
    return k;
  };
  std::vector<std::pair<std::string, bool>> order = {{"token_emb.weight", true}, {"spk_emb.weight", true}};
  for (auto& l : t->layers) {
    order.push_back({l.conv + ".weight", true});
    order.push_back({l.conv + ".bias", true});
    if (l.has_bn) {
      order.push_back({l.bn + ".weight", true});
      order.push_back({l.bn + ".bias", true});
    }
  }
  for (auto& l : t->layers)
    if (l.has_bn) {
      order.push_back({l.bn + ".running_mean", false});
      order.push_back({l.bn + ".running_var", false});
    }
  if (kind == 1) order.push_back({"pe.pe", false});
  size_t off = 0;
  for (auto& o : order) {
    auto it = byname.find(o.first);
    if (it == byname.end()) {
      set_error("dissc_train_create: missing tensor '%s'", o.first.c_str());
      return fail(DISSC_ENOTFOUND);
    }
    TParam p;
    p.name = o.first; p.off = off; p.numel = numel_of(it->second); p.trainable = o.second;
    t->index[p.name] = (int)t->params.size();
    t->params.push_back(p);
    off += t_rup(p.numel, 4);
    if (o.second) t->n_train = off;
  }
  t->n_total = off;
  std::vector<float> host(off, 0.f);
  for (auto& p : t->params) memcpy(host.data() + p.off, byname[p.name]->data, p.numel * sizeof(float));
  int rc;
  if ((rc = upload(host, &t->P))) return fail(rc);
  std::vector<float> zeros(t->n_train, 0.f);
  if ((rc = upload(zeros, &t->G)) || (rc = upload(zeros, &t->M)) || (rc = upload(zeros, &t->V))) return fail(rc);
  const DisscTensor* tk = byname["token_emb.weight"];
  const DisscTensor* sp = byname["spk_emb.weight"];
  t->n_tok = (int)tk->shape[0];
  t->n_spk = (int)sp->shape[0];
  t->E = (int)tk->shape[1];
  if (t->E != 32 || sp->shape[1] != 32) {
    set_error("dissc_train_create: embedding size %d unsupported (32)", t->E);
    return fail(DISSC_EINVAL);
  }
  t->tok = t->index["token_emb.weight"];
  t->spe = t->index["spk_emb.weight"];
  if (kind == 1) {
    t->pe = t->index["pe.pe"];
    t->pe_len = (int)byname["pe.pe"]->shape[1];
  }
  for (auto& l : t->layers) {
    l.w = t->index[l.conv + ".weight"];
    l.b = t->index[l.conv + ".bias"];
    if (l.has_bn) {
      l.g = t->index[l.bn + ".weight"]; l.be = t->index[l.bn + ".bias"];
      l.rm = t->index[l.bn + ".running_mean"]; l.rv = t->index[l.bn + ".running_var"];
    }
    const DisscTensor* w = byname[l.conv + ".weight"];
    if (w->ndim != 3 || w->shape[0] != l.cout || w->shape[1] != l.cin || w->shape[2] != l.k) {
      set_error("dissc_train_create: tensor '%s.weight' has the wrong shape", l.conv.c_str());
      return fail(DISSC_EINVAL);
    }
    if (l.cout >= 32) {  // wide: matrix-core convs, forward and backward-data
      if ((rc = make_conv(w->data, byname[l.conv + ".bias"]->data, l.cout, l.cin, l.k, 1, l.fwd))) return fail(rc);
      std::vector<float> wt((size_t)l.cin * l.cout * l.k);
      for (int co = 0; co < l.cout; ++co)
        for (int ci = 0; ci < l.cin; ++ci)
          for (int j = 0; j < l.k; ++j)
            wt[((size_t)ci * l.cout + co) * l.k + j] = w->data[((size_t)co * l.cin + ci) * l.k + (l.k - 1 - j)];
      if ((rc = make_conv(wt.data(), nullptr, l.cin, l.cout, l.k, 1, l.bwd))) return fail(rc);
      if (!l.fwd.m32 || !l.bwd.m32 || l.fwd.prec || l.bwd.prec) {
        set_error("dissc_train_create: unexpected conv packing");
        return fail(DISSC_EINVAL);
      }
    }
  }
  *out = t;
  return DISSC_OK;
}

void dissc_train_destroy(dissc_trainer_t t) { delete t; }

int dissc_train_set_len_norm(dissc_trainer_t t, float mean, float std) {
  if (!t || t->kind != 0) return DISSC_EINVAL;
  t->nmean = mean;
  t->nstd = std;
  return DISSC_OK;
}

int dissc_train_set_pitch_stats(dissc_trainer_t t, const float* id2mean, const float* id2std, int n) {
  if (!t || t->kind == 0 || !id2mean || !id2std || n <= 0) return DISSC_EINVAL;
  if (t->id2mean) (void)hipFree(t->id2mean);
  if (t->id2std) (void)hipFree(t->id2std);
  t->id2mean = t->id2std = nullptr;
  int rc;
  if ((rc = upload(std::vector<float>(id2mean, id2mean + n), &t->id2mean))) return rc;
  if ((rc = upload(std::vector<float>(id2std, id2std + n), &t->id2std))) return rc;
  t->n_stats = n;
  return DISSC_OK;
}

int dissc_train_num_tensors(dissc_trainer_t t) { return t ? (int)t->params.size() : 0; }
const char* dissc_train_tensor_name(dissc_trainer_t t, int i) {
  return (t && i >= 0 && i < (int)t->params.size()) ? t->params[i].name.c_str() : nullptr;
}
long long dissc_train_tensor_numel(dissc_trainer_t t, int i) {
  return (t && i >= 0 && i < (int)t->params.size()) ? (long long)t->params[i].numel : 0;
}
long long dissc_train_steps(dissc_trainer_t t) { return t ? t->step : 0; }

// which: 0 = value, 1 = gradient of the last step (trainable tensors only); synchronous copy to the host
int dissc_train_read(dissc_trainer_t t, int i, int which, float* host_out, void* stream) {
  if (!t || i < 0 || i >= (int)t->params.size() || !host_out || which < 0 || which > 1 ||
      (which == 1 && !t->params[i].trainable)) {
    set_error("dissc_train_read: bad argument");
    return DISSC_EINVAL;
  }
  DISSC_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  const float* src = (which == 0 ? t->P : t->G) + t->params[i].off;
  DISSC_HIP_CHECK(hipMemcpy(host_out, src, t->params[i].numel * sizeof(float), hipMemcpyDeviceToHost));
  return DISSC_OK;
}

// Diagnostics: copy an activation buffer of the LAST step out of the caller's workspace (valid until the next step).
// which: 0 z (conv output), 1 a (after BatchNorm / LeakyReLU), 2 da, 3 dz; layer -1: x0 (which 0) / dx0 (which 2).
static float *g_dbg_x0 = nullptr, *g_dbg_dx0 = nullptr;
int dissc_train_debug_read(dissc_trainer_t t, int layer, int which, float* host_out, size_t n, void* stream) {
  if (!t || !host_out || layer < -1 || layer >= (int)t->layers.size() || which < 0 || which > 3) return DISSC_EINVAL;
  const float* src = nullptr;
  if (layer < 0) src = which == 0 ? g_dbg_x0 : g_dbg_dx0;
  else {
    TLayer& l = t->layers[layer];
    src = which == 0 ? l.z : which == 1 ? l.a : which == 2 ? l.da : l.dz;
  }
  if (!src) return DISSC_EINVAL;
  DISSC_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  DISSC_HIP_CHECK(hipMemcpy(host_out, src, n * sizeof(float), hipMemcpyDeviceToHost));
  return DISSC_OK;
}

static size_t t_act_floats(int B, int C, int ld) { return t_rup((size_t)B * C * ld, 64); }

size_t dissc_train_workspace_bytes(dissc_trainer_t t, int B, int L) {
  if (!t || B <= 0 || L <= 0) return 0;
  const int ld = (int)t_rup(L, 4);
  size_t f = t_act_floats(B, 64, ld) * 2;  // x0, dx0
  size_t wmax = 0;
  for (auto& l : t->layers) {
    f += 4 * t_act_floats(B, l.cout, ld) + 2 * t_rup(l.cout, 64);
    wmax = std::max(wmax, (size_t)l.cout * l.cin * l.k);
  }
  f += t_rup((size_t)B * WGM_SPLIT * wmax, 64);  // per-utterance (x time half) weight-gradient partials
  f += 2 * t_rup((size_t)B, 64) * 2;       // loss partials (double)
  return f * sizeof(float) + 1024;
}

int dissc_train_step(dissc_trainer_t t, const int64_t* seq, const int64_t* spk, const float* target, const float* keep,
                     const float* pe_mult, int B, int L, float pad_value, float lr, float* loss_out, void* workspace,
                     size_t ws_bytes, void* stream_) {
  if (!t || !seq || !spk || !target || !loss_out || !workspace || B <= 0 || L <= 0 || !(lr > 0.f)) {
    set_error("dissc_train_step: bad argument");
    return DISSC_EINVAL;
  }
  OptScope opt_scope(&t->opt);
  if (t->kind != 0 && (!t->id2mean || !t->id2std)) {
    set_error("dissc_train_step: dissc_train_set_pitch_stats first");
    return DISSC_EINVAL;
  }
  if (t->kind == 1 && L > t->pe_len) {
    set_error("dissc_train_step: %d frames exceed the positional encoding (%d)", L, t->pe_len);
    return DISSC_EINVAL;
  }
  if (ws_bytes < dissc_train_workspace_bytes(t, B, L)) {
    set_error("dissc_train_step: workspace %zu < %zu bytes", ws_bytes, dissc_train_workspace_bytes(t, B, L));
    return DISSC_ENOMEM;
  }
  hipStream_t st = (hipStream_t)stream_;
  const int ld = (int)t_rup(L, 4), E = t->E;
  float* p = (float*)t_rup((size_t)workspace, 256);
  auto take = [&](size_t nfl) {
    float* r = p;
    p += nfl;
    return r;
  };
  float* x0 = take(t_act_floats(B, 64, ld));
  float* dx0 = take(t_act_floats(B, 64, ld));
  g_dbg_x0 = x0;
  g_dbg_dx0 = dx0;
  size_t wmax = 0;
  for (auto& l : t->layers) {
    l.z = take(t_act_floats(B, l.cout, ld));
    l.a = take(t_act_floats(B, l.cout, ld));
    l.da = take(t_act_floats(B, l.cout, ld));
    l.dz = take(t_act_floats(B, l.cout, ld));
    l.mean = take(t_rup(l.cout, 64));
    l.invstd = take(t_rup(l.cout, 64));
    wmax = std::max(wmax, (size_t)l.cout * l.cin * l.k);
  }
  float* part = take(t_rup((size_t)B * WGM_SPLIT * wmax, 64));
  double* loss_part = (double*)take(2 * t_rup((size_t)B, 64));
  auto P = [&](int i) { return t->P + t->params[i].off; };
  auto G = [&](int i) { return t->G + t->params[i].off; };
  const dim3 blk(128);
  auto grid3 = [&](int C) { return dim3((L + 127) / 128, C, B); };
  int rc;

  // ---------------- forward ----------------
  hipLaunchKernelGGL(train_embed_kernel, grid3(2 * E), blk, 0, st, seq, spk, keep, pe_mult, P(t->tok), P(t->spe),
                     t->pe >= 0 ? P(t->pe) : (const float*)nullptr, L, E, t->n_tok, t->n_spk, x0, ld);
  for (size_t li = 0; li < t->layers.size(); ++li) {
    TLayer& l = t->layers[li];
    const float* in = l.in < 0 ? x0 : t->layers[l.in].a;
    if (l.cout >= 32) {
      if ((rc = run_conv(l.fwd, in, l.z, nullptr, nullptr, nullptr, L, 1, B, l.cin, ld, ld, L, 1.0f, EPI_STORE, 1.f, st)))
        return rc;
      if (l.has_bn)
        hipLaunchKernelGGL(train_bn_stats_kernel, dim3(l.cout), dim3(256), 0, st, l.z, B, l.cout, L, ld, l.mean,
                           l.invstd, P(l.rm), P(l.rv));
      hipLaunchKernelGGL(train_act_kernel, grid3(l.cout), blk, 0, st, l.z, l.mean, l.invstd,
                         l.has_bn ? P(l.g) : (const float*)nullptr, l.has_bn ? P(l.be) : (const float*)nullptr, l.cout,
                         L, ld, l.has_bn ? 1 : 0, l.leaky ? 1 : 0, l.a);
    } else {  // scalar head: z = a = [B][ld]
      hipLaunchKernelGGL(train_head_fwd_kernel, dim3((L + 63) / 64, B), dim3(256), 0, st, in, P(l.w), P(l.b), l.cin, l.k, L,
                         ld, l.z, ld);
    }
  }
  // ---------------- loss and d(loss)/d(head outputs) ----------------
  if (t->kind == 0) {
    TLayer& h = t->layers[t->heads[0]];
    hipLaunchKernelGGL(train_len_loss_kernel, dim3(B), dim3(256), 0, st, h.z, target, L, ld, pad_value, t->nmean,
                       t->nstd, h.dz, loss_part);
  } else {
    TLayer& hc = t->layers[t->heads[0]];
    TLayer& hr = t->layers[t->heads[1]];
    hipLaunchKernelGGL(train_pitch_loss_kernel, dim3(B), dim3(256), 0, st, hc.z, hr.z, target, spk, t->id2mean,
                       t->id2std, t->n_stats, L, ld, pad_value, hc.dz, hr.dz, loss_part);
  }
  hipLaunchKernelGGL(train_loss_final_kernel, dim3(1), dim3(64), 0, st, loss_part, B, loss_out);
  // ---------------- backward (layers in reverse; a layer's da is complete when it is reached) ----------------
  std::vector<char> has_da(t->layers.size(), 0);
  bool dx0_set = false;
  for (int li = (int)t->layers.size() - 1; li >= 0; --li) {
    TLayer& l = t->layers[li];
    const float* in = l.in < 0 ? x0 : t->layers[l.in].a;
    float* din = l.in < 0 ? dx0 : t->layers[l.in].da;
    const bool acc_in = l.in < 0 ? dx0_set : (has_da[l.in] != 0);
    if (l.cout < 32) {  // scalar head: dz was written by the loss kernel
      hipLaunchKernelGGL(train_head_wgrad_kernel, dim3(l.cin + 1), dim3(256), 0, st, l.dz, in, B, l.cin, l.k, L, ld, ld,
                         G(l.w), G(l.b));
      hipLaunchKernelGGL(train_head_bwd_data_kernel, grid3(l.cin), blk, 0, st, l.dz, P(l.w), l.cin, l.k, L, ld, ld,
                         acc_in ? 1 : 0, din);
    } else {
      if (l.has_bn)
        hipLaunchKernelGGL(train_bn_bwd_reduce_kernel, dim3(l.cout), dim3(256), 0, st, l.da, l.a, l.z, l.mean, l.invstd,
                           B, l.cout, L, ld, l.leaky ? 1 : 0, G(l.g), G(l.be));
      hipLaunchKernelGGL(train_bn_bwd_apply_kernel, grid3(l.cout), blk, 0, st, l.da, l.a, l.z, l.mean, l.invstd,
                         l.has_bn ? P(l.g) : (const float*)nullptr, l.has_bn ? G(l.g) : (const float*)nullptr,
                         l.has_bn ? G(l.be) : (const float*)nullptr, B, l.cout, L, ld, l.has_bn ? 1 : 0, l.leaky ? 1 : 0,
                         l.dz);
      // weight / bias gradients
      if (l.k == 3)
        hipLaunchKernelGGL(train_wgrad_mfma_kernel, dim3((l.cout + 31) / 32, (l.cin + 127) / 128, B * WGM_SPLIT),
                           dim3(256), 0, st, l.dz, in, l.cout, l.cin, L, ld, ld, part);
      else
        hipLaunchKernelGGL(train_wgrad_kernel, dim3((l.cout + 15) / 16, (l.cin + 15) / 16, B), dim3(256), 0, st, l.dz, in,
                           l.cout, l.cin, l.k, L, ld, ld, part);
      const size_t nw = (size_t)l.cout * l.cin * l.k;
      hipLaunchKernelGGL(train_reduce_b_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, part,
                         l.k == 3 ? B * WGM_SPLIT : B, nw, G(l.w));
      hipLaunchKernelGGL(train_bias_grad_kernel, dim3(l.cout), dim3(256), 0, st, l.dz, B, l.cout, L, ld, G(l.b));
      // input gradient: the same conv with W^T, taps flipped (accumulating where the input feeds two layers)
      if ((rc = run_conv(l.bwd, l.dz, din, acc_in ? din : nullptr, nullptr, nullptr, L, 1, B, l.cout, ld, ld, L, 1.0f,
                         acc_in ? EPI_RES : EPI_STORE, 1.f, st)))
        return rc;
    }
    if (l.in < 0) dx0_set = true; else has_da[l.in] = 1;
  }
  hipLaunchKernelGGL(train_tok_grad_kernel, dim3(t->n_tok), dim3(256), 0, st, dx0, seq, keep, B, L, ld, E,
                     t->n_tok - 1, G(t->tok));
  hipLaunchKernelGGL(train_spk_grad_kernel, dim3(t->n_spk), dim3(256), 0, st, dx0, spk, pe_mult, B, L, ld, E,
                     t->kind == 0 ? -1 : t->n_spk - 1, G(t->spe));
  // ---------------- Adam, then rebuild the packed weights ----------------
  t->step += 1;
  const double b1 = 0.9, b2 = 0.999;
  const double bc1 = 1.0 - pow(b1, (double)t->step), bc2 = 1.0 - pow(b2, (double)t->step);
  hipLaunchKernelGGL(train_adam_kernel, dim3((unsigned)((t->n_train + 255) / 256)), dim3(256), 0, st, t->P, t->G, t->M,
                     t->V, t->n_train, (float)b1, (float)b2, 1e-8f, (float)((double)lr / bc1), (float)sqrt(bc2));
  for (auto& l : t->layers) {
    if (l.cout < 32) continue;
    const size_t nf = (size_t)(l.fwd.Mpad / 32) * l.fwd.nchunk * l.k * 512;
    hipLaunchKernelGGL(train_repack32_kernel, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, st, P(l.w), l.cout, l.cin,
                       l.k, l.fwd.Mpad / 32, l.fwd.nchunk, 0, l.fwd.wpack);
    DISSC_HIP_CHECK(hipMemcpyAsync(l.fwd.bias, P(l.b), l.cout * sizeof(float), hipMemcpyDeviceToDevice, st));
    const size_t nb = (size_t)(l.bwd.Mpad / 32) * l.bwd.nchunk * l.k * 512;
    hipLaunchKernelGGL(train_repack32_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, st, P(l.w), l.cout, l.cin,
                       l.k, l.bwd.Mpad / 32, l.bwd.nchunk, 1, l.bwd.wpack);
  }
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

}  // extern "C"
