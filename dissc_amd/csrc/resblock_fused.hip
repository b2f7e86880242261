// Fused ResBlock1 for the narrow stages of the generator (C = 16 / 32 channels).
//
//   x_k = x;  3 x { x_k = x_k + conv_1(lrelu(conv_d(lrelu(x_k)))) }, d = 1, 3, 5     (k taps each)
//   then the MRF update of the stage accumulator:  acc = r | acc + r | (acc + r) / 3
// (reference sr/models.py:34-41 and :103-109).
//
// Unfused, these stages move 45 activation passes per stage through HBM with only 12-44
// FLOP/byte.  Here one workgroup keeps a [C][BN + 2*halo] window of the running
// value (A) and of the intermediate (T) in LDS, runs the six convs on the fp32 matrix pipe
// (v_mfma_f32_16x16x4_f32, weights streamed from L2 in MFMA-fragment order, the halo
// recomputed) and touches HBM twice: one read of x, one read-modify-write of the accumulator.
// The reference's per-layer zero padding is reproduced by forcing every intermediate to 0
// outside the utterance's [0, len).
#include <algorithm>

#include "common.h"

namespace dissc {

struct ResblockArgs {
  const float* x;        // [B][C][ld]   stage input
  float* acc;            // [B][C][ld]   MRF accumulator (also the output for MRF_SET)
  const float* wpack;    // 6 packed convs, in execution order (c1_0, c2_0, c1_1, c2_1, c1_2, c2_2)
  const float* bias;     // [6][C]
  const int32_t* lengths;
  int len_default, len_mul;
  int KS, dil[3];
  int BN, XW, ld;
  long long bstride;
  float slope, mrf_div;
  int epi;               // EPI_MRF_SET / EPI_MRF_ADD / EPI_MRF_DIV
};

template <int C, int NW, int NIMAX>
__global__ void __launch_bounds__(64 * NW) resblock_fused_kernel(const ResblockArgs a) {
  constexpr int MI = C / 16;   // 16-row subtiles
  constexpr int NCH = C / 16;  // 16-channel chunks
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int XW = a.XW;
  float* A = lds;            // running x_k      [C][XW]
  float* T = lds + C * XW;   // lrelu(conv1 out) [C][XW]

  const int b = blockIdx.y;
  const int len = a.lengths ? a.lengths[b] * a.len_mul : a.len_default;
  const int t0 = blockIdx.x * a.BN;
  if (t0 >= len) return;
  const int KS = a.KS;
  const int p2 = (KS - 1) >> 1;
  const int H = p2 * (a.dil[0] + a.dil[1] + a.dil[2] + 3);  // total halo of the block
  const int tb = (t0 - H) & ~3;
  const int sh = t0 - H - tb;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const float slope = a.slope;
  const float* xb = a.x + (size_t)b * a.bstride;

  // ---- load the window: A[c][u] = x[c][tb + u] inside the utterance, else 0 ----
  {
    const int NV = XW >> 2;
    for (int e = tid; e < C * NV; e += 64 * NW) {
      const int r = e / NV, v = e - r * NV;
      int t = tb + 4 * v;
      const int tc = t < 0 ? 0 : (t > a.ld - 4 ? a.ld - 4 : t);
      f32x4 val = *reinterpret_cast<const f32x4*>(xb + (size_t)r * a.ld + tc);
#pragma unroll
      for (int k = 0; k < 4; ++k) val[k] = (t + k >= 0 && t + k < len) ? val[k] : 0.f;
      *reinterpret_cast<f32x4*>(A + r * XW + 4 * v) = val;
    }
  }
  __syncthreads();

  int lo = sh, hi = sh + a.BN + 2 * H;  // valid columns of A
  const f32x4* wl = reinterpret_cast<const f32x4*>(a.wpack);
  const int nq = NCH * KS;              // (chunk, tap) steps per conv

#pragma unroll 1
  for (int layer = 0; layer < 6; ++layer) {
    const bool first = (layer & 1) == 0;  // conv1 of a pair: dilated, reads A (lrelu on the fly), writes T
    const int d = first ? a.dil[layer >> 1] : 1;
    const int p = p2 * d;
    const float* in = first ? A : T;
    float* out = first ? T : A;
    const float sl = first ? slope : 1.0f;  // branch-free input activation
    lo += p;
    hi -= p;
    // every wave owns NIMAX consecutive 16-column tiles of this layer's output range; the host
    // sized BN so that NW*NIMAX tiles cover the widest layer (no per-tile branches in the loop)
    const int ucol = lo + wave * (NIMAX * 16);
    const float* bl = a.bias + layer * C;
    if (ucol < hi) {
      f32x4 acc[MI][NIMAX];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NIMAX; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
      const f32x4* wp[MI];
      f32x4 av[MI], avn[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        wp[mi] = wl + (size_t)mi * nq * 64 + lane;
        av[mi] = wp[mi][0];
      }
      const float* bcol = in + g * XW + ucol - p + l15;
      const int XW4 = 4 * XW;
      int q = 0;
#pragma unroll 1
      for (int ch = 0; ch < NCH; ++ch) {
        const float* bj = bcol + ch * 16 * XW;
        float b0[NIMAX], b1[NIMAX], b2[NIMAX], b3[NIMAX], b0n[NIMAX];
#pragma unroll
        for (int ni = 0; ni < NIMAX; ++ni) b0[ni] = bj[ni * 16];
#pragma unroll 1
        for (int j = 0; j < KS; ++j, ++q) {
          const int qn = (q + 1 < nq) ? q + 1 : q;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) avn[mi] = wp[mi][(size_t)qn * 64];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ni = 0; ni < NIMAX; ++ni) {
            b1[ni] = bj[XW4 + ni * 16];
            b2[ni] = bj[2 * XW4 + ni * 16];
            b3[ni] = bj[3 * XW4 + ni * 16];
          }
#define DISSC_RB_STEP(CQ, BV)                                                                  \
  _Pragma("unroll") for (int ni = 0; ni < NIMAX; ++ni) {                                         \
    const float bx = BV[ni] > 0.f ? BV[ni] : BV[ni] * sl;                                        \
    _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                            \
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi][CQ], bx, acc[mi][ni], 0, 0, 0); \
  }
          DISSC_RB_STEP(0, b0)
          bj += d;
#pragma unroll
          for (int ni = 0; ni < NIMAX; ++ni) b0n[ni] = bj[ni * 16];
          DISSC_RB_STEP(1, b1)
          DISSC_RB_STEP(2, b2)
          DISSC_RB_STEP(3, b3)
#undef DISSC_RB_STEP
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ni = 0; ni < NIMAX; ++ni) b0[ni] = b0n[ni];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) av[mi] = avn[mi];
        }
      }
      // epilogue: D layout col = lane&15 (time), row = 4*(lane>>4) + r
#pragma unroll
      for (int ni = 0; ni < NIMAX; ++ni) {
        const int u = ucol + ni * 16 + l15;
        const int t = tb + u;
        const bool keep = u < hi;
        const bool inside = t >= 0 && t < len;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = mi * 16 + 4 * g + r;
            float v = acc[mi][ni][r] + bl[row];
            if (first) {
              v = v > 0.f ? v : v * slope;  // T feeds conv2 only: store it already activated
            } else if (keep) {
              v += out[row * XW + u];       // x_k = xt + x_k (same element, in place)
            }
            if (keep) out[row * XW + u] = inside ? v : 0.f;
          }
      }
    }
    wl += (size_t)MI * nq * 64;
    __syncthreads();
  }

  // ---- MRF update: centre [t0, t0+BN) of A -> acc, 16 B per lane ----
  {
    const int u0 = sh + H;  // == t0 - tb, a multiple of 4
    const int nv = a.BN >> 2;
    float* ab = a.acc + (size_t)b * a.bstride;
    for (int e = tid; e < C * nv; e += 64 * NW) {
      const int r = e / nv, v4 = e - r * nv;
      const int t = t0 + 4 * v4;
      if (t >= len) continue;
      f32x4 v = *reinterpret_cast<const f32x4*>(A + r * XW + u0 + 4 * v4);
      float* dst = ab + (size_t)r * a.ld + t;
      const int nval = len - t;
      if (nval >= 4) {
        if (a.epi != EPI_MRF_SET) {
          const f32x4 o = *reinterpret_cast<const f32x4*>(dst);
          v[0] = o[0] + v[0]; v[1] = o[1] + v[1]; v[2] = o[2] + v[2]; v[3] = o[3] + v[3];
          if (a.epi == EPI_MRF_DIV) {
            v[0] = __fdiv_rn(v[0], a.mrf_div); v[1] = __fdiv_rn(v[1], a.mrf_div);
            v[2] = __fdiv_rn(v[2], a.mrf_div); v[3] = __fdiv_rn(v[3], a.mrf_div);
          }
        }
        *reinterpret_cast<f32x4*>(dst) = v;
      } else {
        for (int k = 0; k < nval; ++k) {
          float x = v[k];
          if (a.epi != EPI_MRF_SET) {
            x = dst[k] + x;
            if (a.epi == EPI_MRF_DIV) x = __fdiv_rn(x, a.mrf_div);
          }
          dst[k] = x;
        }
      }
    }
  }
}

// ---- C = 16 specialisation: all taps of a layer's weights live in registers (KS float4 per
// lane), the next layer's set is prefetched behind the current layer's MFMAs, the tap loop is
// fully unrolled (no global load and no wait inside a layer).
template <int KS, int NW, int NI>
__global__ void __launch_bounds__(64 * NW) resblock16_kernel(const ResblockArgs a) {
  constexpr int C = 16;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int XW = a.XW;
  float* A = lds;
  float* T = lds + C * XW;
  const int b = blockIdx.y;
  const int len = a.lengths ? a.lengths[b] * a.len_mul : a.len_default;
  const int t0 = blockIdx.x * a.BN;
  if (t0 >= len) return;
  constexpr int p2 = (KS - 1) >> 1;
  const int H = p2 * (a.dil[0] + a.dil[1] + a.dil[2] + 3);
  const int tb = (t0 - H) & ~3;
  const int sh = t0 - H - tb;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const float slope = a.slope;
  const float* xb = a.x + (size_t)b * a.bstride;
  const f32x4* wl = reinterpret_cast<const f32x4*>(a.wpack) + lane;  // conv l, tap j: wl[(l*KS + j)*64]

  f32x4 wa[KS], wb[KS];
#pragma unroll
  for (int j = 0; j < KS; ++j) wa[j] = wl[j * 64];
  {
    const int NV = XW >> 2;
    for (int e = tid; e < C * NV; e += 64 * NW) {
      const int r = e / NV, v = e - r * NV;
      const int t = tb + 4 * v;
      const int tc = t < 0 ? 0 : (t > a.ld - 4 ? a.ld - 4 : t);
      f32x4 val = *reinterpret_cast<const f32x4*>(xb + (size_t)r * a.ld + tc);
#pragma unroll
      for (int k = 0; k < 4; ++k) val[k] = (t + k >= 0 && t + k < len) ? val[k] : 0.f;
      *reinterpret_cast<f32x4*>(A + r * XW + 4 * v) = val;
    }
  }
  __syncthreads();
  int lo = sh, hi = sh + a.BN + 2 * H;

#define DISSC_RB16_LAYER(FIRST, W, IN, OUT, DIL, LAYER)                                          \
  {                                                                                              \
    const int d = (DIL);                                                                         \
    const int p = p2 * d;                                                                        \
    lo += p;                                                                                     \
    hi -= p;                                                                                     \
    const int ucol = lo + wave * (NI * 16);                                                      \
    if (ucol < hi) {                                                                             \
      f32x4 acc[NI];                                                                             \
      _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) acc[ni] = f32x4{0.f, 0.f, 0.f, 0.f};     \
      const float* bj = (IN) + g * XW + ucol - p + l15;                                          \
      _Pragma("unroll") for (int j = 0; j < KS; ++j) {                                           \
        _Pragma("unroll") for (int cq = 0; cq < 4; ++cq) {                                       \
          _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) {                                    \
            float bx = bj[cq * 4 * XW + ni * 16];                                                \
            if (FIRST) bx = bx > 0.f ? bx : bx * slope;                                          \
            acc[ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[j][cq], bx, acc[ni], 0, 0, 0);      \
          }                                                                                      \
        }                                                                                        \
        bj += d;                                                                                 \
      }                                                                                          \
      const float* bl = a.bias + (LAYER) * C;                                                    \
      _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) {                                        \
        const int u = ucol + ni * 16 + l15;                                                      \
        const int t = tb + u;                                                                    \
        const bool keep = u < hi;                                                                \
        const bool inside = t >= 0 && t < len;                                                   \
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                          \
          const int row = 4 * g + r;                                                             \
          float v = acc[ni][r] + bl[row];                                                        \
          if (FIRST) v = v > 0.f ? v : v * slope;                                                \
          else if (keep) v += (OUT)[row * XW + u];                                               \
          if (keep) (OUT)[row * XW + u] = inside ? v : 0.f;                                      \
        }                                                                                        \
      }                                                                                          \
    }                                                                                            \
  }

#pragma unroll 1
  for (int m = 0; m < 3; ++m) {
#pragma unroll
    for (int j = 0; j < KS; ++j) wb[j] = wl[((2 * m + 1) * KS + j) * 64];  // conv2's weights, used after the barrier
    __builtin_amdgcn_sched_barrier(0);
    DISSC_RB16_LAYER(true, wa, A, T, a.dil[m], 2 * m)
    __syncthreads();
    const int mn = m < 2 ? 2 * m + 2 : 2 * m + 1;  // next pair's conv1 (harmless re-read on the last pair)
#pragma unroll
    for (int j = 0; j < KS; ++j) wa[j] = wl[(mn * KS + j) * 64];
    __builtin_amdgcn_sched_barrier(0);
    DISSC_RB16_LAYER(false, wb, T, A, 1, 2 * m + 1)
    __syncthreads();
  }
#undef DISSC_RB16_LAYER

  {
    const int u0 = sh + H;
    const int nv = a.BN >> 2;
    float* ab = a.acc + (size_t)b * a.bstride;
    for (int e = tid; e < C * nv; e += 64 * NW) {
      const int r = e / nv, v4 = e - r * nv;
      const int t = t0 + 4 * v4;
      if (t >= len) continue;
      f32x4 v = *reinterpret_cast<const f32x4*>(A + r * XW + u0 + 4 * v4);
      float* dst = ab + (size_t)r * a.ld + t;
      const int nval = len - t;
      if (nval >= 4) {
        if (a.epi != EPI_MRF_SET) {
          const f32x4 o = *reinterpret_cast<const f32x4*>(dst);
          v[0] = o[0] + v[0]; v[1] = o[1] + v[1]; v[2] = o[2] + v[2]; v[3] = o[3] + v[3];
          if (a.epi == EPI_MRF_DIV) {
            v[0] = __fdiv_rn(v[0], a.mrf_div); v[1] = __fdiv_rn(v[1], a.mrf_div);
            v[2] = __fdiv_rn(v[2], a.mrf_div); v[3] = __fdiv_rn(v[3], a.mrf_div);
          }
        }
        *reinterpret_cast<f32x4*>(dst) = v;
      } else {
        for (int k = 0; k < nval; ++k) {
          float x = v[k];
          if (a.epi != EPI_MRF_SET) {
            x = dst[k] + x;
            if (a.epi == EPI_MRF_DIV) x = __fdiv_rn(x, a.mrf_div);
          }
          dst[k] = x;
        }
      }
    }
  }
}

static int g_fused_maxc = 0;  // fuse ResBlocks with C <= this ("fused_max_c"; 0 = off, the default until it beats the unfused path)

static int g_fused_var = 0;
void fused_set_option(int which, int value) {
  if (which == 2) g_fused_maxc = value;
  if (which == 3) g_fused_var = value;
}

bool resblock_fused_supported(int C, int KS, const int* dil) {
  if (C > g_fused_maxc) return false;
  if (C != 16 && C != 32) return false;
  if ((KS & 1) == 0 || KS > 11) return false;
  return dil[0] >= 1 && dil[1] >= 1 && dil[2] >= 1 && dil[0] <= 5 && dil[1] <= 5 && dil[2] <= 5;
}

template <int C, int NW, int NIMAX, typename Kern>
static int launch_fused(Kern kern, ResblockArgs a, int B, int Lmax, hipStream_t stream) {
  const int p2 = (a.KS - 1) / 2;
  const int H = p2 * (a.dil[0] + a.dil[1] + a.dil[2] + 3);
  const int dmax = std::max(a.dil[0], std::max(a.dil[1], a.dil[2]));
  const int cols = 16 * NW * NIMAX;        // tile slots per layer
  a.BN = (cols - 2 * H + 2 * p2 * a.dil[0]) / 16 * 16;  // widest layer (first conv) fits the slots
  if (a.BN > cols - 2 * H + 2 * p2 * a.dil[0]) a.BN -= 16;
  if (a.BN < 64) {
    set_error("launch_resblock_fused: kernel %d too wide for the fused tile", a.KS);
    return DISSC_EINVAL;
  }
  // widest read: last layers start at lo <= 3 + H, waves cover `cols` columns, taps reach +p
  const int need = 3 + H + cols + p2 * dmax + 1;
  int xw = (need + 31) / 32 * 32 + 16;
  if (xw - 32 >= need) xw -= 32;
  a.XW = xw;
  const size_t lds = (size_t)2 * C * xw * sizeof(float);
  if (lds > 160 * 1024) {
    set_error("launch_resblock_fused: %zu bytes of LDS needed", lds);
    return DISSC_EINVAL;
  }
  static bool attr_done = false;
  if (!attr_done) {
    DISSC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  dim3 grid((Lmax + a.BN - 1) / a.BN, B);
  hipLaunchKernelGGL(kern, grid, dim3(64 * NW), lds, stream, a);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

int launch_resblock_fused(int C, const float* x, float* acc, const float* wpack, const float* bias,
                          const int32_t* lengths, int len_default, int len_mul, int KS, const int* dil,
                          int B, int Lmax, int ld, float slope, int epi, float mrf_div,
                          hipStream_t stream) {
  ResblockArgs a;
  a.x = x; a.acc = acc; a.wpack = wpack; a.bias = bias; a.lengths = lengths;
  a.len_default = len_default; a.len_mul = len_mul; a.KS = KS;
  a.dil[0] = dil[0]; a.dil[1] = dil[1]; a.dil[2] = dil[2];
  a.ld = ld; a.bstride = (long long)C * ld; a.slope = slope; a.mrf_div = mrf_div; a.epi = epi;
  a.XW = 0; a.BN = 0;
  if (C == 16) {
    if (g_fused_var == 1) {
      if (KS == 3) return launch_fused<16, 16, 3>(&resblock16_kernel<3, 16, 3>, a, B, Lmax, stream);
      if (KS == 7) return launch_fused<16, 16, 3>(&resblock16_kernel<7, 16, 3>, a, B, Lmax, stream);
      if (KS == 11) return launch_fused<16, 16, 3>(&resblock16_kernel<11, 16, 3>, a, B, Lmax, stream);
    } else if (g_fused_var == 2) {
      return launch_fused<16, 16, 2>(&resblock_fused_kernel<16, 16, 2>, a, B, Lmax, stream);
    } else if (g_fused_var == 3) {  // 8 waves x 4 tiles: 2 workgroups per CU overlap their memory phases
      if (KS == 3) return launch_fused<16, 8, 4>(&resblock16_kernel<3, 8, 4>, a, B, Lmax, stream);
      if (KS == 7) return launch_fused<16, 8, 4>(&resblock16_kernel<7, 8, 4>, a, B, Lmax, stream);
      if (KS == 11) return launch_fused<16, 8, 4>(&resblock16_kernel<11, 8, 4>, a, B, Lmax, stream);
    } else if (g_fused_var == 4) {  // 4 waves x 8 tiles: up to 2-3 workgroups per CU
      if (KS == 3) return launch_fused<16, 4, 8>(&resblock16_kernel<3, 4, 8>, a, B, Lmax, stream);
      if (KS == 7) return launch_fused<16, 4, 8>(&resblock16_kernel<7, 4, 8>, a, B, Lmax, stream);
      if (KS == 11) return launch_fused<16, 4, 8>(&resblock16_kernel<11, 4, 8>, a, B, Lmax, stream);
    } else {
      if (KS == 3) return launch_fused<16, 16, 2>(&resblock16_kernel<3, 16, 2>, a, B, Lmax, stream);
      if (KS == 7) return launch_fused<16, 16, 2>(&resblock16_kernel<7, 16, 2>, a, B, Lmax, stream);
      if (KS == 11) return launch_fused<16, 16, 2>(&resblock16_kernel<11, 16, 2>, a, B, Lmax, stream);
    }
    return launch_fused<16, 16, 2>(&resblock_fused_kernel<16, 16, 2>, a, B, Lmax, stream);
  }
  return launch_fused<32, 16, 2>(&resblock_fused_kernel<32, 16, 2>, a, B, Lmax, stream);
}

}  // namespace dissc
