// Fused ResBlock1 in split-bf16 ("bf16x3") arithmetic for the narrow stages of the generator.
// Only used when the "precision" option is 1 (see dissc_set_option); the default fp32 path never
// comes here.
//
//   x_k = x;  3 x { x_k = x_k + conv_1(lrelu(conv_d(lrelu(x_k)))) }, d = dil[0..2]    (KS taps each)
//   then the MRF update of the stage accumulator:  acc = r | acc + r | (acc + r) / 3
// (reference sr/models.py:34-41 and :103-109).
//
// Unfused, a narrow stage moves ~51 activation passes through HBM and is bandwidth bound; on the
// bf16 matrix cores the arithmetic is nearly free, so the whole block runs out of LDS:
//   - every wave OWNS a fixed set of 16-column tiles of the window [t0 - H, t0 + BN + H) for all six
//     convs; the running value x_k of its columns stays in REGISTERS (fp32, MFMA D layout) from the
//     first load to the MRF update;
//   - what the other waves need -- lrelu(x_k) and the activated intermediate -- is published to LDS
//     already split into hi/lo bf16 halves, in planes [channel octet][hi|lo][column][8 bf16], so
//     a B fragment of v_mfma_f32_16x16x32_bf16 is one conflict-free ds_read_b128 at any tap
//     offset, and nothing is split inside the tap loop;
//   - the halo columns are recomputed and go stale inwards by p per layer; only the centre
//     [t0, t0 + BN) is written back.  HBM traffic: one read of x, one read-modify-write of acc.
// The reference's per-layer zero padding is reproduced by forcing every published value to 0
// outside the utterance's [0, len).
#include <string.h>

#include <algorithm>

#include "common.h"

namespace dissc {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct ResblockBf3Args {
  const float* x;        // [B][C][ld]   stage input
  float* acc;            // [B][C][ld]   MRF accumulator
  const bf16x8* wpack;   // [6 convs][NS steps][C/16 row blocks][hi|lo][64 lanes] x 8 bf16
  const float* bias;     // [6][C]
  const int32_t* lengths;
  int len_default, len_mul;
  int dil[3];
  int BN, H4, ld;        // centre width, halo rounded up to 4, row stride
  long long bstride;
  float slope, mrf_div;
  int epi;               // EPI_MRF_SET / EPI_MRF_ADD / EPI_MRF_DIV, or EPI_STORE: acc = x_k (a partial block)
  int m0, m1;            // residual pairs [m0, m1) of the block's three run in this launch
};

__device__ __forceinline__ float lrelu_b(float v, float slope) { return v > 0.f ? v : v * slope; }

// C = 16: K = 32 of one MFMA holds TWO taps x 16 channels (lane group g = l >> 4: tap 2s + (g >> 1),
//         channels 8 (g & 1) .. +7; an odd tap count is padded with a zero-weight tap); all weight
//         fragments of a layer live in registers.
// C = 32, 64: K = 32 is one tap x 32 channels (step s = tap * C/32 + kb: channels 32 kb + 8g .. +7),
//         C/16 row blocks of 16 per tile; the weight fragments are streamed from L2 one step ahead
//         of the MFMAs that use them.
// LDS padding columns on each side of the window: taps reach P2*d <= 5*P2 columns out (C = 16: one
// more dilation step for the zero-weight tap that pads an odd tap count to a pair)
__host__ __device__ constexpr int bf3_pad(int C, int KS) {
  return C == 16 ? 32 : ((KS - 1) / 2 * 5 + 3) & ~3;
}

template <int C, int KS, int NW, int NI>
__global__ void __launch_bounds__(64 * NW) resblock_bf3_kernel(const ResblockBf3Args a) {
  static_assert(C == 16 || C == 32 || C == 64, "narrow stages only");
  constexpr int MI = C / 16;          // 16-row blocks
  constexpr int OCT = C / 8;          // channel octets = LDS planes per half
  constexpr int NT = 64 * NW;
  constexpr int KB = (C == 16) ? 1 : C / 32;              // 32-channel blocks per tap
  constexpr int NS = (C == 16) ? (KS + 1) / 2 : KS * KB;  // MFMA K-steps per conv
  constexpr int P2 = (KS - 1) / 2;
  constexpr int COLS = 16 * NW * NI;  // window columns owned by the workgroup
  constexpr int PAD = bf3_pad(C, KS); // reach of the widest tap (dilation <= 5), a multiple of 4
  constexpr int XW = COLS + 2 * PAD;
  constexpr int TB = NI < 4 ? NI : 4; // tiles per accumulator batch
  static_assert(NI % TB == 0, "NI must be a multiple of the tile batch");
  extern __shared__ __attribute__((aligned(16))) bf16x8 lds8[];
  bf16x8* Ap = lds8;                 // lrelu(x_k):       planes [octet][hi|lo][XW]
  bf16x8* Tp = lds8 + 2 * OCT * XW;  // lrelu(conv1 out): same layout
  float* S = reinterpret_cast<float*>(Tp);  // fp32 [C][XW] transpose scratch (T is dead when used)

  const int b = blockIdx.y;
  const int len = a.lengths ? a.lengths[b] * a.len_mul : a.len_default;
  const int t0 = blockIdx.x * a.BN;
  if (t0 >= len) return;
  const int tw0 = t0 - a.H4;  // time of window column 0 (a multiple of 4)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const float slope = a.slope;
  const float* xb = a.x + (size_t)b * a.bstride;
  // fragment of (layer, step, row block, half): wl[(((layer * NS + step) * MI + mi) * 2 + half) * 64]
  const bf16x8* wl = a.wpack + lane;

  // C = 16: the whole layer; C = 32: the first step of the next layer (the rest is streamed)
  constexpr int NWR = (C == 16) ? NS : 1;
  bf16x8 w[NWR][MI][2];
  auto load_w = [&](int layer) {
#pragma unroll
    for (int s = 0; s < NWR; ++s)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        w[s][mi][0] = wl[(((layer * NS + s) * MI + mi) * 2 + 0) * 64];
        w[s][mi][1] = wl[(((layer * NS + s) * MI + mi) * 2 + 1) * 64];
      }
  };
  load_w(2 * a.m0);

  // ---- window -> scratch (coalesced 16 B per lane), zero the A pads ----
  {
    constexpr int NV = COLS / 4;
    for (int e = tid; e < C * NV; e += NT) {
      const int r = e / NV, v = e - r * NV;
      const int t = tw0 + 4 * v;
      const int tc = t < 0 ? 0 : (t > a.ld - 4 ? a.ld - 4 : t);
      f32x4 val = *reinterpret_cast<const f32x4*>(xb + (size_t)r * a.ld + tc);
#pragma unroll
      for (int k = 0; k < 4; ++k) val[k] = (t + k >= 0 && t + k < len) ? val[k] : 0.f;
      *reinterpret_cast<f32x4*>(S + r * XW + PAD + 4 * v) = val;
    }
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int e = tid; e < 2 * OCT * 2 * PAD; e += NT) {
      const int pl = e / (2 * PAD), c = e - pl * (2 * PAD);
      Ap[pl * XW + (c < PAD ? c : COLS + c)] = z;
    }
  }
  __syncthreads();

  // publish rows 16 mi + 4g .. +3 of column u, split into hi / lo halves
  auto publish = [&](bf16x8* P, int mi, int u, const float (&v)[4]) {
    bf16x4 vh, vl;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const __bf16 h = (__bf16)v[r];
      vh[r] = h;
      vl[r] = (__bf16)(v[r] - (float)h);
    }
    __bf16* p = reinterpret_cast<__bf16*>(P + (2 * (2 * mi + (g >> 1))) * XW + PAD + u) + 4 * (g & 1);
    *reinterpret_cast<bf16x4*>(p) = vh;
    *reinterpret_cast<bf16x4*>(p + XW * 8) = vl;
  };

  const int ucol = wave * (NI * 16);
  float xk[MI][NI][4];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int u = ucol + ni * 16 + l15;
      float av[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        xk[mi][ni][r] = S[(16 * mi + 4 * g + r) * XW + PAD + u];
        av[r] = lrelu_b(xk[mi][ni][r], slope);
      }
      publish(Ap, mi, u, av);
    }
  __syncthreads();
  {  // the scratch is consumed: zero the T pads (ordered before conv2 by the barrier after conv1)
    const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int e = tid; e < 2 * OCT * 2 * PAD; e += NT) {
      const int pl = e / (2 * PAD), c = e - pl * (2 * PAD);
      Tp[pl * XW + (c < PAD ? c : COLS + c)] = z;
    }
  }

  // one conv over TB tiles starting at window column ub: acc += W (x) P[.. - p + j*d ..]
  auto conv = [&](const bf16x8* P, int layer, int d, int ub, f32x4 (&acc)[MI][TB]) {
    const int oct = (C == 16) ? (g & 1) : g;
    const int toff = (C == 16) ? (g >> 1) * d : 0;
    const bf16x8* b0 = P + (2 * oct) * XW + PAD + ub + l15 - P2 * d + toff;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < TB; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 wc[MI][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      wc[mi][0] = w[0][mi][0];
      wc[mi][1] = w[0][mi][1];
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      bf16x8 wn[MI][2];
      if constexpr (C == 16) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          wn[mi][0] = w[s + 1 < NS ? s + 1 : s][mi][0];
          wn[mi][1] = w[s + 1 < NS ? s + 1 : s][mi][1];
        }
      } else if (s + 1 < NS) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          wn[mi][0] = wl[(((layer * NS + s + 1) * MI + mi) * 2 + 0) * 64];
          wn[mi][1] = wl[(((layer * NS + s + 1) * MI + mi) * 2 + 1) * 64];
        }
      }
      // C = 16: step s covers taps 2s, 2s+1; else tap s / KB, channel block s % KB (8 planes each)
      const bf16x8* bj = (C == 16) ? b0 + 2 * s * d : b0 + (s / KB) * d + (s % KB) * (8 * XW);
#pragma unroll
      for (int ni = 0; ni < TB; ++ni) {
        const bf16x8 bh = bj[ni * 16];
        const bf16x8 bl = bj[ni * 16 + XW];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[mi][1], bh, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[mi][0], bl, acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wc[mi][0], bh, acc[mi][ni], 0, 0, 0);
        }
      }
      if (s + 1 < NS) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          wc[mi][0] = wn[mi][0];
          wc[mi][1] = wn[mi][1];
        }
      }
    }
  };

#pragma unroll 1
  for (int m = a.m0; m < a.m1; ++m) {
    // ---- conv1 (dilated): A -> T ----
    {
      float bz[MI][4];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) bz[mi][r] = a.bias[(2 * m) * C + 16 * mi + 4 * g + r];
      const int d = a.dil[m];
      f32x4 acc[NI / TB][MI][TB];
#pragma unroll
      for (int tb = 0; tb < NI / TB; ++tb) conv(Ap, 2 * m, d, ucol + tb * (TB * 16), acc[tb]);
      load_w(2 * m + 1);
#pragma unroll
      for (int tb = 0; tb < NI / TB; ++tb)
#pragma unroll
        for (int ni = 0; ni < TB; ++ni) {
          const int u = ucol + (tb * TB + ni) * 16 + l15;
          const int t = tw0 + u;
          const bool inside = t >= 0 && t < len;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = inside ? lrelu_b(acc[tb][mi][ni][r] + bz[mi][r], slope) : 0.f;
            publish(Tp, mi, u, v);
          }
        }
    }
    __syncthreads();
    // ---- conv2 (dilation 1): T -> x_k += ..., publish lrelu(x_k) to A ----
    {
      float bz[MI][4];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 4; ++r) bz[mi][r] = a.bias[(2 * m + 1) * C + 16 * mi + 4 * g + r];
      f32x4 acc[NI / TB][MI][TB];
#pragma unroll
      for (int tb = 0; tb < NI / TB; ++tb) conv(Tp, 2 * m + 1, 1, ucol + tb * (TB * 16), acc[tb]);
      if (m + 1 < a.m1) load_w(2 * m + 2);
#pragma unroll
      for (int tb = 0; tb < NI / TB; ++tb)
#pragma unroll
        for (int ni = 0; ni < TB; ++ni) {
          const int nn = tb * TB + ni;
          const int u = ucol + nn * 16 + l15;
          const int t = tw0 + u;
          const bool inside = t >= 0 && t < len;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              xk[mi][nn][r] = inside ? xk[mi][nn][r] + (acc[tb][mi][ni][r] + bz[mi][r]) : 0.f;
              v[r] = lrelu_b(xk[mi][nn][r], slope);
            }
            if (m + 1 < a.m1) publish(Ap, mi, u, v);
          }
        }
    }
    __syncthreads();
  }

  // ---- x_k -> scratch -> MRF update of the centre columns, 16 B per lane ----
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int u = ucol + ni * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) S[(16 * mi + 4 * g + r) * XW + PAD + u] = xk[mi][ni][r];
    }
  __syncthreads();
  {
    const int nv = a.BN >> 2;
    const bool rmw = a.epi == EPI_MRF_ADD || a.epi == EPI_MRF_DIV;
    float* ab = a.acc + (size_t)b * a.bstride;
    for (int e = tid; e < C * nv; e += NT) {
      const int r = e / nv, v4 = e - r * nv;
      const int t = t0 + 4 * v4;
      if (t >= len) continue;
      f32x4 v = *reinterpret_cast<const f32x4*>(S + r * XW + PAD + a.H4 + 4 * v4);
      float* dst = ab + (size_t)r * a.ld + t;
      const int nval = len - t;
      if (nval >= 4) {
        if (rmw) {
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
          if (rmw) {
            x = dst[k] + x;
            if (a.epi == EPI_MRF_DIV) x = __fdiv_rn(x, a.mrf_div);
          }
          dst[k] = x;
        }
      }
    }
  }
}

// ---- host side ---------------------------------------------------------------------------

static inline uint16_t bf16_rne_host(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32_host(uint16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// option "fused_variant" (Options::bf3_variant, default 0): "fused_variant" in precision mode: 0 = 512-column window, 1 = 1024

// option "bf3_pairs" (Options::bf3_pairs, default -1): "bf3_pairs": -1 = by channel count (below), 0 = whole blocks, 1 = always pairs
// One launch per residual pair instead of per block?  With 64 channels the LDS holds 256-column
// windows only, and the 120-column halo of a whole 11-tap block would be recomputed ~2x; a pair's
// halo is 10-30 columns, at the price of two more read+write passes of x_k per block.
bool resblock_bf3_pairs(int C) { return opts().bf3_pairs < 0 ? C >= 64 : opts().bf3_pairs != 0; }

bool resblock_bf3_supported(int C, int KS, const int* dil) {
  if (C != 16 && C != 32 && C != 64) return false;
  // (C = 64 leaves room for 256-column windows only; with 11 taps the halo is 120 of them and the
  // fused block alone is slower than its six layers, 3.2 vs 2.6 ms -- but it moves 5x fewer bytes,
  // and with the three ResBlock chains of a stage running concurrently the stage is faster fused.)
  if (KS != 3 && KS != 7 && KS != 11) return false;
  for (int m = 0; m < 3; ++m)
    if (dil[m] < 1 || dil[m] > 5) return false;
  return true;
}

// w6: the six conv weights [C][C][KS] in execution order (c1_0, c2_0, c1_1, c2_1, c1_2, c2_2).
// packed (as raw 32-bit words): [6][NS steps][C/16 row blocks][hi|lo][64 lanes][8 bf16]; lane l of
// row block mi holds row 16 mi + (l & 15) and, with g = l >> 4,
//   C = 16: tap 2s + (g >> 1), channels 8 (g & 1) + e      else: tap s / (C/32), channels 32 (s % (C/32)) + 8g + e
void pack_resblock_bf3(int C, int KS, const float* const* w6, std::vector<float>& packed) {
  const int MI = C / 16;
  const int KB = (C == 16) ? 1 : C / 32;
  const int NS = (C == 16) ? (KS + 1) / 2 : KS * KB;
  packed.assign((size_t)6 * NS * MI * 2 * 64 * 4, 0.f);
  uint16_t* p16 = reinterpret_cast<uint16_t*>(packed.data());
  for (int l = 0; l < 6; ++l)
    for (int s = 0; s < NS; ++s)
      for (int mi = 0; mi < MI; ++mi)
        for (int lane = 0; lane < 64; ++lane) {
          const int co = 16 * mi + (lane & 15), gq = lane >> 4;
          const int tap = (C == 16) ? 2 * s + (gq >> 1) : s / KB;
          for (int e = 0; e < 8; ++e) {
            const int ci = (C == 16) ? 8 * (gq & 1) + e : 32 * (s % KB) + 8 * gq + e;
            const float v = tap < KS ? w6[l][((size_t)co * C + ci) * KS + tap] : 0.f;
            const uint16_t hi = bf16_rne_host(v);
            const uint16_t lo = bf16_rne_host(v - bf16_to_f32_host(hi));
            const size_t slot = ((((size_t)l * NS + s) * MI + mi) * 2) * 64 + lane;
            p16[slot * 8 + e] = hi;
            p16[(slot + 64) * 8 + e] = lo;
          }
        }
}

template <int C, int KS, int NW, int NI>
static int launch_bf3(ResblockBf3Args a, int B, int Lmax, hipStream_t stream) {
  constexpr int COLS = 16 * NW * NI, XW = COLS + 2 * bf3_pad(C, KS);
  const int P2 = (KS - 1) / 2;
  int H = 0;
  for (int m = a.m0; m < a.m1; ++m) H += P2 * (a.dil[m] + 1);
  a.H4 = (H + 3) & ~3;
  a.BN = COLS - 2 * a.H4;
  const size_t lds = (size_t)4 * (C / 8) * XW * 16;
  static DeviceOnce attr_once;  // per device (common.h)
  DISSC_HIP_CHECK(attr_once.max_lds(reinterpret_cast<const void*>(&resblock_bf3_kernel<C, KS, NW, NI>), 160 * 1024));
  dim3 grid((Lmax + a.BN - 1) / a.BN, B);
  hipLaunchKernelGGL((resblock_bf3_kernel<C, KS, NW, NI>), grid, dim3(64 * NW), lds, stream, a);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

int launch_resblock_bf3(int C, const float* x, float* acc, const float* wpack, const float* bias,
                        const int32_t* lengths, int len_default, int len_mul, int KS, const int* dil,
                        int B, int Lmax, int ld, float slope, int epi, float mrf_div, int m0, int m1,
                        hipStream_t stream) {
  if (!resblock_bf3_supported(C, KS, dil) || m0 < 0 || m1 > 3 || m0 >= m1) {
    set_error("launch_resblock_bf3: C=%d k=%d unsupported", C, KS);
    return DISSC_EINVAL;
  }
  ResblockBf3Args a;
  a.x = x; a.acc = acc; a.wpack = reinterpret_cast<const bf16x8*>(wpack); a.bias = bias; a.lengths = lengths;
  a.len_default = len_default; a.len_mul = len_mul;
  a.dil[0] = dil[0]; a.dil[1] = dil[1]; a.dil[2] = dil[2];
  a.ld = ld; a.bstride = (long long)C * ld; a.slope = slope; a.mrf_div = mrf_div; a.epi = epi;
  a.BN = 0; a.H4 = 0; a.m0 = m0; a.m1 = m1;
  if (C == 64) {  // 256-column windows fill the CU's 160 KB of LDS
    if (KS == 3) return launch_bf3<64, 3, 8, 2>(a, B, Lmax, stream);
    if (KS == 7) return launch_bf3<64, 7, 8, 2>(a, B, Lmax, stream);
    return launch_bf3<64, 11, 8, 2>(a, B, Lmax, stream);
  }
  if (C == 32) {  // 512-column windows (135-145 KB of LDS): one workgroup of 8 waves per CU
    if (KS == 3) return launch_bf3<32, 3, 8, 4>(a, B, Lmax, stream);
    if (KS == 7) return launch_bf3<32, 7, 8, 4>(a, B, Lmax, stream);
    return launch_bf3<32, 11, 8, 4>(a, B, Lmax, stream);
  }
  if (opts().bf3_variant == 1) {  // 1024-column windows: less halo recompute, but one workgroup per CU
    if (KS == 3) return launch_bf3<16, 3, 8, 8>(a, B, Lmax, stream);
    if (KS == 7) return launch_bf3<16, 7, 8, 8>(a, B, Lmax, stream);
    return launch_bf3<16, 11, 8, 8>(a, B, Lmax, stream);
  }
  // default: 512-column windows, two workgroups per CU overlap each other's memory phases
  if (KS == 3) return launch_bf3<16, 3, 8, 4>(a, B, Lmax, stream);
  if (KS == 7) return launch_bf3<16, 7, 8, 4>(a, B, Lmax, stream);
  return launch_bf3<16, 11, 8, 4>(a, B, Lmax, stream);
}

}  // namespace dissc
