// conv_wino_kernel: the stride-1 "same" convolutions of the wide ResBlock stages (C -> C channels, k = 3 / 7 / 11,
// dilation 1 / 3 / 5; reference sr/models.py:16-41) evaluated with fewer multiplications: Toom-Cook F(4,3) over
// 3-tap sub-filters, still fp32 operands, fp32 products and fp32 accumulation on v_mfma_f32_32x32x2_f32.
//
//   y[n] = sum_i w[i] xa[n - pad + i d],  xa = lrelu(x) inside the utterance, 0 outside.
//
// * The k taps are split into NS = ceil(k / 3) sub-filters of 3 taps with tap stride NS (sub-filter j = taps j, j + NS,
//   j + 2 NS; missing taps are zero).  With D = d NS, sub-filter j is a 3-tap filter on the sequence sampled every D.
// * F(4,3): 4 outputs of a 3-tap filter from 6 inputs with 6 multiplications: y = A^T [(G g) . (B^T s)], evaluation points
//   0, 1, -1, 2, -1/2, inf (rows of B^T scaled to exactly representable entries; the scaling is folded into G, which the
//   host applies to the weights in double precision).  The products of the 6 points are 6 independent GEMMs over the input
//   channels AND the NS sub-filters (the sums over sub-filters happen in the transform domain):
//       Y_p[co][tau][rho] = sum_j sum_ci U_pj[co][ci] V_p[ci][tau][rho + d j],     rho in [0, D), j in [0, NS)
//       V_p[ci][tau][w]   = sum_m B^T[p][m] xa[ci][o + 4 D tau + w + D m],          w in [0, d (2 NS - 1))
//       y[co][t0 + 4 D tau + rho + D a] = sum_p A^T[a][p] Y_p[co][tau][rho] + bias,  a in [0, 4)
//   i.e. per point a dilation-d, NS-tap convolution in the transform domain: 6 NS MFMA products per 4 outputs instead of
//   4 k (k = 3: 6 vs 12, k = 7: 18 vs 28, k = 11: 24 vs 44).
// * One workgroup = 12 waves = 6 points x 2 halves (three waves on every SIMD; one workgroup fills a CU): for C >= 128 the
//   halves are the 64-row halves of a [128 rows] x [64 transform-domain columns = 240..256 outputs] tile, for C = 64 the
//   64-column halves of a [64 rows] x [128 columns] tile.  All waves share the activated input window of 16 / 32 channels
//   (staged like conv_mfma32_kernel: registers -> lrelu / zero padding -> LDS, double-buffered, loads two rounds ahead),
//   kept in a polyphase layout [channel][x mod D][x div D] so that the 6 samples of an F(4,3) group are two aligned
//   16-byte reads.  V_p is formed 8 channels at a time -- C = 64: by wave (p, half) for itself; C >= 128: by all 12 waves
//   together, (channel pair, point pair) each, one sub-chunk ahead of the MFMAs into a double-buffered shared tile set,
//   one barrier per sub-chunk -- and multiplied by the same tap / k-step MFMA loop as conv_mfma32_kernel with the point's
//   weights U_p (A fragments streamed from L2 in the order the loop walks them).  Epilogue: the 6 waves of a half
//   exchange their Y_p through LDS 32 rows at a time, one thread applies A^T, bias and the residual / MRF mode to 4
//   consecutive outputs.  DESIGN.md section 4 has the measurements and the versions that led here.
// The results differ from the direct kernels by fp32 rounding only (measured per layer: rms 3e-7 relative to O(0.4)
// outputs against 1e-7 for the direct form); the direct kernels remain (option "wino" = 0).
#include <string.h>

#include <type_traits>

#include "common.h"
#include "conv_epilogue32.h"
#include "wino_common.h"

namespace dissc {

// option "wino" (Options::wino, default 1): "wino" option: 1 = wide ResBlock convs through conv_wino_kernel (read at dissc_gen_create);
                         // 2 = the stand-alone dissc_conv1d entry uses it too (tests)
// option "wino_min_c" (Options::wino_min_c, default 64): "wino_min_c" option: narrowest stage that uses it
// option "wino_c64_kmin" (Options::wino_c64_kmin, default 3): "wino_c64_kmin" option: smallest kernel size that uses it in a 64-channel stage (in the generator k = 3 /
                         // 7 gain 2 % per forward there; in isolation they are break-even against the DMA-staged direct pair)
// option "wino_small" (Options::wino_small, default 96): "wino_small" option: launches with fewer 64 x 64-tile workgroups than this use 32 x 32 wave tiles
// option "kernel_dbg": diagnostics: knock-outs, bit 0 transform, 1 MFMAs, 2 epilogue, 3 staging

struct WinoArgs {
  const float* x;
  const float* wpack;   // [6 points][C / 32][nchunk][2 halves][NS taps][64 lanes][4 k-steps] (make_wino)
  const float* bias;    // [C]
  const float* res;
  float* out;
  float* acc;
  const int32_t* lengths;
  int len_default, len_mul;
  int C, nchunk, pad;
  int ldx, ldo;
  long long x_bstride, o_bstride;
  float slope, mrf_div;
  int epi;
  int dbg;
  int gx, gy, B;  // time tiles, 128-row tiles, utterances
};

#ifndef DISSC_WINO_LB
#define DISSC_WINO_LB 3
#endif
#ifndef DISSC_WINO_EPI2
#define DISSC_WINO_EPI2 1  // epilogue: A^T per column through an output tile in LDS (0: per output quad straight from the Y tiles)
#endif

// polyphase row length: indices 0 .. 4 NTU + 7 are touched (the last aligned 16-byte read of a lane starts at
// 4 (NTU - 1) + 8), rounded up so that RL / 4 makes the 16 lanes of a quarter wave start in 16 different bank groups
// where that is possible (lane = tau * D + phi reads at phi * RL + 4 tau)
constexpr int wino_row_len(int D, int NTU, int XRW) {
  int need = 4 * NTU + 8;
  const int span = (XRW - 1 + 4 * D) / D + 1;
  if (span > need) need = span;
  need = (need + 3) / 4 * 4;
  if (D == 1) return need;
  for (int rl = need; rl < need + 64 && 2 * 32 * D * rl * 4 + 12 * 8 * 112 * 4 <= 156 * 1024; rl += 4) {
    bool ok = true;
    for (int c0 = 0; c0 < 64 && ok; c0 += 16) {
      unsigned seen = 0;
      for (int c = c0; c < c0 + 16; ++c) {
        const int g = ((c % D) * (rl / 4) + c / D) % 16;
        if (seen & (1u << g)) ok = false;
        seen |= 1u << g;
      }
    }
    if (ok) return rl;
  }
  return need;
}

template <int NS, int DIL, int CPR, int RH, int TW, int SHV = 0>
__global__ void __launch_bounds__(768, DISSC_WINO_LB) conv_wino_kernel(const WinoArgs a) {
  static_assert(SHV == 0 || RH == 2, "the shared transform is for the row-half form");
  constexpr int NTH = 768;                    // 12 waves: 6 points x 2 row halves (3 waves on every SIMD)
  constexpr int D = DIL * NS;                 // sample step of the F(4,3) sequences
  constexpr int W = DIL * (2 * NS - 1);       // V entries per tile unit
  constexpr int CHV = 3 - RH;                 // column halves per workgroup (RH = 2 row halves -> 1, RH = 1 -> 2)
  constexpr int MI = TW, NI = TW;             // wave tile = (32 MI) rows x (32 NI) columns: 64 x 64, or 32 x 32 on small grids
  constexpr int TC = 32 * NI;                 // columns per wave
  constexpr int NTU = TC / D;                 // tile units per wave
  constexpr int NCOL = NTU * D;               // transform-domain columns of a wave in use (<= TC)
  constexpr int OT = 4 * D * NTU * CHV;       // outputs per workgroup tile
  constexpr int RAW = OT + DIL * (3 * NS - 1);  // input window
  constexpr int XRW = (RAW + 3 + 3) / 4 * 4;  // staged positions per channel (alignment shift <= 3)
  constexpr int NV = XRW / 4;
  constexpr int RL = wino_row_len(D, NTU * CHV, XRW);
  constexpr int CHF = D * RL;                 // floats per channel of the polyphase window
  constexpr int VROW = NTU * W;               // V entries per channel (<= 112)
  constexpr int XV = 112;                     // V row stride (112 % 32 == 16: the two k halves of a fragment read hit different banks)
  constexpr int SV = (CPR * NV + NTH - 1) / NTH;
  constexpr int YS = TC + 4;                  // row stride of the epilogue exchange
  static_assert(VROW <= XV && NCOL <= TC && NTU >= 1, "tile geometry");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // 2 x [CPR][D][RL]: window sample x (position o + x) of a channel at [(x + 4 D) % D][(x + 4 D) / D]
  float* vbuf = lds + 2 * CPR * CHF;  // V tiles: [12 waves][8][XV], or [2][6 points][8][XV] in the shared form

  // 1-D grid, XCD-aware: workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2; the 128-row tiles
  // (their transform-domain weights are 6 NS / k times the direct form's: 3.1 MB per tile for C = 256, k = 11) are
  // pinned to XCD groups so that an L2 holds ONE tile's weights, and the time tiles / utterances are spread inside a group
  const int per = 8 / a.gy;
  const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
  const int mt = xcd / per;  // row tile (32 MI RH rows)
  const int lin = (xcd - mt * per) + per * kq;
  if (lin >= a.gx * a.B) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // lin counts the tiles that EXIST: utterance 0's ceil(len_0 / OT), then utterance 1's, ...  (Enumerating gx tiles for
  // every utterance and returning from those beyond its end costs 7-9 % on padded or ragged batches: the empty workgroups
  // sit between the real ones in dispatch order and the real ones land unevenly on the XCDs.)  Every wave finds (b, tile)
  // by a prefix sum of the tile counts over its lanes.
  int b = -1, len = a.len_default, t0 = 0;
  if (a.lengths == nullptr) {
    b = lin / a.gx;
    t0 = (lin - b * a.gx) * OT;
    if (t0 >= len) return;
  } else {
    int base = 0;
    for (int b0 = 0; b0 < a.B; b0 += 64) {
      const int l = b0 + lane < a.B ? a.lengths[b0 + lane] * a.len_mul : 0;
      const int nt = (l + OT - 1) / OT;
      int incl = nt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
      }
      const int total = __shfl(incl, 63, 64);
      if (lin < base + total) {
        const unsigned long long m = __ballot(base + incl > lin);
        const int lb = __ffsll((long long)m) - 1;
        b = b0 + lb;
        len = __builtin_amdgcn_readfirstlane(__shfl(l, lb, 64));
        t0 = (lin - base - __builtin_amdgcn_readfirstlane(__shfl(incl - nt, lb, 64))) * OT;
        break;
      }
      base += total;
    }
    if (b < 0) return;
    b = __builtin_amdgcn_readfirstlane(b);
    t0 = __builtin_amdgcn_readfirstlane(t0);
  }
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = wave % 6;   // this wave's evaluation point
  const int mh = RH == 2 ? wave / 6 : 0;   // ... its 64-row half of the tile (RH = 2)
  const int chh = RH == 2 ? 0 : wave / 6;  // ... or its 64-column half (RH = 1)
  const int l31 = lane & 31, h = lane >> 5;
  const int o = t0 - a.pad;
  const int tb = o & ~3;
  const int sh = o - tb;
  const float slope = a.slope;
  const float* xb = a.x + (size_t)b * a.x_bstride;
  const int nround = a.nchunk / (CPR / KC);  // CPR channels per barrier round (the host checks divisibility)

  // ---- window staging: unconditional clamped 16-byte loads; activation, zero padding and the polyphase scatter on the
  // way to LDS (samples D apart become neighbours, so that a lane's 6-sample groups are two aligned 16-byte reads)
  const int r0 = tid / NV, v0 = tid - r0 * NV;
  constexpr int dr = NTH / NV, dv = NTH - dr * NV;
  f32x4 sv[SV];
  auto stage_load = [&](int rd) {
    int r = r0, v = v0;
#pragma unroll
    for (int i = 0; i < SV; ++i) {
      int ci = rd * CPR + (r < CPR ? r : CPR - 1);
      ci = ci < a.C ? ci : a.C - 1;
      int t = tb + 4 * v;
      t = t < 0 ? 0 : (t > a.ldx - 4 ? a.ldx - 4 : t);
      sv[i] = *reinterpret_cast<const f32x4*>(xb + (size_t)ci * a.ldx + t);
      v += dv;
      r += dr;
      if (v >= NV) { v -= NV; ++r; }
    }
  };
  auto stage_store = [&](float* raw) {
    int r = r0, v = v0;
#pragma unroll
    for (int i = 0; i < SV; ++i) {
      if (r < CPR) {
        const int t = tb + 4 * v;
        const f32x4 val = sv[i];
        float* rowp = raw + r * CHF;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = (t + e) >= 0 && (t + e) < len;
          const int xs = 4 * v + e - sh + 4 * D;  // >= 4 D - 3 > 0
          rowp[(xs % D) * RL + xs / D] = ok ? lrelu(val[e], slope) : 0.f;
        }
      }
      v += dv;
      r += dr;
      if (v >= NV) { v -= NV; ++r; }
    }
  };

  // ---- transform: lane <-> column (tau, phi); entries w = phi and w = phi + D of unit tau
  const int tcol = lane < NCOL ? lane : NCOL - 1;
  const int ttau = tcol / D, tphi = tcol % D;
  const int toff = tphi * RL + 4 * (chh * NTU + ttau) + 4;  // first of the 8 neighbouring samples (16-byte aligned)
  const int e0 = ttau * W + tphi;
  const bool ok1 = tphi + D < W;
  const int e1 = ok1 ? e0 + D : e0;
  float* vp = vbuf + wave * (8 * XV);

  // ---- B fragment offsets: column (ni * 32 + l31) -> (tau, rho) -> tau * W + rho, k half h -> row h
  int voff[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    int col = ni * 32 + l31;
    col = col < NCOL ? col : NCOL - 1;
    voff[ni] = (col / D) * W + (col % D) + h * XV;
  }

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  // A fragments: one float4 per lane = the 4 k-steps (8 channels) of one (chunk, half, tap) block, packed in the order
  // the loop walks them: [point][32-row subtile][block][lane] -- one pointer step per tap
  const int nsub = a.C / 32;
  const int nblk = a.nchunk * NS * 2;
  f32x4 av[MI], avn[MI];
  // scalar-base loads (common.h: wave_rsrc): this wave's MI subtiles are one contiguous slab [mi][block][lane]
  const int slab = __builtin_amdgcn_readfirstlane(p * nsub + mt * (MI * RH) + mh * MI);
  const __amdgpu_buffer_rsrc_t wrs = wave_rsrc(reinterpret_cast<const f32x4*>(a.wpack) + (size_t)slab * nblk * 64, (unsigned)(MI * nblk) * 1024u);
  const unsigned lane16 = lane * 16u;
  auto a_load = [&](int mi, int bl) __attribute__((always_inline)) { return rsrc_load16(wrs, lane16, (unsigned)(mi * nblk + bl) * 1024u); };
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) av[mi] = a_load(mi, 0);
  stage_load(0);
  stage_store(lds);
  if (nround > 1) stage_load(1);  // the staging registers always hold the round after the newest one in LDS
  __syncthreads();

  // One barrier per round: at the top of round rd the window of round rd + 1 goes into the other buffer (every wave
  // finished reading it before the barrier that closed round rd - 1) and the loads of round rd + 2 are issued; the
  // barrier at the bottom publishes that window and retires this round's.
  // ---- NS taps x 4 k-steps on one point's V tile; the B fragments of tap j + 1 and the A fragments of block blk + 2 are
  // fetched behind the MFMAs of tap j
  int blk = 0;
  auto run_taps = [&](const float* vt) {
      if (!(a.dbg & 2)) {
        const float* bj[NI];
        float b0[NI], b0n[NI], bk[3][NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          bj[ni] = vt + voff[ni];
          b0[ni] = bj[ni][0];
        }
#pragma unroll
        for (int j = 0; j < NS; ++j, ++blk) {
          const int bn = (blk + 1 < nblk) ? blk + 1 : nblk - 1;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) avn[mi] = a_load(mi, bn);
#pragma unroll
          for (int s = 1; s < 4; ++s)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bk[s - 1][ni] = bj[ni][s * 2 * XV + j * DIL];
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) b0n[ni] = (j + 1 < NS) ? bj[ni][(j + 1) * DIL] : 0.f;  // next tap's first k-step
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][0], b0[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
          for (int s = 1; s < 4; ++s)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
              for (int ni = 0; ni < NI; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][s], bk[s - 1][ni], acc[mi][ni], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) b0[ni] = b0n[ni];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) av[mi] = avn[mi];
        }
      } else {
        blk += NS;
      }
  };

  if constexpr (SHV) {
    // ---- shared transform (row-half form): the 12 waves split a sub-chunk's 8 channels x 6 points as (channel pair,
    // point pair) -- every window sample is read from LDS once per point pair instead of once per wave (4x fewer
    // 16-byte reads, half the transform arithmetic: the two row halves used to form the same V_p twice) -- into a
    // double-buffered V [2][6][8][XV]: the tile of sub-chunk g + 1 is formed while g is multiplied; one barrier each
    constexpr int VSZ = 6 * 8 * XV;
    const int chp = wave & 3, pp = wave >> 2;
    f32x4 lo[2], hi[2];
    auto tsv_load = [&](const float* rawbuf, int sc) {
      const float* rw = rawbuf + (sc * 8 + 2 * chp) * CHF + toff;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        lo[i] = *reinterpret_cast<const f32x4*>(rw + i * CHF);
        hi[i] = *reinterpret_cast<const f32x4*>(rw + i * CHF + 4);
      }
    };
    auto tsv_finish = [&](float* vdst) {
      auto two = [&](auto pc) {
        constexpr int P0 = decltype(pc)::value, P1 = P0 + 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float* d0 = vdst + (P0 * 8 + 2 * chp + i) * XV;
          float* d1 = vdst + (P1 * 8 + 2 * chp + i) * XV;
          const float a0 = wino_bt<P0>(lo[i][0], lo[i][1], lo[i][2], lo[i][3], hi[i][0], hi[i][1]);
          const float c0 = wino_bt<P1>(lo[i][0], lo[i][1], lo[i][2], lo[i][3], hi[i][0], hi[i][1]);
          d0[e0] = a0;
          d1[e0] = c0;
          if constexpr (NS > 1) {
            const float a1 = wino_bt<P0>(lo[i][1], lo[i][2], lo[i][3], hi[i][0], hi[i][1], hi[i][2]);
            const float c1 = wino_bt<P1>(lo[i][1], lo[i][2], lo[i][3], hi[i][0], hi[i][1], hi[i][2]);
            d0[e1] = ok1 ? a1 : a0;
            d1[e1] = ok1 ? c1 : c0;
          }
        }
      };
      switch (pp) {  // uniform per wave
        case 0: two(std::integral_constant<int, 0>{}); break;
        case 1: two(std::integral_constant<int, 2>{}); break;
        default: two(std::integral_constant<int, 4>{}); break;
      }
    };
    if (!(a.dbg & 1)) {
      tsv_load(lds, 0);
      tsv_finish(vbuf);
    }
    __syncthreads();
    int g = 0;
    for (int rd = 0; rd < nround; ++rd) {
      const float* raw = lds + (rd & 1) * (CPR * CHF);
      float* rawn = lds + ((rd + 1) & 1) * (CPR * CHF);
      // window rd + 1 goes into the other buffer at the top of the round (its last readers -- the transforms of round
      // rd - 1's last sub-chunk -- finished two barriers ago); the barrier of sub-chunk 0 publishes it before the last
      // sub-chunk of this round transforms its first 8 channels
      if (!(a.dbg & 8) && rd + 1 < nround) stage_store(rawn);
      for (int sc = 0; sc < CPR / 8; ++sc, ++g) {
        if (!(a.dbg & 8) && rd + 2 < nround && sc == (wave >> 2) % (CPR / 8)) stage_load(rd + 2);
        float* vnxt = vbuf + ((g + 1) & 1) * VSZ;
        // (issuing these window reads behind the first tap's fragment reads and finishing them behind its MFMAs, so that
        // their latency never sits in front of a matrix instruction, measured 3-8 % SLOWER: +20 registers)
        if (!(a.dbg & 1)) {
          if (sc + 1 < CPR / 8) {
            tsv_load(raw, sc + 1);
            tsv_finish(vnxt);
          } else if (rd + 1 < nround) {
            tsv_load(rawn, 0);
            tsv_finish(vnxt);
          }
        }
        run_taps(vbuf + (g & 1) * VSZ + p * (8 * XV));
        __syncthreads();
      }
    }
  } else {
  for (int rd = 0; rd < nround; ++rd) {
    const float* raw = lds + (rd & 1) * (CPR * CHF);
    if (!(a.dbg & 8) && rd + 1 < nround) stage_store(lds + ((rd + 1) & 1) * (CPR * CHF));
    for (int sc = 0; sc < CPR / 8; ++sc) {
      // The loads of round rd + 2 are issued at a DIFFERENT sub-chunk by each of the three waves of a SIMD: vmcnt retires
      // in order, so a wave cannot consume a weight fragment fetched after its window loads before those have landed
      // (~2 us from HBM) -- staggered, the other two waves of the SIMD keep the matrix pipe busy meanwhile
      if (!(a.dbg & 8) && rd + 2 < nround && sc == (wave >> 2) % (CPR / 8)) stage_load(rd + 2);
      // ---- V_p[i][e] of channels sc * 8 + i (branch-free: lanes without a second entry rewrite their first one, lanes
      // beyond the last column duplicate it)
      if (!(a.dbg & 1)) {
        const float* rw = raw + (sc * 8) * CHF + toff;
        auto transform8 = [&](auto pc) {
          constexpr int P = decltype(pc)::value;
#pragma unroll
          for (int i0 = 0; i0 < 8; i0 += 2) {
            f32x4 lo[2], hi[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              lo[i] = *reinterpret_cast<const f32x4*>(rw + (i0 + i) * CHF);
              hi[i] = *reinterpret_cast<const f32x4*>(rw + (i0 + i) * CHF + 4);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const float v0 = wino_bt<P>(lo[i][0], lo[i][1], lo[i][2], lo[i][3], hi[i][0], hi[i][1]);
              vp[(i0 + i) * XV + e0] = v0;
              if constexpr (NS > 1) {
                const float v1 = wino_bt<P>(lo[i][1], lo[i][2], lo[i][3], hi[i][0], hi[i][1], hi[i][2]);
                vp[(i0 + i) * XV + e1] = ok1 ? v1 : v0;
              }
            }
          }
        };
        switch (p) {  // uniform per wave
          case 0: transform8(std::integral_constant<int, 0>{}); break;
          case 1: transform8(std::integral_constant<int, 1>{}); break;
          case 2: transform8(std::integral_constant<int, 2>{}); break;
          case 3: transform8(std::integral_constant<int, 3>{}); break;
          case 4: transform8(std::integral_constant<int, 4>{}); break;
          default: transform8(std::integral_constant<int, 5>{}); break;
        }
      }
      run_taps(vp);
    }
    __syncthreads();
  }
  }

  // ---- epilogue: the 6 waves exchange their Y_p through LDS, 32 rows at a time; one thread = 4 consecutive outputs
  // (16-byte residual / accumulator / output accesses) of one row: output u = 4 g + e of unit tau comes from column
  // (tau, u % D) with A^T row a = u / D
  float* yb = lds;  // [6][32][YS]
  // epilogue form: A^T per column through an output tile in LDS for the steps D that are neither 1 nor a multiple of 4
  // (k = 7 and the dilated k = 3 instances, whose per-quad form needs per-element divisions by D); per output quad
  // straight from the Y tiles otherwise (measured: the D % 4 == 0 instances are 1-4 % SLOWER through the output tile)
  constexpr bool EPI2 = DISSC_WINO_EPI2 && D != 1 && D % 4 != 0;
  constexpr int OSW = 4 * D * NTU + 4;     // row stride of the output tile of a pass (one column half)
  float* const ot = lds + 6 * 32 * YS;     // [32][OSW]
  if (a.dbg & 4) {
    if (acc[0][0][0] == 123.f) a.out[0] = 1.f;
    return;
  }
  const int epi = a.epi;
  const size_t ob = (size_t)b * a.o_bstride;
#pragma unroll
  for (int ps = 0; ps < 2 * MI; ++ps) {  // 32 rows x TC columns per pass: (row or column half, 32-row subtile)
    const int phalf = ps / MI, mi = ps % MI;
    // this thread's residual quads of the pass are fetched BEFORE the exchange (one workgroup fills the CU:
    // nothing else would cover their HBM latency); ragged tails take the scalar path below
    constexpr int NIT = (32 * NCOL + NTH - 1) / NTH;
    f32x4 pres[NIT];
    if (epi != EPI_STORE) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * NTH;
        const int row = idx / NCOL, qi = idx - row * NCOL;
        const int tau = qi / D, g = qi - tau * D;
        const int n0 = t0 + 4 * D * ((RH == 2 ? 0 : phalf * NTU) + tau) + 4 * g;
        if (idx < 32 * NCOL && n0 + 4 <= len) {
          const size_t ix = ob + (size_t)(mt * (32 * MI * RH) + (RH == 2 ? ps * 32 : mi * 32) + row) * a.ldo + n0;
          pres[it] = *reinterpret_cast<const f32x4*>(a.res + ix);
        }
      }
    }
    if (ps > 0) __syncthreads();
    if ((RH == 2 ? mh : chh) == phalf) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          yb[(p * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * YS + ni * 32 + l31] = acc[mi][ni][r];
    }
    __syncthreads();
    if constexpr (EPI2) {
    // A^T per COLUMN: one thread forms the four outputs of a column from its six Y values (sums and differences of the +-
    // point pairs shared: 10 operations for 4 outputs, six 4-byte LDS reads with the lanes on neighbouring columns) and
    // scatters them into an output tile in LDS; the quad loop below only adds bias / residual / MRF mode and stores.
    // (Per output QUAD straight from the Y tiles it was six 16-byte reads and 6 operations per OUTPUT, per-element
    // divisions by D for odd D: 9-23 % of a C = 64 layer, conv_wino8.hip has the before / after.)
    {
      constexpr int NCI = 32 * NCOL, NCT = (NCI + NTH - 1) / NTH;
#pragma unroll
      for (int it = 0; it < NCT; ++it) {
        const int idx = tid + it * NTH;
        if (idx >= NCI) continue;
        const int row = idx / NCOL, c = idx - row * NCOL;
        const int tau = c / D, rho = c - tau * D;
        const float* yc = yb + row * YS + c;
        const float y0 = yc[0], y1 = yc[32 * YS], y2 = yc[2 * 32 * YS], y3 = yc[3 * 32 * YS], y4 = yc[4 * 32 * YS],
                    y5 = yc[5 * 32 * YS];
        const float s12 = y1 + y2, d12 = y1 - y2;
        float* op = ot + row * OSW + 4 * D * tau + rho;
        op[0] = (y0 + s12) + (y3 + y4);
        op[D] = d12 + fmaf(2.f, y3, -0.5f * y4);
        op[2 * D] = s12 + fmaf(4.f, y3, 0.25f * y4);
        op[3 * D] = d12 + fmaf(8.f, y3, -0.125f * y4) + y5;
      }
    }
    __syncthreads();
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * NTH;
      if (idx >= 32 * NCOL) continue;
      const int row = idx / NCOL, qi = idx - row * NCOL;
      const int tau = qi / D, g = qi - tau * D;
      const int n0 = t0 + 4 * D * ((RH == 2 ? 0 : phalf * NTU) + tau) + 4 * g;
      if (n0 >= len) continue;
      const int grow = mt * (32 * MI * RH) + (RH == 2 ? ps * 32 : mi * 32) + row;
      const float bz = a.bias[grow];
      f32x4 v;
      const float* yr = yb + row * YS + tau * D;
      if constexpr (EPI2) {
        v = *reinterpret_cast<const f32x4*>(ot + row * OSW + 4 * D * tau + 4 * g);
        v[0] += bz; v[1] += bz; v[2] += bz; v[3] += bz;
      } else if constexpr (D == 1) {
        const float y0 = yr[0], y1 = yr[32 * YS], y2 = yr[2 * 32 * YS], y3 = yr[3 * 32 * YS], y4 = yr[4 * 32 * YS],
                    y5 = yr[5 * 32 * YS];
        const float s12 = y1 + y2, d12 = y1 - y2;
        v[0] = (y0 + s12) + (y3 + y4) + bz;
        v[1] = d12 + fmaf(2.f, y3, -0.5f * y4) + bz;
        v[2] = s12 + fmaf(4.f, y3, 0.25f * y4) + bz;
        v[3] = d12 + fmaf(8.f, y3, -0.125f * y4) + y5 + bz;
      } else if constexpr (D % 4 == 0) {
        // k = 11 (D = 4 d): the 4 outputs are 4 neighbouring columns of ONE A^T row -> six 16-byte LDS reads
        const int ao = (4 * g) / D, rho = 4 * g - ao * D;
        const float* yc = yr + rho;  // (tau D + rho) % 4 == 0 and YS % 4 == 0: aligned
        const f32x4 y0 = *reinterpret_cast<const f32x4*>(yc), y1 = *reinterpret_cast<const f32x4*>(yc + 32 * YS),
                    y2 = *reinterpret_cast<const f32x4*>(yc + 2 * 32 * YS), y3 = *reinterpret_cast<const f32x4*>(yc + 3 * 32 * YS),
                    y4 = *reinterpret_cast<const f32x4*>(yc + 4 * 32 * YS), y5 = *reinterpret_cast<const f32x4*>(yc + 5 * 32 * YS);
        const float sg = (ao & 1) ? -1.f : 1.f;
        const float p2 = __int_as_float((127 + ao) << 23), c4 = sg * __int_as_float((127 - ao) << 23);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = (ao == 0 ? y0[e] : 0.f) + y1[e];
          x = fmaf(sg, y2[e], x);
          x = fmaf(p2, y3[e], x);
          x = fmaf(c4, y4[e], x);
          x += (ao == 3 ? y5[e] : 0.f);
          v[e] = x + bz;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int u = 4 * g + e;
          const int ao = u / D, rho = u - ao * D;  // A^T row, column within the unit
          const float* yc = yr + rho;
          const float y0 = yc[0], y1 = yc[32 * YS], y2 = yc[2 * 32 * YS], y3 = yc[3 * 32 * YS], y4 = yc[4 * 32 * YS],
                      y5 = yc[5 * 32 * YS];
          // A^T[ao] = {ao == 0, 1, (-1)^ao, 2^ao, (-1/2)^ao, ao == 3}
          const float sg = (ao & 1) ? -1.f : 1.f;
          const float p2 = __int_as_float((127 + ao) << 23), ip2 = __int_as_float((127 - ao) << 23);
          float x = (ao == 0 ? y0 : 0.f) + y1;
          x = fmaf(sg, y2, x);
          x = fmaf(p2, y3, x);
          x = fmaf(sg * ip2, y4, x);
          x += (ao == 3 ? y5 : 0.f);
          v[e] = x + bz;
        }
      }
      const size_t ix = ob + (size_t)grow * a.ldo + n0;
      if (n0 + 4 <= len) {
        if (epi == EPI_STORE) {
          *reinterpret_cast<f32x4*>(a.out + ix) = v;
        } else {
          const f32x4 rs = pres[it];
          v[0] += rs[0]; v[1] += rs[1]; v[2] += rs[2]; v[3] += rs[3];
          if (epi == EPI_RES) {
            *reinterpret_cast<f32x4*>(a.out + ix) = v;
          } else if (epi == EPI_MRF_SET) {
            *reinterpret_cast<f32x4*>(a.acc + ix) = v;
          } else {
            const f32x4 ac = *reinterpret_cast<const f32x4*>(a.acc + ix);
            v[0] = ac[0] + v[0]; v[1] = ac[1] + v[1]; v[2] = ac[2] + v[2]; v[3] = ac[3] + v[3];
            if (epi == EPI_MRF_DIV) {
              v[0] = __fdiv_rn(v[0], a.mrf_div); v[1] = __fdiv_rn(v[1], a.mrf_div);
              v[2] = __fdiv_rn(v[2], a.mrf_div); v[3] = __fdiv_rn(v[3], a.mrf_div);
            }
            *reinterpret_cast<f32x4*>(a.acc + ix) = v;
          }
        }
      } else {
        for (int e = 0; e < len - n0; ++e) {
          float x = v[e];
          if (epi == EPI_STORE) {
            a.out[ix + e] = x;
          } else {
            x += a.res[ix + e];
            if (epi == EPI_RES) {
              a.out[ix + e] = x;
            } else if (epi == EPI_MRF_SET) {
              a.acc[ix + e] = x;
            } else {
              x = a.acc[ix + e] + x;
              if (epi == EPI_MRF_DIV) x = __fdiv_rn(x, a.mrf_div);
              a.acc[ix + e] = x;
            }
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// the generator's policy (per ResBlock): which (C, k) run in the transform domain
bool wino_wanted(int C, int KS) {
  if (!opts().wino || C < opts().wino_min_c) return false;
  if (C < 128 && KS < opts().wino_c64_kmin) return false;
  return wino_supported(C, C, KS, 1);
}

bool wino_supported(int Cout, int Cin, int KS, int dil) {
  // C in {64, 128, 256, 512}: every launch variant (large tiles and the small-grid RH = 2 / TW = 1 one, whose row-tile
  // count C / 64 must divide the 8 XCDs) exists for these; C = 1024 would only fail at forward time on short batches
  return Cout == Cin && (Cout == 64 || Cout == 128 || Cout == 256 || Cout == 512) && (KS == 3 || KS == 7 || KS == 11) && (dil == 1 || dil == 3 || dil == 5);
}

// w: [C][C][KS] -> transform-domain weights U[p][co][ci][j] = sum_i G[p][i] w[co][ci][j + NS i] as a grouped conv tensor
// [6 C][C][NS] (group = point), packed in A-fragment order for the 32x32x2 kernel
int make_wino(const float* w, const float* bias, int C, int KS, int dil, DevConv& dc) {
  const int NS = (KS + 2) / 3;
  std::vector<float> wt((size_t)6 * C * C * NS);
  for (int p = 0; p < 6; ++p)
    for (int co = 0; co < C; ++co)
      for (int ci = 0; ci < C; ++ci)
        for (int j = 0; j < NS; ++j) {
          double u = 0.0;
          for (int i = 0; i < 3; ++i) {
            const int tap = j + NS * i;
            if (tap < KS) u += kWinoG[p][i] * (double)w[((size_t)co * C + ci) * KS + tap];
          }
          wt[(((size_t)p * C + co) * C + ci) * NS + j] = (float)u;
        }
  // A-fragment order of the 32x32x2 MFMA, blocks in the order the kernel walks them:
  // [point][32-row subtile][chunk][half][tap][lane][k-step e]: W[32 ms + (lane & 31)][16 c + 8 half + 2 e + (lane >> 5)][tap]
  if (C % 32 != 0 || C % KC != 0) {
    set_error("make_wino: C = %d is not a multiple of the row tile", C);
    return DISSC_EINVAL;
  }
  const int nchunk = C / KC, nsub = C / 32;
  std::vector<float> packed((size_t)6 * nsub * nchunk * 2 * NS * 64 * 4);
  size_t o = 0;
  for (int p = 0; p < 6; ++p)
    for (int ms = 0; ms < nsub; ++ms)
      for (int c = 0; c < nchunk; ++c)
        for (int hf = 0; hf < 2; ++hf)
          for (int j = 0; j < NS; ++j)
            for (int lane = 0; lane < 64; ++lane)
              for (int e = 0; e < 4; ++e) {
                const int co = ms * 32 + (lane & 31), ci = c * KC + 8 * hf + 2 * e + (lane >> 5);
                packed[o++] = wt[(((size_t)p * C + co) * C + ci) * NS + j];
              }
  std::vector<float> b(C, 0.f);
  if (bias) memcpy(b.data(), bias, C * sizeof(float));
  dc.CIN = C; dc.M = C; dc.KS = KS; dc.dil = dil; dc.nchunk = nchunk; dc.up = 1;
  dc.groups = 1; dc.Mpad = C; dc.stride = 1; dc.pad_left = -1; dc.m32 = 1; dc.prec = 0;
  dc.macs_per_t = (double)C * C * KS;  // algorithmic (direct-form) MACs
  dc.wino = 1;
  int rc = upload(packed, &dc.wpack);
  if (rc) return rc;
  return upload(b, &dc.bias);
}

// MACs the matrix pipe executes per output position (6 NS / 4 per input/output channel pair)
double wino_executed_macs_per_t(int C, int KS) { return (double)C * C * 6.0 * ((KS + 2) / 3) / 4.0; }

template <int NS, int DIL, int CPR, int RH, int TW, int SHV = 0>
static int launch_wino_c(const WinoArgs& a, int B, int Lmax, hipStream_t stream) {
  constexpr int CHV = 3 - RH;
  constexpr int D = DIL * NS, NTU = 32 * TW / D, OT = 4 * D * NTU * CHV, RAW = OT + DIL * (3 * NS - 1), XRW = (RAW + 6) / 4 * 4;
  constexpr int RL = wino_row_len(D, NTU * CHV, XRW);
  size_t lds_f = (size_t)2 * CPR * D * RL + 12 * 8 * 112;
  const size_t epi_f = (size_t)6 * 32 * (32 * TW + 4) + ((DISSC_WINO_EPI2 && D != 1 && D % 4 != 0) ? (size_t)32 * (4 * D * NTU + 4) : 0);
  if (lds_f < epi_f) lds_f = epi_f;
  static DeviceOnce attr_once;  // per device (common.h)
  DISSC_HIP_CHECK(attr_once.max_lds(reinterpret_cast<const void*>(&conv_wino_kernel<NS, DIL, CPR, RH, TW, SHV>), 160 * 1024));
  WinoArgs aa = a;
  aa.gx = (Lmax + OT - 1) / OT;
  aa.gy = a.C / (32 * TW * RH);
  aa.B = B;
  if (8 % aa.gy != 0) {
    set_error("launch_wino: %d row tiles do not divide the 8 XCDs", aa.gy);
    return DISSC_EINVAL;
  }
  const int per = 8 / aa.gy;
  dim3 grid(8 * ((aa.gx * B + per - 1) / per));
  hipLaunchKernelGGL((conv_wino_kernel<NS, DIL, CPR, RH, TW, SHV>), grid, dim3(768), lds_f * sizeof(float), stream, aa);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

// option "wino_cpr" (Options::wino_cpr, default 32): "wino_cpr" option: channels per barrier round (16 or 32)
// option "wino_sv" (Options::wino_sv, default 1): "wino_sv" option: shared transform in the row-half form (C >= 128)
template <int NS, int DIL>
static int launch_wino_t(const WinoArgs& a, int B, int Lmax, hipStream_t stream) {
  // Small grids (a short or single utterance: the reference's one-at-a-time mode) step down to 32 x 32 wave tiles: four
  // times the workgroups.  Every output element sees the same MFMA sequence and the same transforms for either tile,
  // so the result is bit-identical and an utterance's samples stay independent of the batch it runs in.
  constexpr int D = DIL * NS;
  const bool c64 = a.C % 128 != 0;
  const int ot = 4 * D * (64 / D) * (c64 ? 2 : 1);
  const long long nwg = (long long)((Lmax + ot - 1) / ot) * (c64 ? a.C / 64 : a.C / 128) * B;
  const bool small = opts().small_grid && nwg < (long long)opts().wino_small;
  if (a.C == 32) return launch_wino_c<NS, DIL, 16, 1, 1>(a, B, Lmax, stream);  // diagnostics only (dissc_respair1d mode 2)
  if (c64) {
    if (small) return launch_wino_c<NS, DIL, 16, 1, 1>(a, B, Lmax, stream);  // 32 rows x 64 columns
    return launch_wino_c<NS, DIL, 16, 1, 2>(a, B, Lmax, stream);            // 64 rows x 128 columns
  }
  if (small) return launch_wino_c<NS, DIL, 16, 2, 1>(a, B, Lmax, stream);    // 64 rows x 32 columns
  // (k = 11, d = 5 with 32 channels per round needs more registers than three waves per SIMD leave: 16 there)
  const bool r32 = opts().wino_cpr == 32 && a.nchunk % 2 == 0 && !(NS == 4 && DIL == 5);
  if (opts().wino_sv) {
    // (with 32 channels per round the k = 7 / 11 instances spill a few registers; k = 3 has room)
    if (NS == 1 && r32) return launch_wino_c<NS, DIL, 32, 2, 2, 1>(a, B, Lmax, stream);
    return launch_wino_c<NS, DIL, 16, 2, 2, 1>(a, B, Lmax, stream);
  }
  if (r32) return launch_wino_c<NS, DIL, 32, 2, 2>(a, B, Lmax, stream);
  return launch_wino_c<NS, DIL, 16, 2, 2>(a, B, Lmax, stream);
}

int run_wino(const DevConv& dc, const float* x, float* out, const float* res, float* acc, const int32_t* lengths,
             int len_default, int len_mul, int B, int ldx, int ldo, int Lmax, float slope, int epi, float mrf_div,
             hipStream_t stream) {
  WinoArgs a;
  a.x = x; a.wpack = dc.wpack; a.bias = dc.bias; a.res = res; a.out = out; a.acc = acc;
  a.lengths = lengths; a.len_default = len_default; a.len_mul = len_mul;
  a.C = dc.M; a.nchunk = dc.nchunk; a.pad = (dc.KS - 1) * dc.dil / 2;
  a.ldx = ldx; a.ldo = ldo;
  a.x_bstride = (long long)dc.M * ldx; a.o_bstride = (long long)dc.M * ldo;
  a.slope = slope; a.mrf_div = mrf_div; a.epi = epi; a.dbg = opts().kernel_dbg;
  // the window staging, the residual / accumulator reads and the stores are 16-byte accesses
  auto misaligned = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) != 0; };
  if (ldx < 4 || ldx % 4 || ldo % 4 || misaligned(x) || misaligned(out) || misaligned(res) || misaligned(acc)) {
    set_error("run_wino: rows must be 16-byte aligned (ldx %d, ldo %d: multiples of 4 floats, ldx >= 4)", ldx, ldo);
    return DISSC_EINVAL;
  }
  if (B <= 0 || Lmax <= 0) {
    set_error("run_wino: empty batch (B %d, Lmax %d)", B, Lmax);
    return DISSC_EINVAL;
  }
  const int ns = (dc.KS + 2) / 3;
#define DISSC_WINO_CASE(NS_, D_) if (ns == NS_ && dc.dil == D_) return launch_wino_t<NS_, D_>(a, B, Lmax, stream);
  DISSC_WINO_CASE(1, 1) DISSC_WINO_CASE(1, 3) DISSC_WINO_CASE(1, 5)
  DISSC_WINO_CASE(3, 1) DISSC_WINO_CASE(3, 3) DISSC_WINO_CASE(3, 5)
  DISSC_WINO_CASE(4, 1) DISSC_WINO_CASE(4, 3) DISSC_WINO_CASE(4, 5)
#undef DISSC_WINO_CASE
  set_error("run_wino: k = %d, dilation %d unsupported", dc.KS, dc.dil);
  return DISSC_EINVAL;
}

}  // namespace dissc
