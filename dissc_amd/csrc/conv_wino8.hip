// conv_wino8_kernel: the k = 7 / 11 (and some k = 3) ResBlock convs of the C >= 64 stages (C -> C channels, dilation 1 / 3 / 5;
// reference sr/models.py:16-41) in a Toom-Cook transform domain of EIGHT points -- as F(6,3) (described first) or, template
// parameter R = 4, as F(5,4) (below); which shape takes which form: wino8_mask / wino8_r4_mask.  F(6,3): 6 outputs of a 3-tap filter from 8 products (evaluation
// points 0, +-1, +-2, +-1/2, inf) instead of F(4,3)'s 4 from 6 (conv_wino.hip) -- 8 ceil(k / 3) / 6 MFMA products per
// output, 11 % fewer -- still fp32 operands, fp32 products, fp32 accumulation on v_mfma_f32_32x32x2_f32.
//
// Why a second transform-domain kernel: EIGHT points map onto EIGHT waves, two per SIMD, and the tap loop of a
// 2-waves-per-SIMD workgroup sustains 138-145 TFLOP/s where conv_wino's 12 waves (three per SIMD) reach 122-126 with
// the same operand traffic (tools/ubench/mfma_mix8.hip: the global-load return path of the A fragments interacts with
// the MFMA issue of three waves, NOTES.md round 3 #14) -- and two waves per SIMD may hold a 128 x 64 accumulator tile.
//
//   y[n] = sum_i w[i] xa[n - pad + i d],  xa = lrelu(x) inside the utterance, 0 outside.
// The k taps are split into NS = ceil(k / 3) sub-filters of 3 taps with tap stride NS, D = d NS (conv_wino.hip); per point
//       Y_p[co][tau][rho] = sum_j sum_ci U_pj[co][ci] V_p[ci][tau][rho + d j],       rho in [0, D), j in [0, NS)
//       V_p[ci][tau][w]   = sum_q B^T[p][q] xa[ci][o + 6 D tau + w + D q],            w in [0, d (2 NS - 1)), q in [0, 8)
//       y[co][t0 + 6 D tau + rho + D a] = sum_p A^T[a][p] Y_p[co][tau][rho] + bias,   a in [0, 6)
// One workgroup = 8 waves = 8 points, every wave on the whole (32 MI rows) x (32 NI columns) tile of its point; the
// activated input window of 16 channels is staged like conv_wino's (registers -> lrelu / zero padding -> LDS in the
// polyphase layout [channel][x mod D][x div D], double-buffered, loads two rounds ahead); each wave forms V_p of 8
// channels for its point (8 neighbouring samples = four aligned 8-byte reads) into a private tile; A fragments streamed
// from L2 one tap ahead; epilogue: the 8 waves exchange Y_p through LDS 32 rows at a time, one thread applies A^T, bias
// and the residual / MRF mode to 4 consecutive outputs.
// R = 4 (template parameter; "wino8_r4" option): the same eight points used as F(5,4) -- FIVE outputs of a FOUR-tap sub-filter
// from the same 8 products, NS = ceil(k / 4): 8 NS / 5 = 3.2 products per output for k = 7 (4.0 as F(6,3), 7 direct), 4.8 for
// k = 11 (5.33 / 11).  B^T depends on the points only, so the transform is the same; a unit advances 5 D samples instead of
// 6 D (its 8-byte window reads are then only 4-byte aligned), G has a fourth column and A^T five rows.
// Rounding: per layer 1.8x the F(4,3) form's error (numpy model of a C = 256 layer, k = 11: rms 1.5e-7 vs 8.3e-8 on O(0.3)
// outputs, direct 9.3e-8); DESIGN.md section 4.
#include <string.h>

#include <type_traits>

#include "common.h"
#include "conv_epilogue32.h"

namespace dissc {

// option "wino8" (Options::wino8, default 1): "wino8" option (read at dissc_gen_create): 1 (default) = the k = 7 / 11 ResBlock convs the "wino8_mask" names use
                      // this kernel instead of conv_wino's F(4,3) form; 0 = none; 2 = the stand-alone dissc_conv1d entry uses it too
                      // (tests).  Per launch it is 3-17 % faster than the F(4,3) form on 26 of the 36 (C, k, d, epilogue) shapes of the
                      // generator (tools/wino8_gate.py) and 4-9 % slower on the d = 1 shapes of the 128-channel stage (864 workgroups
                      // = 3.4 rounds of 256 CUs where the F(4,3) tiles make exactly 5.0): the default mask leaves that stage alone.
                      // Whole forward, same box, two runs each: 35.17 / 35.27 -> 34.84 / 34.91 ms (1.0 %), executed-FLOP utilisation
                      // 0.62 -> 0.60 (11 % fewer products on those layers in 1 % less time), in-run parity rms 5.5e-7 -> 6.5e-7.
// option "kernel_dbg": diagnostics: knock-outs, bit 0 transform, 1 MFMAs, 2 epilogue
// option "wino8_c64_wide" (Options::wino8_c64_wide, default 3): "wino8_c64_wide" option, C = 64 instances: 1 = 64 x 128 tiles (768 outputs), 0 = 64 x 64, 2 = 64 x 64 built for TWO
                           // workgroups per CU (<= 128 registers, <= 80 KB LDS: their phases overlap; +3-9 % on k = 11, mixed on
                           // k = 7 as F(6,3), +2-8 % on k = 7 as F(5,4)), 3 (default) = 1 for k = 7 as F(6,3), 2 otherwise (run_wino8)

struct Wino8Args {
  const float* x;
  const float* wpack;   // [8 points][C / 32][nchunk][2 halves][NS taps][64 lanes][4 k-steps] (make_wino8)
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
  int gx, gy, B;  // time tiles, row tiles, utterances
};

// G (host, double) at the points 0, 1, -1, 2, -2, 1/2, -1/2, inf: row p = scale_p * (1, x_p, x_p^2 [, x_p^3]) for the finite points,
// the last unit vector for inf -- three columns as F(6,3), four as F(5,4); B^T and A^T are written out below
static const double kW8Pt[7] = {0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5};
static const double kW8Scale[7] = {-1.0, -2.0 / 9.0, -2.0 / 9.0, 1.0 / 90.0, 1.0 / 90.0, 32.0 / 45.0, 32.0 / 45.0};
static double w8_g(int p, int i, int R) {
  if (p == 7) return i == R - 1 ? 1.0 : 0.0;
  double v = kW8Scale[p];
  for (int e = 0; e < i; ++e) v *= kW8Pt[p];
  return v;
}

// row P of B^T (all entries exactly representable) applied to eight neighbouring samples
template <int P>
__device__ __forceinline__ float w8_bt(float r0, float r1, float r2, float r3, float r4, float r5, float r6, float r7) {
  if constexpr (P == 0) return fmaf(5.25f, r2 - r4, r6 - r0);
  if constexpr (P == 1) return fmaf(-4.25f, r3 + r4, (r1 + r2) + (r5 + r6));
  if constexpr (P == 2) return fmaf(4.25f, r3 - r4, (r2 - r1) + (r6 - r5));
  if constexpr (P == 3) return fmaf(0.5f, r1, fmaf(0.25f, r2, fmaf(-2.5f, r3, fmaf(-1.25f, r4, fmaf(2.f, r5, r6)))));
  if constexpr (P == 4) return fmaf(-0.5f, r1, fmaf(0.25f, r2, fmaf(2.5f, r3, fmaf(-1.25f, r4, fmaf(-2.f, r5, r6)))));
  if constexpr (P == 5) return fmaf(2.f, r1, fmaf(4.f, r2, fmaf(-2.5f, r3, fmaf(-5.f, r4, fmaf(0.5f, r5, r6)))));
  if constexpr (P == 6) return fmaf(-2.f, r1, fmaf(4.f, r2, fmaf(2.5f, r3, fmaf(-5.f, r4, fmaf(-0.5f, r5, r6)))));
  if constexpr (P == 7) return fmaf(5.25f, r3 - r5, r7 - r1);
  return 0.f;
}

constexpr int w8_rup(int n, int m) { return (n + m - 1) / m * m; }

// polyphase row length >= need (even): lane = tau * D + phi of the transform reads 8 bytes at float offset phi * RL + MO tau
// (+ 2 q); a ds_read_b64 is serviced 32 lanes at a time over 64 banks of 4 bytes -- pick the RL (<= need + 62) whose worst
// 32-lane group touches the fewest lanes per bank pair
constexpr int w8_row_len(int D, int NTU, int need, int MO) {
  int best = need, best_cost = 1 << 30;
  for (int rl = need; rl <= need + 62; rl += 2) {
    int cost = 0;
    for (int g0 = 0; g0 < 64; g0 += 32) {
      int cnt[64] = {0};  // lanes per 4-byte bank (MO = 5: the reads start on odd floats too)
      for (int l = g0; l < g0 + 32; ++l) {
        const int c = l < NTU * D ? l : NTU * D - 1;
        const int at = (c % D) * rl + MO * (c / D);
        ++cnt[at % 64];
        ++cnt[(at + 1) % 64];
      }
      for (int i = 0; i < 64; ++i)
        if (cnt[i] > cost) cost = cnt[i];
    }
    if (cost < best_cost) { best_cost = cost; best = rl; }
    if (cost <= 1) break;
  }
  return best;
}

// WPS: waves per SIMD the instance is built for -- 2: one 8-wave workgroup per CU (up to 256 registers); 4: TWO workgroups
// per CU (<= 128 registers and <= 80 KB of LDS each: windows of 8 channels, epilogue passes of 16 rows), whose phases overlap
// R: taps per sub-filter (3: F(6,3), 4: F(5,4)); MO = 9 - R outputs per unit
template <int NS, int DIL, int MI, int NI, int WPS = 2, int R = 3>
struct Wino8Geo {
  static constexpr int NTH = 512;
  static constexpr int MO = 9 - R;
  static constexpr int D = DIL * NS, W = DIL * (2 * NS - 1);
  static constexpr int NCW = 32 * NI;
  static constexpr int NTU0 = NCW / D;
  // units per tile.  F(6,3): an odd D takes an even count, so that the tile is a whole number of output quads; F(5,4) takes
  // them all (8 tiles instead of 9 on a 2 500-sample row) and stores the quads it shares with its neighbours element by element
  static constexpr int NTU = (R == 3 && D % 2 == 1 && NTU0 % 2 == 1) ? NTU0 - 1 : NTU0;
  static constexpr int NCOL = NTU * D;
  static constexpr int OT = MO * D * NTU;              // outputs per workgroup tile
  static constexpr int RAW = OT + (R - 2) * D + W;      // input samples the transforms touch: units of MO D, 8 samples D apart, W entries
  static constexpr int XRW = w8_rup(RAW + 3 + 3, 4);    // staged positions per channel (alignment shift <= 3)
  static constexpr int NV = XRW / 4;
  static constexpr int OFF = D >= 2 ? 2 : 4;            // polyphase index offset (even, OFF D >= 3): window sample x at [(x + OFF D) % D][(x + OFF D) / D]
  static constexpr int RL_MIN = w8_rup(((XRW - 1 + OFF * D) / D + 1) > (MO * NTU + OFF + 10) ? ((XRW - 1 + OFF * D) / D + 1) : (MO * NTU + OFF + 10), 2);
  static constexpr int RL = w8_row_len(D, NTU, RL_MIN, MO);  // bumped so that the 8-byte transform reads of a half wave spread over the banks
  static constexpr int CHF = D * RL;
  // Latency form (NI = 1: the small-grid tiers -- one workgroup per CU at most, two waves per SIMD, so nothing hides a wave's
  // chains): see the kernel's `if constexpr (LAT)` main loop.
  static constexpr bool LAT = NI == 1;
  static constexpr int CPR = LAT ? 16 : (NI == 4 || WPS == 4) ? 8 : 16;   // channels of the window per round (LDS budget)
  static constexpr int RPP = (!LAT && (NI == 4 || WPS == 4)) ? 16 : 32;  // rows per epilogue pass (likewise)
  static constexpr int CG = NCW / 64 > 0 ? NCW / 64 : 1;  // 64-column groups a lane transforms
  static constexpr int XV = (LAT && NTU * W <= 80) ? 80 : NTU * W <= 112 ? 112 : 240; // V row stride (% 32 == 16: the two k halves of a fragment read hit different banks)
  static constexpr int YS = NCW + 4;
  static constexpr int WIN_FLOATS = 2 * CPR * CHF;
  static constexpr int V_FLOATS = 8 * 8 * XV * (LAT ? 2 : 1);  // (latency form: two V tiles per wave)
  static constexpr int Y_FLOATS = 8 * RPP * YS;
  static constexpr int OS = w8_rup(OT, 4) + 4;          // row stride of the epilogue's output tile (16-byte aligned rows)
  static constexpr int EPI_FLOATS = Y_FLOATS + RPP * OS;
  static constexpr int LDS_FLOATS = (WIN_FLOATS + V_FLOATS) > EPI_FLOATS ? (WIN_FLOATS + V_FLOATS) : EPI_FLOATS;
  static_assert(NTU >= 1 && NTU * W <= XV && RL % 2 == 0, "tile geometry");
  static_assert(LDS_FLOATS * 4 <= ((WPS == 4 && !LAT) ? 80 : 160) * 1024, "LDS");
};

template <int NS, int DIL, int MI, int NI, int WPS = 2, int R = 3>
__global__ void __launch_bounds__(512, (NI == 1 ? 2 : WPS)) conv_wino8_kernel(const Wino8Args a) {
  using G = Wino8Geo<NS, DIL, MI, NI, WPS, R>;
  constexpr bool LAT = G::LAT;
  constexpr int NTH = G::NTH, D = G::D, W = G::W, NCOL = G::NCOL, OT = G::OT, NV = G::NV, RL = G::RL, MO = G::MO,
                CHF = G::CHF, CPR = G::CPR, XV = G::XV, YS = G::YS, OFF = G::OFF, RPP = G::RPP, CG = G::CG;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const vbuf = lds + G::WIN_FLOATS;  // [8 waves][8][XV]

  // 1-D grid, XCD-aware (conv_wino.hip): row tiles pinned to XCD groups, time tiles / utterances spread inside a group
  const int per = 8 / a.gy;
  const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
  const int mt = xcd / per;
  const int lin = (xcd - mt * per) + per * kq;
  if (lin >= a.gx * a.B) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  int b = -1, len = a.len_default, t0 = 0;
  if (a.lengths == nullptr) {
    b = lin / a.gx;
    t0 = (lin - b * a.gx) * OT;
    if (t0 >= len) return;
  } else {  // only the tiles that exist are enumerated (conv_wino.hip)
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
  const int p = __builtin_amdgcn_readfirstlane(tid >> 6);  // this wave's evaluation point
  const int l31 = lane & 31, h = lane >> 5;
  const int o = t0 - a.pad;
  const int tb = o & ~3;
  const int sh = o - tb;
  const float slope = a.slope;
  const float* xb = a.x + (size_t)b * a.x_bstride;
  const int nround = a.C / CPR;

  // ---- window staging (conv_wino.hip): clamped 16-byte loads; activation, zero padding and the polyphase scatter on the way to LDS
  const int r0 = tid / NV, v0 = tid - r0 * NV;
  constexpr int dr = NTH / NV, dv = NTH - dr * NV;
  constexpr int SV = (CPR * NV + NTH - 1) / NTH;
  f32x4 sv[SV];
  auto stage_load = [&](int rd) __attribute__((always_inline)) {
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
  auto stage_store = [&](float* raw) __attribute__((always_inline)) {
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
          const int xs = 4 * v + e - sh + OFF * D;  // >= OFF D - 3 >= 0
          rowp[(xs % D) * RL + xs / D] = ok ? lrelu(val[e], slope) : 0.f;
        }
      }
      v += dv;
      r += dr;
      if (v >= NV) { v -= NV; ++r; }
    }
  };

  // ---- transform: lane <-> column (tau, phi) (+ 64 per column group); entries w = phi and w = phi + D of unit tau: samples
  // MO tau + OFF (+ 1) + q
  int toff[CG], e0[CG], e1[CG];
  bool ok1[CG];
#pragma unroll
  for (int g = 0; g < CG; ++g) {
    const int c0 = lane + 64 * g;
    const int tcol = c0 < NCOL ? c0 : NCOL - 1;
    const int ttau = tcol / D, tphi = tcol % D;
    toff[g] = tphi * RL + MO * ttau + OFF;  // F(6,3): even, 8-byte aligned reads
    e0[g] = ttau * W + tphi;
    ok1[g] = tphi + D < W;
    e1[g] = ok1[g] ? e0[g] + D : e0[g];
  }
  float* const vp = vbuf + p * (8 * XV);
  typedef float f32x2a __attribute__((ext_vector_type(2)));
  auto transform8 = [&](const float* raw8) __attribute__((always_inline)) {
    if (a.dbg & 1) return;
    auto go = [&](auto pc) __attribute__((always_inline)) {
      constexpr int P = decltype(pc)::value;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int g = 0; g < CG; ++g) {
          const float* rw = raw8 + i * CHF + toff[g];
          f32x2a q0, q1, q2, q3;
          if constexpr (MO % 2 == 0) {
            q0 = *reinterpret_cast<const f32x2a*>(rw); q1 = *reinterpret_cast<const f32x2a*>(rw + 2);
            q2 = *reinterpret_cast<const f32x2a*>(rw + 4); q3 = *reinterpret_cast<const f32x2a*>(rw + 6);
          } else {  // (F(5,4): units start on odd floats too -- scalar reads, which the compiler pairs as far as 4-byte alignment allows)
            q0[0] = rw[0]; q0[1] = rw[1]; q1[0] = rw[2]; q1[1] = rw[3]; q2[0] = rw[4]; q2[1] = rw[5]; q3[0] = rw[6]; q3[1] = rw[7];
          }
          const float v0 = w8_bt<P>(q0[0], q0[1], q1[0], q1[1], q2[0], q2[1], q3[0], q3[1]);
          vp[i * XV + e0[g]] = v0;
          if constexpr (NS > 1) {
            const float r8 = rw[8];
            const float v1 = w8_bt<P>(q0[1], q1[0], q1[1], q2[0], q2[1], q3[0], q3[1], r8);
            vp[i * XV + e1[g]] = ok1[g] ? v1 : v0;
          }
        }
      }
    };
    switch (p) {  // uniform per wave
      case 0: go(std::integral_constant<int, 0>{}); break;
      case 1: go(std::integral_constant<int, 1>{}); break;
      case 2: go(std::integral_constant<int, 2>{}); break;
      case 3: go(std::integral_constant<int, 3>{}); break;
      case 4: go(std::integral_constant<int, 4>{}); break;
      case 5: go(std::integral_constant<int, 5>{}); break;
      case 6: go(std::integral_constant<int, 6>{}); break;
      default: go(std::integral_constant<int, 7>{}); break;
    }
  };

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

  // A fragments: one float4 per lane = the 4 k-steps (8 channels) of one (chunk, half, tap) block, packed in the order the
  // loop walks them: [point][32-row subtile][block][lane]
  const int nsub = a.C / 32;
  const int nblk = a.nchunk * NS * 2;
  f32x4 av[MI], avn[MI];
#ifndef DISSC_WINO8_BUF
#define DISSC_WINO8_BUF 1
#endif
#if DISSC_WINO8_BUF
  // scalar-base loads (common.h: wave_rsrc): this wave's MI subtiles are one contiguous slab [mi][block][lane]
  const __amdgpu_buffer_rsrc_t wrs = wave_rsrc(reinterpret_cast<const f32x4*>(a.wpack) + ((size_t)p * nsub + mt * MI) * nblk * 64,
                                               (unsigned)(MI * nblk) * 1024u);
  const unsigned lane16 = lane * 16u;
  auto a_load = [&](int mi, int bl) __attribute__((always_inline)) { return rsrc_load16(wrs, lane16, (unsigned)(mi * nblk + bl) * 1024u); };
#else
  const f32x4* wp[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
    wp[mi] = reinterpret_cast<const f32x4*>(a.wpack) + ((size_t)p * nsub + mt * MI + mi) * nblk * 64 + lane;
  auto a_load = [&](int mi, int bl) __attribute__((always_inline)) { return wp[mi][(size_t)bl * 64]; };
#endif
  if constexpr (!LAT) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) av[mi] = a_load(mi, 0);
  }
  stage_load(0);
  stage_store(lds);
  if (nround > 1) stage_load(1);
  __syncthreads();

  if constexpr (LAT) {
    // ---- latency form.  On a small grid a CU holds ONE workgroup: two waves per SIMD, in step from barrier to barrier, so the
    // loop below runs its phases back to back (B = 1, C = 256, k = 11, 136 workgroups: 66.7 us = 15 fixed + 29 transform + 24 MFMAs,
    // tools/wino8_b1_ko.py on the round-5 form) and a 32 x 32 tile gives its one-block weight prefetch 4 MFMAs = 0.1 us to cover
    // an L2 / MALL miss.  Here
    //  * both half waves work on the tile's 32 columns: half h transforms channels 4h .. 4h + 3 of the sub-chunk (the other
    //    form leaves lanes 32-63 repeating the last column): half the transform's instructions per wave -- 66.7 -> 47.6 us;
    //  * the transform of sub-chunk s + 1 goes into the wave's OTHER V tile while sub-chunk s multiplies (no LDS round trip
    //    between a sub-chunk's transform and its first MFMA);
    //  * the weight fragments of a whole round are in flight: slot (sub-chunk, tap block), refilled for the next round right
    //    after its MFMAs;
    //  * the round's barrier moves before its last sub-chunk (whose companion transform reads the NEXT round's window).
    // Same w8_bt arithmetic per value and the same MFMA sequence per accumulator: bit-identical to the other form.
    // (Knock-outs `kernel_dbg` bits 0 / 1 do not apply here.)
    constexpr int SCR = CPR / 8;
    static_assert(SCR == 2, "latency form: two sub-chunks per round");
    const int tcol = l31 < NCOL ? l31 : NCOL - 1;
    const int toffl = (tcol % D) * RL + MO * (tcol / D) + OFF + 4 * h * CHF;
    const int e0l = (tcol / D) * W + tcol % D + 4 * h * XV;
    const bool ok1l = tcol % D + D < W;
    const int e1l = ok1l ? e0l + D : e0l;
    float* const vt[2] = {vbuf + p * (8 * XV), vbuf + 8 * 8 * XV + p * (8 * XV)};
    float q[4][9];
    auto tr_read = [&](const float* raw8) __attribute__((always_inline)) {
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) {
        const float* rw = raw8 + ii * CHF + toffl;
        if constexpr (MO % 2 == 0) {
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            const f32x2a t = *reinterpret_cast<const f32x2a*>(rw + e);
            q[ii][e] = t[0];
            q[ii][e + 1] = t[1];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) q[ii][e] = rw[e];
        }
        q[ii][8] = NS > 1 ? rw[8] : 0.f;
      }
    };
    auto tr_piece = [&](auto pc, float* vdst, int ii) __attribute__((always_inline)) {  // channel ii of this half wave's four
      constexpr int P = decltype(pc)::value;
      const float v0 = w8_bt<P>(q[ii][0], q[ii][1], q[ii][2], q[ii][3], q[ii][4], q[ii][5], q[ii][6], q[ii][7]);
      vdst[ii * XV + e0l] = v0;
      if constexpr (NS > 1) {
        const float v1 = w8_bt<P>(q[ii][1], q[ii][2], q[ii][3], q[ii][4], q[ii][5], q[ii][6], q[ii][7], q[ii][8]);
        vdst[ii * XV + e1l] = ok1l ? v1 : v0;
      }
    };
    f32x4 aq[SCR][NS][MI];
#pragma unroll
    for (int sc = 0; sc < SCR; ++sc)
#pragma unroll
      for (int j = 0; j < NS; ++j)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) aq[sc][j][mi] = a_load(mi, sc * NS + j);
    int blk = 0;
    // One sub-chunk: the MFMAs of sub-chunk SC (V tile SC & 1) and the transform of the next one (from raw_next into the other
    // V tile).  The order inside does not matter (measured, C = 256, k = 11, B = 1: 47.6 us as written; 46.8 with the SIMD's two
    // waves -- points P and P + 4, tools/ubench/wave_simd.hip -- in opposite phase order; 49.4 with one channel's transform
    // pinned between the chain's MFMAs; 48.8 with the MFMA operands read before the window): a SIMD issues the transform's
    // VALU / LDS instructions and the MFMAs' passes from one budget, the two times add (knock-outs: 20 us of MFMAs = the
    // tile's floor at two chains per SIMD, 17 us of transform, 11 us fixed).  What helped was fewer instructions per wave.
    auto slot = [&](auto pc, auto scc, const float* raw_next) __attribute__((always_inline)) {
      constexpr int SC = decltype(scc)::value;
      tr_read(raw_next);
      const float* bj = vt[SC & 1] + voff[0];
#pragma unroll
      for (int j = 0; j < NS; ++j, ++blk) {
        float bk[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) bk[s] = bj[s * 2 * XV + j * DIL];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[SC][j][mi][s], bk[s], acc[mi][0], 0, 0, 0);
        const int bn = blk + SCR * NS < nblk ? blk + SCR * NS : nblk - 1;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) aq[SC][j][mi] = a_load(mi, bn);
      }
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) tr_piece(pc, vt[(SC + 1) & 1], ii);
    };
    auto body = [&](auto pc) __attribute__((always_inline)) {
      tr_read(lds);
#pragma unroll
      for (int ii = 0; ii < 4; ++ii) tr_piece(pc, vt[0], ii);
      for (int rd = 0; rd < nround; ++rd) {
        const float* raw = lds + (rd & 1) * (CPR * CHF);
        float* rawn = lds + ((rd + 1) & 1) * (CPR * CHF);
        if (rd + 1 < nround) stage_store(rawn);
        if (rd + 2 < nround) stage_load(rd + 2);
        slot(pc, std::integral_constant<int, 0>{}, raw + 8 * CHF);
        __syncthreads();  // every wave is done reading this round's window; the next one is complete
        slot(pc, std::integral_constant<int, 1>{}, rawn);  // (after the last round: a transform nobody reads)
      }
    };
    switch (p) {  // uniform per wave
      case 0: body(std::integral_constant<int, 0>{}); break;
      case 1: body(std::integral_constant<int, 1>{}); break;
      case 2: body(std::integral_constant<int, 2>{}); break;
      case 3: body(std::integral_constant<int, 3>{}); break;
      case 4: body(std::integral_constant<int, 4>{}); break;
      case 5: body(std::integral_constant<int, 5>{}); break;
      case 6: body(std::integral_constant<int, 6>{}); break;
      default: body(std::integral_constant<int, 7>{}); break;
    }
    __syncthreads();
  }

  int blk = 0;
  auto run_taps = [&]() __attribute__((always_inline)) {
    if (a.dbg & 2) {
      blk += NS;
      return;
    }
    const float* bj[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) bj[ni] = vp + voff[ni];
#pragma unroll
    for (int j = 0; j < NS; ++j, ++blk) {
      const int bn = (blk + 1 < nblk) ? blk + 1 : nblk - 1;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) avn[mi] = a_load(mi, bn);
      float bk[4][NI];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bk[s][ni] = bj[ni][s * 2 * XV + j * DIL];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][s], bk[s][ni], acc[mi][ni], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) av[mi] = avn[mi];
    }
  };

  // One barrier per round: at the top of round rd the window of round rd + 1 goes into the other buffer (every wave finished
  // reading it before the barrier that closed round rd - 1) and the loads of round rd + 2 are issued
  for (int rd = 0; rd < (LAT ? 0 : nround); ++rd) {
    const float* raw = lds + (rd & 1) * (CPR * CHF);
    if (rd + 1 < nround) stage_store(lds + ((rd + 1) & 1) * (CPR * CHF));
    if (rd + 2 < nround) stage_load(rd + 2);
#pragma unroll
    for (int sc = 0; sc < CPR / 8; ++sc) {
      transform8(raw + (sc * 8) * CHF);
      run_taps();
    }
    __syncthreads();
  }

  // ---- epilogue, 32 rows at a time: the 8 waves put their Y_p into LDS; one thread per (row, column) applies A^T -- all
  // six outputs of the column from its eight Y values, sharing the sums and differences of the +- point pairs (22
  // operations for 6 outputs; eight 4-byte LDS reads, lanes on neighbouring columns) -- and scatters them into an output
  // tile in LDS; then one thread per 4 consecutive outputs adds bias / residual / the MRF mode and stores 16 bytes.
  // (The first version applied A^T per output quad straight from the Y tiles: eight 16-byte reads and ~10 operations per
  // OUTPUT, with divisions by D for odd D; the epilogue was 16-21 % of a k = 7 / C = 64 layer.)
  float* const yb = lds;                 // [8][RPP][YS]
  float* const ot = lds + G::Y_FLOATS;   // [RPP][OS]
  if (a.dbg & 4) {
    if (acc[0][0][0] == 123.f) a.out[0] = 1.f;
    return;
  }
  const int epi = a.epi;
  const size_t ob = (size_t)b * a.o_bstride;
  constexpr int OS = G::OS;
  // Output quads are aligned in ABSOLUTE time.  A tile of OT outputs with OT % 4 != 0 (F(5,4) with an odd unit count) starts
  // qsh = t0 & 3 samples into its first quad: that quad and the last one are shared with the neighbouring tiles and stored
  // element by element, the full ones read their four values from the output tile with 4-byte reads.
  constexpr bool AL = OT % 4 == 0;
  constexpr int QPR = AL ? OT / 4 : (OT + 3) / 4 + 1;  // quads a row can touch
  constexpr int NQ = RPP * QPR, NIT = (NQ + NTH - 1) / NTH;
  const int qsh = AL ? 0 : (t0 & 3);
  const int tend = (t0 + OT < len) ? t0 + OT : len;     // this tile's outputs: [t0, tend)
  constexpr int NCI = RPP * NCOL, NCT = (NCI + NTH - 1) / NTH;
  constexpr int PS = RPP * YS;  // point stride
  constexpr int NPASS = 32 * MI / RPP;
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const int mi = ps * RPP / 32;          // compile-time after unrolling
    const int sp = (ps * RPP % 32) / 16;   // which 16-row half of the 32-row block (RPP = 16)
    // residual quads of the pass -- and, in the read-modify-write MRF modes, the accumulator's -- fetched before the exchange.
    // (Loaded inside the store loop the accumulator quads cannot be hoisted over the previous iterations' stores to the same
    // array: NIT dependent round trips to HBM per pass, 47-112 us per MRF launch of the 128-channel stage.)
    f32x4 pres[NIT], pacc[NIT];
    const bool rmw = epi != EPI_STORE && epi != EPI_RES && epi != EPI_MRF_SET;
    if (epi != EPI_STORE) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * NTH;
        const int row = idx / QPR, qi = idx - row * QPR;
        const int n0 = t0 - qsh + 4 * qi;
        if (idx < NQ && n0 >= t0 && n0 + 4 <= tend) {
          const size_t ixp = ob + (size_t)(mt * (32 * MI) + ps * RPP + row) * a.ldo + n0;
          pres[it] = *reinterpret_cast<const f32x4*>(a.res + ixp);
          if (rmw) pacc[it] = *reinterpret_cast<const f32x4*>(a.acc + ixp);
        }
      }
    }
    if (ps > 0) __syncthreads();  // the previous pass has read its output tile and its Y tiles
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if constexpr (RPP == 32) {
          yb[(p * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * YS + ni * 32 + l31] = acc[mi][ni][r];
        } else {
          if ((r >> 3) == sp) yb[(p * 16 + (r & 3) + 8 * ((r >> 2) & 1) + 4 * h) * YS + ni * 32 + l31] = acc[mi][ni][r];
        }
      }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NCT; ++it) {
      const int idx = tid + it * NTH;
      if (idx >= NCI) continue;
      const int row = idx / NCOL, c = idx - row * NCOL;
      const int tau = c / D, rho = c - tau * D;
      const float* yc = yb + row * YS + c;
      const float y0 = yc[0], y1 = yc[PS], y2 = yc[2 * PS], y3 = yc[3 * PS], y4 = yc[4 * PS], y5 = yc[5 * PS], y6 = yc[6 * PS],
                  y7 = yc[7 * PS];
      const float s12 = y1 + y2, d12 = y1 - y2, s34 = y3 + y4, d34 = y3 - y4, s56 = y5 + y6, d56 = y5 - y6;
      float* op = ot + row * OS + MO * D * tau + rho;
      op[0] = (y0 + s12) + (s34 + s56);
      op[D] = fmaf(2.f, d34, fmaf(0.5f, d56, d12));
      op[2 * D] = fmaf(4.f, s34, fmaf(0.25f, s56, s12));
      op[3 * D] = fmaf(8.f, d34, fmaf(0.125f, d56, d12));
      if constexpr (MO == 6) {
        op[4 * D] = fmaf(16.f, s34, fmaf(0.0625f, s56, s12));
        op[5 * D] = fmaf(32.f, d34, fmaf(0.03125f, d56, d12)) + y7;
      } else {  // F(5,4): the point at infinity belongs to the fifth (last) output
        op[4 * D] = fmaf(16.f, s34, fmaf(0.0625f, s56, s12)) + y7;
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + it * NTH;
      if (idx >= NQ) continue;
      const int row = idx / QPR, qi = idx - row * QPR;
      const int n0 = t0 - qsh + 4 * qi;
      const int lo = n0 > t0 ? n0 : t0, hi = n0 + 4 < tend ? n0 + 4 : tend;  // this tile's elements of the quad: [lo, hi)
      if (hi <= lo) continue;
      const int grow = mt * (32 * MI) + ps * RPP + row;
      const float bz = a.bias[grow];
      f32x4 v;
      if constexpr (AL) {
        v = *reinterpret_cast<const f32x4*>(ot + row * OS + 4 * qi);
      } else {
        const float* src = ot + row * OS + (4 * qi - qsh);  // (elements outside [lo, hi) are read as far as the row reaches, never used)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (4 * qi - qsh + e >= 0) ? src[e] : 0.f;
      }
      v[0] += bz; v[1] += bz; v[2] += bz; v[3] += bz;
      const size_t ix = ob + (size_t)grow * a.ldo + n0;
      if (hi - lo == 4) {
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
            const f32x4 ac = pacc[it];
            v[0] = ac[0] + v[0]; v[1] = ac[1] + v[1]; v[2] = ac[2] + v[2]; v[3] = ac[3] + v[3];
            if (epi == EPI_MRF_DIV) {
              v[0] = __fdiv_rn(v[0], a.mrf_div); v[1] = __fdiv_rn(v[1], a.mrf_div);
              v[2] = __fdiv_rn(v[2], a.mrf_div); v[3] = __fdiv_rn(v[3], a.mrf_div);
            }
            *reinterpret_cast<f32x4*>(a.acc + ix) = v;
          }
        }
      } else {
        // a quad shared with the neighbouring tile, or cut by the utterance's end: element by element -- every load first (loaded
        // inside the store loop, an element's residual could not be hoisted over the previous element's store: up to three
        // dependent HBM round trips per quad, +50-90 us on the F(5,4) layers whose tiles are not whole quads)
        float rr[4], aa[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = n0 + e >= lo && n0 + e < hi;
          rr[e] = (ok && epi != EPI_STORE) ? a.res[ix + e] : 0.f;
          aa[e] = (ok && rmw) ? a.acc[ix + e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (n0 + e < lo || n0 + e >= hi) continue;
          float x = v[e];
          if (epi == EPI_STORE) {
            a.out[ix + e] = x;
          } else {
            x += rr[e];
            if (epi == EPI_RES) {
              a.out[ix + e] = x;
            } else if (epi == EPI_MRF_SET) {
              a.acc[ix + e] = x;
            } else {
              x = aa[e] + x;
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
bool wino8_supported(int Cout, int Cin, int KS, int dil) {
  if (KS == 3 && !DISSC_EXPERIMENTAL && !(Cout == 64 && dil == 1)) return false;  // (see run_wino8)
  return Cout == Cin && (Cout == 64 || Cout == 128 || Cout == 256 || Cout == 512) && (KS == 3 || KS == 7 || KS == 11) &&
         (dil == 1 || dil == 3 || dil == 5);
}

// Which form a ResBlock conv of the C >= 64 stages takes is decided per SHAPE (stage width class, kernel size, dilation), from the
// per-launch measurements of tools/wino8_gate.py and same-box forward A/Bs (tools/opt_ab.sh): bit 9 cls + 3 ki + di with
// cls = 0 / 1 / 2 for C = 64 / 128 / >= 256, ki = 0 / 1 / 2 for k = 3 / 7 / 11, di = 0 / 1 / 2 for dilation 1 / 3 / 5 -- in octal
// three digits per class (k = 11, k = 7, k = 3 from the left), each digit = the dilations d5 d3 d1.
static int w8_shape_bit(int C, int KS, int dil) {
  const int cls = C >= 256 ? 2 : C >= 128 ? 1 : 0;
  return 9 * cls + 3 * (KS == 11 ? 2 : KS == 7 ? 1 : 0) + (dil == 1 ? 0 : dil == 3 ? 1 : 2);
}
// option "wino8_mask" (Options::wino8_mask, default 0770770771): "wino8_mask" option: the shapes that run on conv_wino8_kernel (the others stay on conv_wino's F(4,3)) --
                                // default: every k = 7 / 11 shape, and k = 3, d = 1 at C = 64 (as F(6,3) on the two-per-CU tiles: 4 of the
                                // 6 launches of that chain, forward 33.47 -> 33.16 ms; the other k = 3 shapes measured neutral)
bool wino8_wanted(int C, int KS, int dil) {
  if (!opts().wino8 || !opts().wino || C < opts().wino_min_c || !wino8_supported(C, C, KS, dil)) return false;
  return (opts().wino8_mask >> w8_shape_bit(C, KS, dil)) & 1;
}

// option "wino8_r4" (Options::wino8_r4, default 1): "wino8_r4" option (read at dissc_gen_create): 1 = the layers "wino8_r4_mask" names run the eight points as
                           // F(5,4) (k = 7: 2 sub-filters of 4 taps, k = 11: 3) instead of F(6,3); 2 = dissc_conv1d too (tests); 0 = never
// option "wino8_r4_mask" (Options::wino8_r4_mask, default 0770770010): "wino8_r4_mask" option, same bit layout as wino8_mask (k = 3 bits ignored) -- default: every k = 7 / 11
                                   // shape of the C >= 128 stages, and k = 7, d = 1 at C = 64 (the rest of that stage is faster as F(6,3))
bool wino8_r4_supported(int C, int KS, int dil) { return wino8_supported(C, C, KS, dil) && (KS == 7 || KS == 11); }
int wino8_taps(int C, int KS, int dil) {  // taps per sub-filter the generator's policy picks for a wino8 layer
  if (!opts().wino8_r4 || !wino8_r4_supported(C, KS, dil)) return 3;
  return ((opts().wino8_r4_mask >> w8_shape_bit(C, KS, dil)) & 1) ? 4 : 3;
}

// w: [C][C][KS] -> U[p][co][ci][j] = sum_i G[p][i] w[co][ci][j + NS i], packed in A-fragment order (make_wino's, 8 points);
// R = taps per sub-filter (3: F(6,3), 4: F(5,4)), NS = ceil(KS / R)
int make_wino8(const float* w, const float* bias, int C, int KS, int dil, DevConv& dc, int R) {
  if (R != 3 && !(R == 4 && wino8_r4_supported(C, KS, dil))) {
    set_error("make_wino8: no instance with %d-tap sub-filters for C = %d, k = %d, dilation %d", R, C, KS, dil);
    return DISSC_EINVAL;
  }
  const int NS = (KS + R - 1) / R;
  if (C % 32 != 0 || C % KC != 0) {
    set_error("make_wino8: C = %d is not a multiple of the row tile", C);
    return DISSC_EINVAL;
  }
  const int nchunk = C / KC, nsub = C / 32;
  std::vector<float> packed((size_t)8 * nsub * nchunk * 2 * NS * 64 * 4);
  size_t o = 0;
  for (int p = 0; p < 8; ++p)
    for (int ms = 0; ms < nsub; ++ms)
      for (int c = 0; c < nchunk; ++c)
        for (int hf = 0; hf < 2; ++hf)
          for (int j = 0; j < NS; ++j)
            for (int lane = 0; lane < 64; ++lane)
              for (int e = 0; e < 4; ++e) {
                const int co = ms * 32 + (lane & 31), ci = c * KC + 8 * hf + 2 * e + (lane >> 5);
                double u = 0.0;
                for (int i = 0; i < R; ++i) {
                  const int tap = j + NS * i;
                  if (tap < KS) u += w8_g(p, i, R) * (double)w[((size_t)co * C + ci) * KS + tap];
                }
                packed[o++] = (float)u;
              }
  std::vector<float> b(C, 0.f);
  if (bias) memcpy(b.data(), bias, C * sizeof(float));
  dc.CIN = C; dc.M = C; dc.KS = KS; dc.dil = dil; dc.nchunk = nchunk; dc.up = 1;
  dc.groups = 1; dc.Mpad = C; dc.stride = 1; dc.pad_left = -1; dc.m32 = 1; dc.prec = 0;
  dc.macs_per_t = (double)C * C * KS;  // algorithmic (direct-form) MACs
  dc.wino = 2;
  dc.wr = R;
  int rc = upload(packed, &dc.wpack);
  if (rc) return rc;
  return upload(b, &dc.bias);
}

// MACs the matrix pipe executes per output position: 8 NS / (9 - R) per input/output channel pair
double wino8_executed_macs_per_t(int C, int KS, int R) { return (double)C * C * 8.0 * ((KS + R - 1) / R) / (9.0 - R); }

template <int NS, int DIL, int MI, int NI, int WPS = 2, int R = 3>
static int launch_wino8_t(const Wino8Args& a, int B, int Lmax, hipStream_t stream) {
  using G = Wino8Geo<NS, DIL, MI, NI, WPS, R>;
  static DeviceOnce attr_once;  // per device (common.h)
  DISSC_HIP_CHECK(attr_once.max_lds(reinterpret_cast<const void*>(&conv_wino8_kernel<NS, DIL, MI, NI, WPS, R>), 160 * 1024));
  Wino8Args aa = a;
  aa.gx = (Lmax + G::OT - 1) / G::OT;
  aa.gy = a.C / (32 * MI);
  aa.B = B;
  if (aa.gy < 1 || 8 % aa.gy != 0) {
    set_error("launch_wino8: %d row tiles do not divide the 8 XCDs", aa.gy);
    return DISSC_EINVAL;
  }
  const int per = 8 / aa.gy;
  dim3 grid(8 * ((aa.gx * B + per - 1) / per));
  if (a.C % G::CPR != 0) {
    set_error("launch_wino8: %d channels are not a multiple of the %d a round stages", a.C, G::CPR);
    return DISSC_EINVAL;
  }
  hipLaunchKernelGGL((conv_wino8_kernel<NS, DIL, MI, NI, WPS, R>), grid, dim3(512), (size_t)G::LDS_FLOATS * sizeof(float), stream, aa);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

int run_wino8(const DevConv& dc, const float* x, float* out, const float* res, float* acc, const int32_t* lengths,
              int len_default, int len_mul, int B, int ldx, int ldo, int Lmax, float slope, int epi, float mrf_div,
              hipStream_t stream) {
  Wino8Args a;
  a.x = x; a.wpack = dc.wpack; a.bias = dc.bias; a.res = res; a.out = out; a.acc = acc;
  a.lengths = lengths; a.len_default = len_default; a.len_mul = len_mul;
  a.C = dc.M; a.nchunk = dc.nchunk; a.pad = (dc.KS - 1) * dc.dil / 2;
  a.ldx = ldx; a.ldo = ldo;
  a.x_bstride = (long long)dc.M * ldx; a.o_bstride = (long long)dc.M * ldo;
  a.slope = slope; a.mrf_div = mrf_div; a.epi = epi; a.dbg = opts().kernel_dbg;
  a.gx = a.gy = a.B = 0;
  auto misaligned = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) != 0; };
  if (ldx < 4 || ldx % 4 || ldo % 4 || misaligned(x) || misaligned(out) || misaligned(res) || misaligned(acc)) {
    set_error("run_wino8: rows must be 16-byte aligned (ldx %d, ldo %d: multiples of 4 floats, ldx >= 4)", ldx, ldo);
    return DISSC_EINVAL;
  }
  if (B <= 0 || Lmax <= 0 || !wino8_supported(dc.M, dc.M, dc.KS, dc.dil)) {
    set_error("run_wino8: unsupported call (B %d, Lmax %d, C %d, k %d, d %d)", B, Lmax, dc.M, dc.KS, dc.dil);
    return DISSC_EINVAL;
  }
  const int R = dc.wr == 4 ? 4 : 3;
  const int ns = (dc.KS + R - 1) / R;
  // C = 64 (measured per shape, tools/wino8_c64.py): k = 7 as F(6,3) on 64 x 128 tiles, everything else -- k = 11, and k = 7 as
  // F(5,4) -- on 64 x 64 tiles built for two workgroups per CU (4 = the round's earlier policy: k = 7 wide in both forms)
  const int c64_mode = opts().wino8_c64_wide == 3 ? ((dc.KS == 7 && R == 3) ? 1 : 2) : opts().wino8_c64_wide == 4 ? (dc.KS == 7 ? 1 : 2) : opts().wino8_c64_wide;
  // wave tile = workgroup tile: 128 rows x 64 columns for C >= 128, 64 x 128 ("wino8_c64_wide", default) or 64 x 64 for C = 64
  // Small grids (a short or single utterance: the reference's one-at-a-time mode): the 128-row tiles of C >= 128 give a 10 s utterance
  // 16-32 workgroups on 256 CUs, each walking the whole K loop (B = 1: 240 us per k = 11 launch of the first stage); below one
  // workgroup per two CUs the launch steps down to 64-row tiles on two workgroups per CU.  Same column tiles (NI), so the same unit
  // grid and the same MFMA and transform sequence per output: bit-identical, an utterance's samples stay independent of its batch.
#define DISSC_W8_NWG(R_, NS_, D_) \
  ((long long)((Lmax + Wino8Geo<NS_, D_, 4, 2, 2, R_>::OT - 1) / Wino8Geo<NS_, D_, 4, 2, 2, R_>::OT) * (dc.M / 128) * B)
  // ... below one workgroup per four CUs to 64 rows x 32 columns (NI = 1), below one per eight to 32 x 32: every tile still starts
  // on a multiple of the unit width MO D, so the global unit grid -- and with it every output's arithmetic -- is unchanged.
  // B = 1 x 10 s (tools/batch_scaling.py): 3.78 ms (128-row tiles only) -> 3.35 -> 2.60 (NI = 1) -> see NOTES round 6.
#define DISSC_W8_NWG64(R_, NS_, D_) \
  ((long long)((Lmax + Wino8Geo<NS_, D_, 2, 2, 4, R_>::OT - 1) / Wino8Geo<NS_, D_, 2, 2, 4, R_>::OT) * (dc.M / 64) * B)
#define DISSC_W8(R_, NS_, D_)                                                                        \
  if (R == R_ && ns == NS_ && dc.dil == D_)                                                          \
    return dc.M >= 128 ? ((opts().small_grid && DISSC_W8_NWG(R_, NS_, D_) < 32)                      \
                              ? launch_wino8_t<NS_, D_, 1, 1, 4, R_>(a, B, Lmax, stream)             \
                          : (opts().small_grid && DISSC_W8_NWG(R_, NS_, D_) < 64)                    \
                              ? launch_wino8_t<NS_, D_, 2, 1, 4, R_>(a, B, Lmax, stream)             \
                          : (opts().small_grid && DISSC_W8_NWG(R_, NS_, D_) < 128)                   \
                              ? launch_wino8_t<NS_, D_, 2, 2, 4, R_>(a, B, Lmax, stream)             \
                              : launch_wino8_t<NS_, D_, 4, 2, 2, R_>(a, B, Lmax, stream))            \
                       : (opts().small_grid && DISSC_W8_NWG64(R_, NS_, D_) < 192)                     \
                          ? launch_wino8_t<NS_, D_, 2, 1, 4, R_>(a, B, Lmax, stream)                 \
                       : (c64_mode == 1 ? launch_wino8_t<NS_, D_, 2, 4, 2, R_>(a, B, Lmax, stream)  \
                          : c64_mode == 2 ? launch_wino8_t<NS_, D_, 2, 2, 4, R_>(a, B, Lmax, stream) \
                                          : launch_wino8_t<NS_, D_, 2, 2, 2, R_>(a, B, Lmax, stream));
#if DISSC_EXPERIMENTAL  // k = 3 through this kernel: only the C = 64, d = 1 shape on two-per-CU tiles is a default (wino8_mask); the rest
  DISSC_W8(3, 1, 1) DISSC_W8(3, 1, 3) DISSC_W8(3, 1, 5)  // measured neutral and is only in DISSC_EXPERIMENTAL=1 builds
#else
  // (the default build carries ONE k = 3 instance -- C = 64, d = 1 on the two-per-CU tiles: "wino8_c64_wide" does not apply to it,
  //  a handle created under another tile mode runs this instance instead of failing every forward: ADVICE r05)
  if (R == 3 && ns == 1 && dc.dil == 1 && dc.M == 64) return launch_wino8_t<1, 1, 2, 2, 4, 3>(a, B, Lmax, stream);  // (no small-grid tier: 36 us at B = 1)
  if (R == 3 && ns == 1) {
    set_error("run_wino8: k = 3 with C = %d, dilation %d, tile mode %d is only in DISSC_EXPERIMENTAL=1 builds", dc.M, dc.dil, c64_mode);
    return DISSC_EINVAL;
  }
#endif
  DISSC_W8(3, 3, 1) DISSC_W8(3, 3, 3) DISSC_W8(3, 3, 5)
  DISSC_W8(3, 4, 1) DISSC_W8(3, 4, 3) DISSC_W8(3, 4, 5)
  DISSC_W8(4, 2, 1) DISSC_W8(4, 2, 3) DISSC_W8(4, 2, 5) DISSC_W8(4, 3, 1) DISSC_W8(4, 3, 3) DISSC_W8(4, 3, 5)
#undef DISSC_W8
#undef DISSC_W8_NWG
#undef DISSC_W8_NWG64
  set_error("run_wino8: k = %d, dilation %d unsupported", dc.KS, dc.dil);
  return DISSC_EINVAL;
}

}  // namespace dissc
