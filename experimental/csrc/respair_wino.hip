// respair_wino_kernel: one residual pair of a ResBlock1,
//
//     y = x + conv_1(lrelu(conv_d(lrelu(x))))            (k taps each; reference sr/models.py:34-41)
//
// in ONE launch with BOTH convolutions in the Toom-Cook F(4,3) transform domain of conv_wino.hip (6 ceil(k / 3) MFMA
// products per 4 outputs instead of 4 k), the intermediate t = lrelu(conv_d + b1) never leaving LDS.  It replaces
//   * respair32_kernel (direct form, C = 32) for k = 7 / 11: 18 vs 28 and 24 vs 44 products, and
//   * the two conv_wino_kernel launches of a C = 64, k = 3 pair (five tensor passes -> two).
//
// What makes it different from conv_wino_kernel besides the fusion: with C = 32 (or C = 64, k = 3) a point's whole
// transform-domain weight set U_p [C][C][NS] is 64 floats per lane -- every wave keeps the weights of ITS point in
// registers for the whole tile, so the tap loops hold nothing but LDS fragment reads and MFMAs (conv_wino's loop streams
// its A fragments from L2 and saturates at 0.79 of the MFMA rate because of it, NOTES.md round 3 #6).
//
// One workgroup = 12 waves = 6 points x 2 column halves (three waves on every SIMD; one workgroup per CU), wave tile =
// C rows x NCW transform-domain columns (C = 32: 32 x 64, C = 64: 64 x 32).  Phases of a tile of OT outputs:
//   A  conv_d: lrelu(x) windows of 16 channels at a time -> LDS in conv_d's polyphase layout (conv_wino.hip); each wave
//      forms V_p of 8 channels for its point and columns, multiplies.
//   B  the 6 waves of a column half exchange Y_p through LDS; one thread applies A^T + bias + lrelu + the utterance mask to
//      4 consecutive t values and writes them INTO LDS in conv_1's polyphase layout (region T, all C channels).
//   C  conv_1 on T (no staging, no barriers).
//   D  exchange, A^T, bias, residual (raw x, L2-hot), the epilogue modes of the other pair kernels.
// Every output element sees the arithmetic of two conv_wino_kernel launches (same transforms, chunk / sub-chunk / tap /
// k-step MFMA order, same epilogue expressions), so for shapes conv_wino supports the result is bit-identical to the
// two-launch transform-domain path (tests/test_gpu_generator.py).
#include <string.h>

#include <type_traits>

#include "common.h"
#include "conv_epilogue32.h"
#include "wino_common.h"

namespace dissc {

// option "pair_wino" (Options::pair_wino, default 0): "pair_wino" option (read at dissc_gen_create): 0 (default) = not used (respair32 direct / two conv_wino
                      // launches); 1 = the shapes pairw_wanted() names run as fused transform-domain pairs, 2 = every shape
                      // with an instance.  Off by default: the gate experiment failed (below) -- per launch the winning
                      // shapes are 6-15 % faster, in the whole forward (three chains overlapping on their streams) that
                      // is 35.36 against 35.42 ms, while the executed-FLOP utilisation falls from 0.619 to 0.602.

struct PairWArgs {
  const float* x;     // [B][C][ld] pair input x_k
  float* out;         // EPI_RES: x_k' (may not alias x: neighbouring workgroups still read x's halo)
  float* acc;         // EPI_MRF_*: the stage accumulator
  const float* w1;    // register-order transform-domain weights (make_pairw): [6][MI][NS][C/8][64 lanes][4]
  const float* w2;
  const float* b1;    // [C]
  const float* b2;
  const int32_t* lengths;
  int len_default, len_mul;
  int ld;
  long long bstride;
  float slope, mrf_div;
  int epi;
  int gx, B;
  int dbg;  // diagnostics ("kernel_dbg" option): knock-outs, bit 0 transforms, 1 MFMAs, 2 the A^T passes of the exchanges
};

constexpr int pw_rup4(int n) { return (n + 3) / 4 * 4; }

// polyphase row length for NTU tile units of step D over XRW staged positions: indices 0 .. 4 NTU + 7 are touched (the
// last aligned 16-byte read of a lane starts at 4 (NTU - 1) + 8); bumped (<= 16 floats) so that the 16 lanes of a quarter
// wave -- lane = tau * D + phi reads at phi * RL + 4 tau -- start in different bank groups where that is possible
constexpr int pw_row_len(int D, int NTU, int XRW) {
  int need = 4 * NTU + 8;
  const int span = (XRW - 1 + 4 * D) / D + 1;
  if (span > need) need = span;
  need = pw_rup4(need);
  if (D == 1) return need;
  for (int rl = need; rl <= need + 16; rl += 4) {
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

// geometry of one (C, KS, DIL) instance, shared by the kernel and the host
// CHV: column halves per workgroup -- 2: 12 waves, one workgroup per CU; 1: 6 waves and half the tile, TWO workgroups per
// CU whose phases overlap (the staging / exchange / transform phases of one cover MFMA phases of the other)
template <int C, int KS, int DIL, int CHV>
struct PairWGeo {
  static constexpr int NS = (KS + 2) / 3;
  static constexpr int MI = C / 32;
  static constexpr int NI = 64 / C;            // C = 32: 2, C = 64: 1
  static constexpr int NCW = 32 * NI;          // transform-domain columns per wave
  static constexpr int P2 = (KS - 1) / 2, P1 = P2 * DIL;
  static constexpr int D1 = DIL * NS, W1 = DIL * (2 * NS - 1), D2 = NS, W2 = 2 * NS - 1;
  static constexpr int NTU1 = NCW / D1, NCOL1 = NTU1 * D1, NTU2 = NCW / D2, NCOL2 = NTU2 * D2;
  static constexpr int T1 = 4 * D1 * NTU1 * CHV;         // t values conv_d produces
  static constexpr int O2 = 4 * D2 * NTU2 * CHV;         // outputs conv_1 produces
  static constexpr int OTR = (T1 - 2 * P2) < O2 ? (T1 - 2 * P2) : O2;
  static constexpr int OT = OTR & ~3;                    // outputs a workgroup stores
  static constexpr int RAW1 = T1 + DIL * (3 * NS - 1);   // x samples conv_d's transforms touch
  static constexpr int XRW1 = pw_rup4(RAW1 + 3 + 3);     // staged positions per channel (alignment shift <= 3)
  static constexpr int NV1 = XRW1 / 4;
  static constexpr int RL1 = pw_row_len(D1, CHV * NTU1, XRW1), CHF1 = D1 * RL1;
  static constexpr int XT2 = O2 + D2 * (3 * NS - 1);     // t positions conv_1's transforms touch (beyond T1: zeros)
  static constexpr int RL2 = pw_row_len(D2, CHV * NTU2, pw_rup4(XT2 + 3)), CHF2 = D2 * RL2;
  static constexpr int XV = (NTU1 * W1 > NTU2 * W2 ? NTU1 * W1 : NTU2 * W2) <= 48 ? 48 : 112;  // V row stride (% 32 == 16)
  static constexpr int CPR = CHV == 2 ? 16 : 8;           // channels of the x window per round
  static constexpr int RP = CHV == 2 ? 1 : 2;             // row parts of a Y exchange pass (LDS budget)
  static constexpr int NWV = 6 * CHV, NTH = 64 * NWV;
  static constexpr int YS = NCW + 4;
  static constexpr int T_FLOATS = C * CHF2;
  static constexpr int XW_FLOATS = CPR * CHF1;
  static constexpr int V_FLOATS = NWV * 8 * XV;
  static constexpr int Y_FLOATS = 6 * (C / RP) * YS;      // one exchange pass: a column half, C / RP rows
  static constexpr int S_FLOATS = (XW_FLOATS + V_FLOATS) > Y_FLOATS ? (XW_FLOATS + V_FLOATS) : Y_FLOATS;
  static constexpr int LDS_FLOATS = T_FLOATS + S_FLOATS;
  static constexpr int NW4 = MI * NS * (C / 8);           // float4 weight registers per lane and conv
  static_assert(C == 32 || C == 64, "C");
  static_assert(NTU1 >= 1 && NTU2 >= 1 && NTU1 * W1 <= XV && NTU2 * W2 <= XV, "tile geometry");
  static_assert(OT > 0 && OT + 2 * P2 <= T1 && OT <= O2, "output tile");
  static_assert(LDS_FLOATS * 4 <= (CHV == 2 ? 160 : 80) * 1024, "LDS");
  static_assert(NW4 <= 16, "a point's weights must fit 64 registers");
};

template <int C, int KS, int DIL, int CHV>
__global__ void __launch_bounds__(384 * CHV, 3) respair_wino_kernel(const PairWArgs a) {
  using G = PairWGeo<C, KS, DIL, CHV>;
  constexpr int NTH = G::NTH, RP = G::RP;
  constexpr int NS = G::NS, MI = G::MI, NI = G::NI, NCW = G::NCW, XV = G::XV, YS = G::YS, CPR = G::CPR;
  constexpr int CL = 64 / NCW;                 // lanes per column in the transform (1 or 2: the halves take alternate channels)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const Treg = lds;                     // [C][D2][RL2]: t in conv_1's polyphase layout
  float* const scr = lds + G::T_FLOATS;        // x window + V tiles | Y exchange
  float* const xwin = scr;                     // [CPR][D1][RL1]
  float* const vbuf = scr + G::XW_FLOATS;      // [waves][8][XV]

  // ---- which tile: only the tiles that exist are enumerated (conv_wino.hip: utterance 0's ceil(len_0 / OT) tiles, then
  // utterance 1's, ...), found by a prefix sum of the tile counts over the lanes; the empty workgroups sit at the end.
  // (A persistent form -- workgroups looping over tiles, the next tile's weights and window fetched behind the last
  // phase -- measured 10-15 % SLOWER: the loop keeps enough per-lane state alive to spill 50-80 registers.)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  auto find_tile = [&](int lin, int& tb_, int& tlen, int& to0) __attribute__((always_inline)) -> bool {
    if (a.lengths == nullptr) {
      tb_ = lin / a.gx;
      tlen = a.len_default;
      to0 = (lin - tb_ * a.gx) * G::OT;
      return tb_ < a.B && to0 < tlen;
    }
    int base = 0;
    for (int b0 = 0; b0 < a.B; b0 += 64) {
      const int l = b0 + lane < a.B ? a.lengths[b0 + lane] * a.len_mul : 0;
      const int nt = (l + G::OT - 1) / G::OT;
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
        tb_ = __builtin_amdgcn_readfirstlane(b0 + lb);
        tlen = __builtin_amdgcn_readfirstlane(__shfl(l, lb, 64));
        to0 = __builtin_amdgcn_readfirstlane((lin - base - __shfl(incl - nt, lb, 64)) * G::OT);
        return true;
      }
      base += total;
    }
    return false;
  };
  const int lin = blockIdx.x;
  int b, len, o0;
  if (!find_tile(lin, b, len, o0)) return;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = wave % 6;       // this wave's evaluation point
  const int chh = CHV == 2 ? wave / 6 : 0;     // ... and its column half
  const int l31 = lane & 31, h = lane >> 5;
  const float slope = a.slope;
  const float* xb = a.x + (size_t)b * a.bstride;
  const int o2 = o0 - G::P2;            // position of t's first value (= conv_1's window origin)
  const int o1 = o2 - G::P1;            // conv_d's window origin
  const int tb = o1 & ~3, sh = o1 - tb;

  // ---- this wave's weights: NW4 float4 per lane, [p][mi][j][ksub][lane]; conv_d's first, conv_1's take the registers over
  f32x4 wr[G::NW4];
  auto load_w = [&](const float* w) __attribute__((always_inline)) {
    const f32x4* wp = reinterpret_cast<const f32x4*>(w) + (size_t)p * G::NW4 * 64 + lane;
#pragma unroll
    for (int i = 0; i < G::NW4; ++i) wr[i] = wp[i * 64];
  };
  load_w(a.w1);

  // ---- x window staging (conv_wino.hip): clamped 16-byte loads, activation + zero padding + polyphase scatter to LDS
  constexpr int NV = G::NV1, SV = (CPR * NV + NTH - 1) / NTH;
  const int r0 = tid / NV, v0 = tid - r0 * NV;
  constexpr int dr = NTH / NV, dv = NTH - dr * NV;
  f32x4 sv[SV];
  auto stage_load = [&](int rd, const float* xb_, int tb_) __attribute__((always_inline)) {
    int r = r0, v = v0;
#pragma unroll
    for (int i = 0; i < SV; ++i) {
      const int ci = rd * CPR + (r < CPR ? r : CPR - 1);
      int t = tb_ + 4 * v;
      t = t < 0 ? 0 : (t > a.ld - 4 ? a.ld - 4 : t);
      sv[i] = *reinterpret_cast<const f32x4*>(xb_ + (size_t)ci * a.ld + t);
      v += dv;
      r += dr;
      if (v >= NV) { v -= NV; ++r; }
    }
  };
  auto stage_store = [&]() __attribute__((always_inline)) {
    int r = r0, v = v0;
#pragma unroll
    for (int i = 0; i < SV; ++i) {
      if (r < CPR) {
        const int t = tb + 4 * v;
        const f32x4 val = sv[i];
        float* rowp = xwin + r * G::CHF1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = (t + e) >= 0 && (t + e) < len;
          const int xs = 4 * v + e - sh + 4 * G::D1;  // >= 4 D - 3 > 0
          rowp[(xs % G::D1) * G::RL1 + xs / G::D1] = ok ? lrelu(val[e], slope) : 0.f;
        }
      }
      v += dv;
      r += dr;
      if (v >= NV) { v -= NV; ++r; }
    }
  };
  stage_load(0, xb, tb);
  // region T: zero (conv_1's transforms read a few positions beyond what conv_d produces; every sample a stored
  // output's F(4,3) group touches must be finite -- and zero costs no accuracy)
  for (int i = tid; i < G::T_FLOATS / 4; i += NTH) reinterpret_cast<f32x4*>(Treg)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x16 acc[MI][NI];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
  };
  float* const vp = vbuf + wave * (8 * XV);

  // transform of 8 channels for this wave's point and columns: lane <-> column (tau, phi), entries w = phi and phi + D
  auto transform8 = [&](const float* win8, auto dc, auto wc, auto rlc, auto ntuc) __attribute__((always_inline)) {
    constexpr int D = decltype(dc)::value, W = decltype(wc)::value, RL = decltype(rlc)::value, NTU = decltype(ntuc)::value;
    constexpr int NCOL = NTU * D, CHF = D * RL;
    const int cl = lane % NCW, csel = lane / NCW;
    const int tcol = cl < NCOL ? cl : NCOL - 1;
    const int ttau = tcol / D, tphi = tcol % D;
    const int toff = tphi * RL + 4 * (chh * NTU + ttau) + 4;
    const int e0 = ttau * W + tphi;
    const bool ok1 = tphi + D < W;
    const int e1 = ok1 ? e0 + D : e0;
    const float* rw = win8 + toff;
    if (a.dbg & 1) return;
    auto go = [&](auto pc) __attribute__((always_inline)) {
      constexpr int P = decltype(pc)::value;
#pragma unroll
      for (int it = 0; it < 8 / (2 * CL); ++it) {
        f32x4 lo[2], hi[2];
        int chn[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          chn[i] = (it * 2 + i) * CL + csel;
          lo[i] = *reinterpret_cast<const f32x4*>(rw + chn[i] * CHF);
          hi[i] = *reinterpret_cast<const f32x4*>(rw + chn[i] * CHF + 4);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float v0 = wino_bt<P>(lo[i][0], lo[i][1], lo[i][2], lo[i][3], hi[i][0], hi[i][1]);
          vp[chn[i] * XV + e0] = v0;
          if constexpr (W > D) {
            const float v1 = wino_bt<P>(lo[i][1], lo[i][2], lo[i][3], hi[i][0], hi[i][1], hi[i][2]);
            vp[chn[i] * XV + e1] = ok1 ? v1 : v0;
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
      default: go(std::integral_constant<int, 5>{}); break;
    }
  };

  // NS taps x 4 k-steps on this wave's V tile with the register weights of sub-chunk `ksub` (compile-time)
  auto run_taps = [&](auto ksc, auto dc, auto wc, auto ntuc, auto dilc) __attribute__((always_inline)) {
    constexpr int KSUB = decltype(ksc)::value, D = decltype(dc)::value, W = decltype(wc)::value, NTU = decltype(ntuc)::value;
    constexpr int TD = decltype(dilc)::value, NCOL = NTU * D;
    if (a.dbg & 2) return;
    const float* bj[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      int col = ni * 32 + l31;
      col = col < NCOL ? col : NCOL - 1;
      bj[ni] = vp + (col / D) * W + (col % D) + h * XV;
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      float bk[4][NI];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bk[s][ni] = bj[ni][s * 2 * XV + j * TD];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[(mi * NS + j) * (C / 8) + KSUB][s], bk[s][ni], acc[mi][ni], 0, 0, 0);
    }
  };

  using ID1 = std::integral_constant<int, G::D1>;
  using IW1 = std::integral_constant<int, G::W1>;
  using IR1 = std::integral_constant<int, G::RL1>;
  using IN1 = std::integral_constant<int, G::NTU1>;
  using ID2 = std::integral_constant<int, G::D2>;
  using IW2 = std::integral_constant<int, G::W2>;
  using IR2 = std::integral_constant<int, G::RL2>;
  using IN2 = std::integral_constant<int, G::NTU2>;

  stage_store();
  if (C / CPR > 1) stage_load(1, xb, tb);
  __syncthreads();
  // ================================ phase A: conv_d ================================
  zero_acc();
  auto round_a = [&](auto rdc) __attribute__((always_inline)) {
    constexpr int RD = decltype(rdc)::value, SPR = CPR / 8;  // 8-channel sub-chunks per round
    if constexpr (RD < C / CPR) {
      transform8(xwin, ID1{}, IW1{}, IR1{}, IN1{});
      run_taps(std::integral_constant<int, SPR * RD>{}, ID1{}, IW1{}, IN1{}, std::integral_constant<int, DIL>{});
      if constexpr (SPR > 1) {
        transform8(xwin + 8 * G::CHF1, ID1{}, IW1{}, IR1{}, IN1{});
        run_taps(std::integral_constant<int, SPR * RD + 1>{}, ID1{}, IW1{}, IN1{}, std::integral_constant<int, DIL>{});
      }
      if constexpr (RD + 1 < C / CPR) {
        __syncthreads();   // every wave is done with this round's window
        stage_store();     // round RD + 1 (its loads were issued a round ago)
        if constexpr (RD + 2 < C / CPR) stage_load(RD + 2, xb, tb);
        __syncthreads();
      }
    }
  };
  round_a(std::integral_constant<int, 0>{});
  round_a(std::integral_constant<int, 1>{});
  round_a(std::integral_constant<int, 2>{});
  round_a(std::integral_constant<int, 3>{});
  round_a(std::integral_constant<int, 4>{});
  round_a(std::integral_constant<int, 5>{});
  round_a(std::integral_constant<int, 6>{});
  round_a(std::integral_constant<int, 7>{});

  // ---- A^T of one row's quad: 4 consecutive outputs u = 4 g + e of unit tau from the Y_p [6][C][YS] in LDS
  auto at_quad = [&](const float* yb, int row, int tau, int g, auto dc) __attribute__((always_inline)) -> f32x4 {
    constexpr int D = decltype(dc)::value;
    const float* yr = yb + row * YS + tau * D;
    constexpr int PS = (C / RP) * YS;  // point stride
    f32x4 v;
    if constexpr (D == 1) {
      const float y0 = yr[0], y1 = yr[PS], y2 = yr[2 * PS], y3 = yr[3 * PS], y4 = yr[4 * PS], y5 = yr[5 * PS];
      const float s12 = y1 + y2, d12 = y1 - y2;
      v[0] = (y0 + s12) + (y3 + y4);
      v[1] = d12 + fmaf(2.f, y3, -0.5f * y4);
      v[2] = s12 + fmaf(4.f, y3, 0.25f * y4);
      v[3] = d12 + fmaf(8.f, y3, -0.125f * y4) + y5;
    } else if constexpr (D % 4 == 0) {
      const int ao = (4 * g) / D, rho = 4 * g - ao * D;
      const float* yc = yr + rho;
      const f32x4 y0 = *reinterpret_cast<const f32x4*>(yc), y1 = *reinterpret_cast<const f32x4*>(yc + PS),
                  y2 = *reinterpret_cast<const f32x4*>(yc + 2 * PS), y3 = *reinterpret_cast<const f32x4*>(yc + 3 * PS),
                  y4 = *reinterpret_cast<const f32x4*>(yc + 4 * PS), y5 = *reinterpret_cast<const f32x4*>(yc + 5 * PS);
      const float sg = (ao & 1) ? -1.f : 1.f;
      const float p2 = __int_as_float((127 + ao) << 23), c4 = sg * __int_as_float((127 - ao) << 23);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = (ao == 0 ? y0[e] : 0.f) + y1[e];
        x = fmaf(sg, y2[e], x);
        x = fmaf(p2, y3[e], x);
        x = fmaf(c4, y4[e], x);
        x += (ao == 3 ? y5[e] : 0.f);
        v[e] = x;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int u = 4 * g + e;
        const int ao = u / D, rho = u - ao * D;
        const float* yc = yr + rho;
        const float y0 = yc[0], y1 = yc[PS], y2 = yc[2 * PS], y3 = yc[3 * PS], y4 = yc[4 * PS], y5 = yc[5 * PS];
        const float sg = (ao & 1) ? -1.f : 1.f;
        const float p2 = __int_as_float((127 + ao) << 23), ip2 = __int_as_float((127 - ao) << 23);
        float x = (ao == 0 ? y0 : 0.f) + y1;
        x = fmaf(sg, y2, x);
        x = fmaf(p2, y3, x);
        x = fmaf(sg * ip2, y4, x);
        x += (ao == 3 ? y5 : 0.f);
        v[e] = x;
      }
    }
    return v;
  };
  auto y_write = [&](float* yb, auto rpc) __attribute__((always_inline)) {  // this wave's accumulators of row part RPI -> Y_p[row][col]
    constexpr int RPI = decltype(rpc)::value, RPP = C / RP;  // rows per part
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          constexpr int dummy = 0;
          (void)dummy;
          const int rbase = mi * 32 + 8 * (r >> 2);          // compile-time after unrolling; (r & 3) + 4 h stays below 8
          if (rbase / RPP == RPI)
            yb[(p * RPP + (rbase % RPP) + (r & 3) + 4 * h) * YS + ni * 32 + l31] = acc[mi][ni][r];
        }
  };

  // ================================ phase B: Y exchange -> t in LDS ================================
  __syncthreads();  // all MFMAs of conv_d are done: the scratch region becomes the exchange buffer
  load_w(a.w2);  // conv_1's weights take over the weight registers (fetched behind the exchange)
  float* const yb = scr;
  auto pass_b = [&](auto hc, auto rpc) __attribute__((always_inline)) {
    constexpr int half = decltype(hc)::value, RPI = decltype(rpc)::value, RPP = C / RP;
    if constexpr (half < CHV && RPI < RP) {
      if (half > 0 || RPI > 0) __syncthreads();
      if (chh == half) y_write(yb, rpc);
      __syncthreads();
      constexpr int NQ = RPP * G::NCOL1, NIT = (NQ + NTH - 1) / NTH;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * NTH;
        if (idx >= NQ) continue;
        if (a.dbg & 4) continue;
        const int lrow = idx / G::NCOL1, qi = idx - lrow * G::NCOL1;
        const int row = RPI * RPP + lrow;
        const int tau = qi / G::D1, g = qi - tau * G::D1;
        f32x4 v = at_quad(yb, lrow, tau, g, ID1{});
        const float bz = a.b1[row];
        const int x2 = 4 * G::D1 * (half * G::NTU1 + tau) + 4 * g;  // position within t (multiple of 4)
        float* trow = Treg + row * G::CHF2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int nt = o2 + x2 + e;
          v[e] = (nt >= 0 && nt < len) ? lrelu(v[e] + bz, slope) : 0.f;
        }
        if constexpr (G::D2 == 1) {
          *reinterpret_cast<f32x4*>(trow + x2 + 4) = v;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int xs = x2 + e + 4 * G::D2;
            trow[(xs % G::D2) * G::RL2 + xs / G::D2] = v[e];
          }
        }
      }
    }
  };
  pass_b(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  pass_b(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
  pass_b(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
  pass_b(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
  __syncthreads();  // t is complete; the exchange buffer is free (the V tiles live in it)

  // ================================ phase C: conv_1 on T ================================
  zero_acc();
  auto sub_c = [&](auto kc) __attribute__((always_inline)) {
    constexpr int KSUB = decltype(kc)::value;
    transform8(Treg + (KSUB * 8) * G::CHF2, ID2{}, IW2{}, IR2{}, IN2{});
    run_taps(kc, ID2{}, IW2{}, IN2{}, std::integral_constant<int, 1>{});
  };
  sub_c(std::integral_constant<int, 0>{});
  sub_c(std::integral_constant<int, 1>{});
  sub_c(std::integral_constant<int, 2>{});
  sub_c(std::integral_constant<int, 3>{});
  if constexpr (C / 8 > 4) {
    sub_c(std::integral_constant<int, 4>{});
    sub_c(std::integral_constant<int, 5>{});
    sub_c(std::integral_constant<int, 6>{});
    sub_c(std::integral_constant<int, 7>{});
  }

  // ================================ phase D: Y exchange -> y = x + conv + b2 (or the MRF modes) ================================
  const int epi = a.epi;
  const size_t ob = (size_t)b * a.bstride;
  __syncthreads();
  auto pass_d = [&](auto hc, auto rpc) __attribute__((always_inline)) {
    constexpr int half = decltype(hc)::value, RPI = decltype(rpc)::value, RPP = C / RP;
    if constexpr (half < CHV && RPI < RP) {
      constexpr int NQ = RPP * G::NCOL2, NIT = (NQ + NTH - 1) / NTH;
      // residual quads of the pass, fetched before the exchange (raw x: this workgroup staged it a moment ago)
      f32x4 pres[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * NTH;
        const int lrow = idx / G::NCOL2, qi = idx - lrow * G::NCOL2;
        const int row = RPI * RPP + lrow;
        const int tau = qi / G::D2, g = qi - tau * G::D2;
        const int xo = 4 * G::D2 * (half * G::NTU2 + tau) + 4 * g;
        const int n0 = o0 + xo;
        if (idx < NQ && xo < G::OT && n0 + 4 <= len) pres[it] = *reinterpret_cast<const f32x4*>(a.x + ob + (size_t)row * a.ld + n0);
      }
      if (half > 0 || RPI > 0) __syncthreads();
      if (chh == half) y_write(yb, rpc);
      __syncthreads();
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * NTH;
        if (idx >= NQ) continue;
        const int lrow = idx / G::NCOL2, qi = idx - lrow * G::NCOL2;
        const int row = RPI * RPP + lrow;
        const int tau = qi / G::D2, g = qi - tau * G::D2;
        const int xo = 4 * G::D2 * (half * G::NTU2 + tau) + 4 * g;
        const int n0 = o0 + xo;
        if (xo >= G::OT || n0 >= len || (a.dbg & 4)) continue;   // (OT is a multiple of 4: a quad is stored whole or not at all)
        f32x4 v = at_quad(yb, lrow, tau, g, ID2{});
        const float bz = a.b2[row];
        v[0] += bz; v[1] += bz; v[2] += bz; v[3] += bz;
        const size_t ix = ob + (size_t)row * a.ld + n0;
        if (n0 + 4 <= len) {
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
        } else {
          for (int e = 0; e < len - n0; ++e) {
            float x = v[e] + a.x[ix + e];
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
  };
  pass_d(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  pass_d(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
  pass_d(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{});
  pass_d(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{});
}


// ---------------------------------------------------------------------------------------------
// host side of the F(4,3) pair kernel (the dispatch that is shared with the register-only F(2,3) pairs is dissc_amd/csrc/pair_host.hip)
// ---------------------------------------------------------------------------------------------
bool pairw43_built() { return true; }

int pack_pairw43(const float* w, int C, int KS, float** dev) {
  const int NS = (KS + 2) / 3, MI = C / 32, NK = C / 8;
  std::vector<float> packed((size_t)6 * MI * NS * NK * 64 * 4);
  size_t o = 0;
  for (int p = 0; p < 6; ++p)
    for (int mi = 0; mi < MI; ++mi)
      for (int j = 0; j < NS; ++j)
        for (int ks = 0; ks < NK; ++ks)
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 4; ++e) {
              const int co = 32 * mi + (lane & 31), ci = 8 * ks + 2 * e + (lane >> 5);
              double u = 0.0;
              for (int i = 0; i < 3; ++i) {
                const int tap = j + NS * i;
                if (tap < KS) u += kWinoG[p][i] * (double)w[((size_t)co * C + ci) * KS + tap];
              }
              packed[o++] = (float)u;
            }
  return upload(packed, dev);
}

template <int C, int KS, int DIL, int CHV>
static int launch_pairw_t(PairWArgs a, int B, int Lmax, hipStream_t stream) {
  using G = PairWGeo<C, KS, DIL, CHV>;
  static DeviceOnce attr_once;  // per device (common.h)
  DISSC_HIP_CHECK(attr_once.max_lds(reinterpret_cast<const void*>(&respair_wino_kernel<C, KS, DIL, CHV>), 160 * 1024));
  a.gx = (Lmax + G::OT - 1) / G::OT;
  a.B = B;
  const int grid = a.gx * B;
  hipLaunchKernelGGL((respair_wino_kernel<C, KS, DIL, CHV>), dim3(grid), dim3(G::NTH), (size_t)G::LDS_FLOATS * sizeof(float),
                     stream, a);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

int launch_pairw43(const DevPairW& pw, const float* x, float* out, float* acc, const int32_t* lengths, int len_default,
                   int len_mul, int B, int Lmax, int ld, float slope, int epi, float mrf_div, hipStream_t stream) {
  PairWArgs a;
  a.x = x; a.out = out; a.acc = acc; a.w1 = pw.w1; a.w2 = pw.w2; a.b1 = pw.b1; a.b2 = pw.b2;
  a.lengths = lengths; a.len_default = len_default; a.len_mul = len_mul; a.ld = ld;
  a.bstride = (long long)pw.C * ld; a.slope = slope; a.mrf_div = mrf_div; a.epi = epi; a.gx = 0; a.B = B; a.dbg = opts().kernel_dbg;
#define DISSC_PAIRW(C_, K_, D_)                                                                       \
  if (pw.C == C_ && pw.KS == K_ && pw.dil == D_)                                                      \
    return opts().pairw_chv == 2 ? launch_pairw_t<C_, K_, D_, 2>(a, B, Lmax, stream) : launch_pairw_t<C_, K_, D_, 1>(a, B, Lmax, stream);
  DISSC_PAIRW(32, 7, 1) DISSC_PAIRW(32, 7, 3) DISSC_PAIRW(32, 7, 5)
  DISSC_PAIRW(32, 11, 1) DISSC_PAIRW(32, 11, 3) DISSC_PAIRW(32, 11, 5)
  DISSC_PAIRW(64, 3, 1) DISSC_PAIRW(64, 3, 3) DISSC_PAIRW(64, 3, 5)
#undef DISSC_PAIRW
  set_error("launch_pairw43: no instance for C = %d, k = %d, dilation %d", pw.C, pw.KS, pw.dil);
  return DISSC_EINVAL;
}

}  // namespace dissc
