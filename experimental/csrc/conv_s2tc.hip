// conv_s2tc_kernel: HuBERT's stride-2, k = 3 feature convs (conv1..conv4 of the CNN feature extractor: 512 -> 512 channels,
// no padding, no bias, exact-erf GELU; fairseq ConvFeatureExtractionModel as data/encode.py:21-22,32 reaches it through
// textless -- third-party, architecture per HF modeling_hubert.py:154-213, restated in oracle/hubert_ref.py) in POLYPHASE
// Toom-Cook form, fp32 operands / products / accumulation on v_mfma_f32_32x32x2_f32.
//
//     y[co][t] = sum_ci  w[co][ci][0] x[ci][2t] + w[co][ci][1] x[ci][2t+1] + w[co][ci][2] x[ci][2t+2]
// With e[n] = x[2n], o[n] = x[2n+1] this is a 2-tap stride-1 filter on the even samples plus a 1-tap one on the odd samples:
//     y[t] = (w0 e[t] + w2 e[t+1]) + w1 o[t].
// The 2-tap part runs as F(7,2) on the eight points 0, +-1, +-2, +-1/2, inf (conv_wino8.hip's B^T): 8 products for 7 outputs;
// the 1-tap part is a plain GEMM, 7 products for 7 outputs, accumulated AFTER the inverse transform -- 15 MFMA products per
// 7 outputs instead of 21 (2.14 per output instead of 3, the minimum for this shape: 2m + 1 distinct inputs per m outputs).
//
// One workgroup = 64 output rows x 32 units (224 outputs) = FOUR waves, and TWO workgroups per CU (<= 80 KB of LDS, <= 256
// registers): two "point waves" hold the transform-domain accumulators of the SEVEN finite points of one 32 x 32 half of the tile
// each, two "direct waves" the SEVEN odd-sample accumulators of the same halves -- and the point at infinity, whose product enters
// the last output of a unit only (A^T's last column is a unit vector), accumulates straight into that output's direct
// accumulator.  7 + 8 = 15 MFMAs per k-step for a pair of waves and no idle slot (a one-wave-per-point layout would leave a sixteenth
// of the pipe empty), every wave holds 112 accumulator registers, and a point wave owns all finite points of its outputs: the
// inverse transform A^T happens in registers, nothing is exchanged per point.  The two workgroups of a CU run out of phase --
// what one does between its MFMAs (staging, transform, barrier, epilogue) the other covers; the first build, ONE 8-wave
// workgroup of 64 x 64 units per CU, ran at exactly MFMA time + everything else, whatever the order inside the round
// (profiles/r05/s2tc_gate.txt).
//   * window: raw rows [8 channels][452 samples] (the even / odd planes stay interleaved: unit u, sample q is float 14 u + 2 q
//     (+ 1 for the odd plane) -- lane stride 14 floats, conflict-free), three buffers: written one round, transformed the
//     next, multiplied the third; loaded into registers a round before they are written;
//   * transform IN PLACE: thread (channel = tid / 32, unit = tid % 32) reads its 8 even samples, forms all eight points with shared
//     sub-expressions (26 operations) and writes the seven finite ones back over the unit's seven even samples (a row belongs
//     to one wave, whose reads all precede its writes), the point at infinity into a small side array.  The point waves'
//     B fragments are then the EVEN plane and the direct waves' the ODD plane of the same rows: one fragment address pattern,
//     one float apart -- and one code path for both kinds of wave (two paths through 112 accumulator registers made the
//     register allocator keep two copies).  Inputs at or beyond the utterance's length are zeroed here (the transform mixes
//     a unit's inputs, so garbage behind the row's end would leak into valid outputs by rounding -- the direct form never
//     touches it);
//   * one barrier per 8 channels; A fragments (transform-domain weights U_p = G w, packed in walk order) half a round ahead
//     (two register sets of 4 k-step pairs: 32 registers instead of 64 for a whole round);
//   * epilogue: direct waves put their partial sums into an output tile in LDS, point waves add A^T Y, all 256 threads
//     apply bias / GELU and store 16 bytes per lane.
// Rounding: F(7,2) on these points is the worst-conditioned form in the library (A^T reaches 64 and 1/64); per layer the
// numpy model gives 2.7x a blocked-fp32 direct conv's rms error on GELU-shaped inputs (F(5,4): 1.8x F(4,3)); measured numbers
// in DESIGN.md section 5.
#include <string.h>

#include "common.h"
#include "conv_epilogue32.h"

namespace dissc {

// option "enc_tc" (Options::enc_tc, default 0): "enc_tc" option (read at dissc_hubert_create): 1 = conv1..conv4 of the feature extractor use this kernel.
                        // OFF by default: the round-5 gate (conv1 <= 5.3 ms AND per-layer rms <= 2x the direct form's) failed on both
                        // counts -- 5.95-6.09 ms against the direct kernel's 6.30-6.45 on the same boxes (conv1..4: 11.7 vs 12.4-12.5 ms,
                        // encode 27.48 vs 27.99 ms) and 2.35x the rms error (profiles/r05/s2tc_gate.txt)
// option "s2tc_xmode" (Options::s2tc_xmode, default 0): "s2tc_xmode" option: 0 = row tiles pinned to XCDs (weights L2-resident), 1 = row tiles of a time tile share an XCD
// option "kernel_dbg": diagnostics: knock-outs, bit 0 transform, 1 MFMAs, 2 epilogue, 3 staging loads

namespace {

constexpr int S2_MO = 7;                 // outputs per unit
constexpr int S2_NU = 32;                // units per tile
constexpr int S2_OT = S2_MO * S2_NU;     // 224 outputs per tile
constexpr int S2_CPR = 8;                // channels per round
constexpr int S2_RS = 452;               // window row stride (floats): 2 * 224 + 1 samples, rounded up to float4
constexpr int S2_NV = S2_RS / 4;         // float4 per row
constexpr int S2_NSLOT = S2_CPR * S2_NV; // float4 staging slots per round
constexpr int S2_NTH = 256;
constexpr int S2_SV = (S2_NSLOT + S2_NTH - 1) / S2_NTH;
constexpr int S2_XI = 40;                // row stride of the point-at-infinity array [8 channels][XI]
constexpr int S2_WIN = S2_CPR * S2_RS;   // floats per window buffer
constexpr int S2_INF = S2_CPR * S2_XI;   // floats per point-at-infinity buffer
constexpr int S2_OS = S2_OT + 4;         // epilogue tile row stride
constexpr int S2_LOOP_FLOATS = 3 * (S2_WIN + S2_INF);
constexpr int S2_LDS_FLOATS = S2_LOOP_FLOATS > 64 * S2_OS ? S2_LOOP_FLOATS : 64 * S2_OS;  // (the epilogue's output tile is the larger)
static_assert(S2_LDS_FLOATS * 4 <= 80 * 1024, "LDS: two workgroups per CU");

struct S2tcArgs {
  const float* x;        // [B][CIN][ldx], valid conv: nothing is padded
  const float* wpack;    // point waves [M / 32][CIN / 8][2 halves][4 slot pairs][64 lanes][4], then direct waves [..][..][2][2][64][4] (make_s2tc)
  const float* bias;     // [M] or nullptr
  float* out;            // [B][M][ldo]
  const int32_t* lengths_in;   // [B] valid input samples, or nullptr
  const int32_t* lengths_out;  // [B] valid outputs, or nullptr
  int len_default, olen_default;
  int CIN, M, act;
  int ldx, ldo;
  long long x_bstride, o_bstride;
  int gx, nrt, B, xmode, dbg;
};

}  // namespace

// DBG (diagnostics, instances of their own): knock-outs, bit 0 transform, 1 MFMAs, 2 epilogue, 3 window staging, 4 A loads, 5 B reads
template <int DBG>
__global__ void __launch_bounds__(S2_NTH, 2) conv_s2tc_kernel(const S2tcArgs a) {
  constexpr int RS = S2_RS, XI = S2_XI, OS = S2_OS, NTH = S2_NTH;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const win = lds;                 // [3][8][RS]
  float* const vinf = lds + 3 * S2_WIN;   // [3][8 channels][XI]: V of the point at infinity

  // ---- tile: 1-D grid, XCD-aware (workgroup id % 8 = XCD)
  const int xcd = blockIdx.x & 7, kq = blockIdx.x >> 3;
  int mt, lin;
  if (a.xmode == 0) {  // row tiles pinned to XCD groups: an XCD's L2 keeps its rows of the weights
    const int per = 8 / a.nrt;
    mt = xcd / per;
    lin = (xcd - mt * per) + per * kq;
  } else {             // the row tiles of one time tile run side by side on one XCD and share its window through L2
    mt = kq % a.nrt;
    lin = (kq / a.nrt) * 8 + xcd;
  }
  if (lin >= a.gx * a.B) return;
  const int b = lin / a.gx;
  const int t0 = (lin - b * a.gx) * S2_OT;
  const int len = a.lengths_in ? a.lengths_in[b] : a.len_default;
  const int olen = a.lengths_out ? a.lengths_out[b] : a.olen_default;
  if (t0 >= olen) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  // 0: point wave, 1: direct wave.  Wave i of a CU's two workgroups tends to share SIMD i: every other workgroup swaps the roles, so
  // that a SIMD hosts one wave of each kind (7 + 8 MFMAs per k-step) rather than two of a kind (14 or 16)
  const int kind = (wave >> 1) ^ ((kq >> 5) & 1);
  const int mi = wave & 1;
  const int tin0 = 2 * t0;                 // multiple of 4: 16-byte aligned window rows
  const float* xb = a.x + (size_t)b * a.x_bstride;
  const int nround = a.CIN / S2_CPR;

  // ---- window staging: raw copy, registers -> LDS a round later.  Slot e = tid + 256 i -> float4 e of the linear [8][RS] buffer.
  // Loads are clamped into the row (a clamped slot only carries positions >= ldx >= len: masked by the transform, never stored
  // by the direct part).
  int soff[S2_SV];
#pragma unroll
  for (int i = 0; i < S2_SV; ++i) {
    int e = tid + i * NTH;
    e = e < S2_NSLOT ? e : S2_NSLOT - 1;
    const int row = e / S2_NV, c4 = e - row * S2_NV;
    int t = tin0 + 4 * c4;
    t = t > a.ldx - 4 ? a.ldx - 4 : t;
    soff[i] = row * a.ldx + t;
  }
  f32x4 sv[S2_SV];
  auto stage_load = [&](int rd) __attribute__((always_inline)) {
    if constexpr (DBG & 8) return;
    const float* xr = xb + (size_t)rd * S2_CPR * a.ldx;
#pragma unroll
    for (int i = 0; i < S2_SV; ++i) sv[i] = *reinterpret_cast<const f32x4*>(xr + soff[i]);
  };
  auto stage_store = [&](float* wbuf) __attribute__((always_inline)) {
    if constexpr (DBG & 8) return;
#pragma unroll
    for (int i = 0; i < S2_SV; ++i) {
      const int e = tid + i * NTH;
      if (e < S2_NSLOT) *reinterpret_cast<f32x4*>(wbuf + 4 * e) = sv[i];
    }
  };

  // ---- transform: thread (channel, unit): even samples 14 unit + 2 q of its channel's row -> eight points
  const int tch = tid >> 5, tun = tid & 31;                     // a row's 32 units sit in one wave: its reads all precede its writes
  const int troff = tch * RS + 14 * tun;
  const int nval = len - tin0 - 14 * tun;                      // valid raw samples from this unit's first one on
  const bool tail = tin0 + 14 * (S2_NU - 1) + 15 > len;        // (uniform) this tile reaches beyond the utterance
  auto transform = [&](float* wbuf, float* idst) __attribute__((always_inline)) {
    if constexpr (DBG & 1) return;
    float* r = wbuf + troff;
    float r0 = r[0], r1 = r[2], r2 = r[4], r3 = r[6], r4 = r[8], r5 = r[10], r6 = r[12], r7 = r[14];
    if (tail) {
      r0 = 0 < nval ? r0 : 0.f; r1 = 2 < nval ? r1 : 0.f; r2 = 4 < nval ? r2 : 0.f; r3 = 6 < nval ? r3 : 0.f;
      r4 = 8 < nval ? r4 : 0.f; r5 = 10 < nval ? r5 : 0.f; r6 = 12 < nval ? r6 : 0.f; r7 = 14 < nval ? r7 : 0.f;
    }
    // rows of B^T (conv_wino8.hip's w8_bt), the +- point pairs sharing their even / odd halves
    const float p0 = fmaf(5.25f, r2 - r4, r6 - r0);
    const float p7 = fmaf(5.25f, r3 - r5, r7 - r1);
    const float ea = fmaf(-4.25f, r4, r2 + r6), oa = fmaf(-4.25f, r3, r1 + r5);
    const float eb = fmaf(0.25f, r2, fmaf(-1.25f, r4, r6)), ob = fmaf(0.5f, r1, fmaf(-2.5f, r3, 2.f * r5));
    const float ec = fmaf(4.f, r2, fmaf(-5.f, r4, r6)), oc = fmaf(2.f, r1, fmaf(-2.5f, r3, 0.5f * r5));
    // (LDS operations of one wave execute in order: every lane has read r7 = its neighbour's first sample by now)
    r[0] = p0;
    r[2] = ea + oa;
    r[4] = ea - oa;
    r[6] = eb + ob;
    r[8] = eb - ob;
    r[10] = ec + oc;
    r[12] = ec - oc;
    idst[tch * XI + tun] = p7;
  };

  // ---- accumulators: point waves slots 0..6 = the finite points, direct waves slots 0..6 = output positions of the unit
  f32x16 acc[7];
#pragma unroll
  for (int s = 0; s < 7; ++s)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[s][e] = 0.f;

  // A fragments, packed in walk order.  Point waves: per (32-row block, round, half) four float4 per lane, float4 s = slots 2 s and
  // 2 s + 1 (U_p of the finite points) x the half's two k-steps.  Direct waves: w1 in all of slots 0..6 and U_inf in slot 7 -- the
  // same register shape, so both kinds run the same code -- from TWO float4 per (block, round, half): (w1, w1) x 2 k-steps, read
  // three times (hits in L1), and (w1, U_inf).  (The first build replicated w1 in memory: 37 % more fragment traffic out of L2.)
  const int nsub = a.M / 32;
  const int ms = mt * 2 + mi;
  const int hstride = kind ? 2 : 4;               // float4 blocks per (round, half)
  const f32x4* const wpk = reinterpret_cast<const f32x4*>(a.wpack) +
                           (kind ? (size_t)nsub * nround * 8 + (size_t)ms * nround * 4 : (size_t)ms * nround * 8) * 64 + lane;
  f32x4 a0[4], a1[4];  // half 0 / half 1 of a round
  auto load_a = [&](f32x4 (&dst)[4], int rd, int half) __attribute__((always_inline)) {
    if constexpr (DBG & 16) {
      if (rd > 0) return;  // (diagnostics: the first round's fragments for ever)
    }
    const f32x4* q = wpk + (size_t)(rd * 2 + half) * hstride * 64;
    const int s3 = kind ? 64 : 3 * 64, s1 = kind ? 0 : 64;
    dst[0] = q[0];
    dst[1] = q[s1];
    dst[2] = q[2 * s1];
    dst[3] = q[s3];
  };

  // B fragments: slot s of k-step ks = row 2 ks + h, float 14 l31 + 2 s of the round's window -- the transformed even
  // plane for the point waves, the raw odd plane (+ 1) for the direct waves
  const int boff = h * RS + 14 * l31 + kind;
  const int ioff = h * XI + l31;

  // the MFMAs of one half round (k-steps 2 half, 2 half + 1)
  auto half_round = [&](const f32x4 (&af)[4], int wsel, int half) __attribute__((always_inline)) {
    if constexpr (DBG & 2) return;
    const float* bsrc = win + wsel * S2_WIN + boff + half * (4 * RS);
    const float* isrc = vinf + wsel * S2_INF + ioff + half * (4 * XI);
    float bk[2][7], bi[2];
    if constexpr (DBG & 32) {  // (diagnostics: no B fragment reads)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int sl = 0; sl < 7; ++sl) bk[kk][sl] = (float)(lane + sl);
      bi[0] = bi[1] = 1.f;
    } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int sl = 0; sl < 7; ++sl) bk[kk][sl] = bsrc[kk * 2 * RS + 2 * sl];
      if (kind) {
        bi[0] = isrc[0];
        bi[1] = isrc[2 * XI];
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise hoists the next phases' LDS reads up here until it spills)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int sl = 0; sl < 7; ++sl)
        acc[sl] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[sl >> 1][(sl & 1) * 2 + kk], bk[kk][sl], acc[sl], 0, 0, 0);
    }
    if (kind) {  // U_inf V_inf -> the unit's last output
      acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3][2], bi[0], acc[6], 0, 0, 0);
      acc[6] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3][3], bi[1], acc[6], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: window 0 and 1 in LDS, V[0] formed, window 2 in registers, A fragments of round 0 / half 0.
  // Round indices beyond the last one are clamped, never branched on: the extra copies land in buffers nobody reads any more.
  const int last = nround - 1;
  auto rc = [&](int rd) { return rd < last ? rd : last; };
  stage_load(0);
  stage_store(win);
  stage_load(rc(1));
  __syncthreads();
  transform(win, vinf);
  stage_store(win + S2_WIN);
  stage_load(rc(2));
  __syncthreads();

  load_a(a0, 0, 0);
  load_a(a1, 0, 1);
  // Every wave issues its loads in the same order, and ALL of a round's A loads go out in the middle of the previous round (behind
  // its first half's MFMAs): the compiler's vmcnt waits are conservative around the loop's back edge -- at the first MFMA of a
  // round it waits for everything but the newest window loads -- so a fragment set fetched just before the barrier is waited for
  // with its whole L2 latency exposed (measured: +600 us per conv1).  The second half's fragments therefore alternate between two
  // register sets (a1, a2) and the loop is unrolled by two rounds.  The phases of a workgroup's round do not overlap each other
  // (one barrier per round); the CU's OTHER workgroup fills them.
  f32x4 a2[4];
  int wb = 0;  // rd % 3
  auto round = [&](int rd, const f32x4 (&h1)[4], f32x4 (&h1n)[4]) __attribute__((always_inline)) {
    const int wb1 = wb == 2 ? 0 : wb + 1, wb2 = wb1 == 2 ? 0 : wb1 + 1;
    // window rd + 2 (in registers for a round) goes into the buffer round rd - 1's direct waves read last
    stage_store(win + wb2 * S2_WIN);
    stage_load(rc(rd + 3));
    __builtin_amdgcn_sched_barrier(0);
    transform(win + wb1 * S2_WIN, vinf + wb1 * S2_INF);
    __builtin_amdgcn_sched_barrier(0);
    half_round(a0, wb, 0);
    load_a(a0, rc(rd + 1), 0);
    load_a(h1n, rc(rd + 1), 1);
    half_round(h1, wb, 1);
    __syncthreads();
    wb = wb1;
  };
  for (int rd = 0; rd < nround; rd += 2) {  // (nround is even: s2tc_supported)
    round(rd, a1, a2);
    round(rd + 1, a2, a1);
  }

  // ---- epilogue.  C/D layout of 32x32x2: column = lane & 31 (unit), row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5).
  if constexpr (DBG & 4) {
    float chk = 0.f;  // (keeps every accumulator alive)
#pragma unroll
    for (int sl = 0; sl < 7; ++sl)
#pragma unroll
      for (int e = 0; e < 16; ++e) chk += acc[sl][e];
    if (chk == 123.f) a.out[0] = 1.f;
    return;
  }
  float* const ot = lds;  // [64][OS]: output position 7 unit + j of the tile
  float* const oq = ot + (32 * mi + 4 * h) * OS + S2_MO * l31;
  if (kind == 1) {
#pragma unroll
    for (int j = 0; j < S2_MO; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) oq[((e & 3) + 8 * (e >> 2)) * OS + j] = acc[j][e];
  }
  __syncthreads();
  if (kind == 0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float* o = oq + ((e & 3) + 8 * (e >> 2)) * OS;
      const float y0 = acc[0][e], y1 = acc[1][e], y2 = acc[2][e], y3 = acc[3][e], y4 = acc[4][e], y5 = acc[5][e], y6 = acc[6][e];
      const float s12 = y1 + y2, d12 = y1 - y2, s34 = y3 + y4, d34 = y3 - y4, s56 = y5 + y6, d56 = y5 - y6;
      o[0] += (y0 + s12) + (s34 + s56);
      o[1] += fmaf(2.f, d34, fmaf(0.5f, d56, d12));
      o[2] += fmaf(4.f, s34, fmaf(0.25f, s56, s12));
      o[3] += fmaf(8.f, d34, fmaf(0.125f, d56, d12));
      o[4] += fmaf(16.f, s34, fmaf(0.0625f, s56, s12));
      o[5] += fmaf(32.f, d34, fmaf(0.03125f, d56, d12));
      o[6] += fmaf(64.f, s34, fmaf(0.015625f, s56, s12));  // (+ Y_inf: already in the direct accumulator)
    }
  }
  __syncthreads();
  constexpr int QPR = S2_OT / 4;               // float4 per output row
  constexpr int NIT = 64 * QPR / NTH;          // 14
  static_assert(64 * QPR % NTH == 0, "whole passes");
  const size_t ob = (size_t)b * a.o_bstride + (size_t)(mt * 64) * a.ldo;
#pragma unroll 2
  for (int it = 0; it < NIT; ++it) {
    const int idx = tid + it * NTH;
    const int row = idx / QPR, c4 = idx - row * QPR;
    const int t = t0 + 4 * c4;
    if (t >= olen) continue;
    f32x4 v = *reinterpret_cast<const f32x4*>(ot + row * OS + 4 * c4);
    if (a.bias) {
      const float bz = a.bias[mt * 64 + row];
      v[0] += bz; v[1] += bz; v[2] += bz; v[3] += bz;
    }
    if (a.act == 1) {
      v[0] = gelu_exact(v[0]); v[1] = gelu_exact(v[1]); v[2] = gelu_exact(v[2]); v[3] = gelu_exact(v[3]);
    }
    float* dst = a.out + ob + (size_t)row * a.ldo + t;
    if (olen - t >= 4) {
      *reinterpret_cast<f32x4*>(dst) = v;
    } else {
      for (int e = 0; e < olen - t; ++e) dst[e] = v[e];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
bool s2tc_supported(int Cout, int Cin, int KS, int stride) {
  return KS == 3 && stride == 2 && Cout % 64 == 0 && Cin % (2 * S2_CPR) == 0;  // (the round loop is unrolled by two)
}

// G at the points 0, 1, -1, 2, -2, 1/2, -1/2, inf for a TWO-tap filter (g0, g1) = (w[..][0], w[..][2]) on the even samples
// (conv_wino8.hip's w8_g with R = 2): U_p = scale_p (g0 + x_p g1), U_inf = g1; slot 8 = w[..][1], the odd samples' tap.
int make_s2tc(const float* w, const float* bias, int Cout, int Cin, DevS2tc& dc) {
  if (!s2tc_supported(Cout, Cin, 3, 2)) {
    set_error("make_s2tc: %d -> %d channels unsupported (rows %% 64, channels %% 16)", Cin, Cout);
    return DISSC_EINVAL;
  }
  static const double pt[7] = {0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5};
  static const double sc[7] = {-1.0, -2.0 / 9.0, -2.0 / 9.0, 1.0 / 90.0, 1.0 / 90.0, 32.0 / 45.0, 32.0 / 45.0};
  const int nsub = Cout / 32, nround = Cin / S2_CPR;
  auto u_of = [&](int slot, int co, int ci) -> float {  // slots 0..6: finite points, 7: inf, 8: the odd samples' tap
    const float* wk = w + ((size_t)co * Cin + ci) * 3;
    if (slot < 7) return (float)(sc[slot] * ((double)wk[0] + pt[slot] * (double)wk[2]));
    return slot == 7 ? wk[2] : wk[1];
  };
  std::vector<float> packed((size_t)nsub * nround * 12 * 64 * 4);
  size_t o = 0;
  for (int ms = 0; ms < nsub; ++ms)  // point waves
    for (int rd = 0; rd < nround; ++rd)
      for (int half = 0; half < 2; ++half)
        for (int s = 0; s < 4; ++s)
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 4; ++e) {
              const int ks = 2 * half + (e & 1), slot = 2 * s + (e >> 1);
              packed[o++] = slot < 7 ? u_of(slot, ms * 32 + (lane & 31), rd * S2_CPR + 2 * ks + (lane >> 5)) : 0.f;
            }
  for (int ms = 0; ms < nsub; ++ms)  // direct waves: (w1, w1), (w1, U_inf)
    for (int rd = 0; rd < nround; ++rd)
      for (int half = 0; half < 2; ++half)
        for (int s = 0; s < 2; ++s)
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 4; ++e) {
              const int ks = 2 * half + (e & 1);
              packed[o++] = u_of((s == 1 && (e >> 1)) ? 7 : 8, ms * 32 + (lane & 31), rd * S2_CPR + 2 * ks + (lane >> 5));
            }
  dc.CIN = Cin; dc.M = Cout;
  int rc = upload(packed, &dc.wpack);
  if (rc) return rc;
  if (bias) {
    std::vector<float> bz(bias, bias + Cout);
    if ((rc = upload(bz, &dc.bias))) return rc;
  }
  return DISSC_OK;
}

void free_s2tc(DevS2tc& dc) {
  if (dc.wpack) (void)hipFree(dc.wpack);
  if (dc.bias) (void)hipFree(dc.bias);
  dc.wpack = dc.bias = nullptr;
}

// MACs the matrix pipe executes per OUTPUT position and (input, output) channel pair: 15 / 7
double s2tc_executed_macs_per_out(int Cout, int Cin) { return (double)Cout * Cin * 15.0 / 7.0; }

template <int DBG>
static int launch_s2tc_t(const S2tcArgs& a, long long nwg, hipStream_t stream) {
  static DeviceOnce attr_once;  // per device (common.h)
  DISSC_HIP_CHECK(attr_once.max_lds(reinterpret_cast<const void*>(&conv_s2tc_kernel<DBG>), 160 * 1024));
  hipLaunchKernelGGL(conv_s2tc_kernel<DBG>, dim3((unsigned)nwg), dim3(S2_NTH), (size_t)S2_LDS_FLOATS * sizeof(float), stream, a);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

int run_s2tc(const DevS2tc& dc, const float* x, float* out, const int32_t* lengths_in, const int32_t* lengths_out,
             int len_default, int olen_default, int B, int ldx, int ldo, int Lmax_out, hipStream_t stream) {
  auto misaligned = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) != 0; };
  if (B <= 0 || Lmax_out <= 0 || !dc.wpack || ldx < 4 || ldx % 4 || ldo % 4 || misaligned(x) || misaligned(out)) {
    set_error("run_s2tc: bad call (B %d, Lmax_out %d, ldx %d, ldo %d: rows must be 16-byte aligned)", B, Lmax_out, ldx, ldo);
    return DISSC_EINVAL;
  }
  S2tcArgs a;
  a.x = x; a.wpack = dc.wpack; a.bias = dc.bias; a.out = out;
  a.lengths_in = lengths_in; a.lengths_out = lengths_out; a.len_default = len_default; a.olen_default = olen_default;
  a.CIN = dc.CIN; a.M = dc.M; a.act = dc.act; a.ldx = ldx; a.ldo = ldo;
  a.x_bstride = (long long)dc.CIN * ldx; a.o_bstride = (long long)dc.M * ldo;
  a.gx = (Lmax_out + S2_OT - 1) / S2_OT;
  a.nrt = dc.M / 64;
  a.B = B;
  a.dbg = 0;
  a.xmode = (opts().s2tc_xmode == 0 && a.nrt <= 8 && 8 % a.nrt == 0) ? 0 : 1;
  const long long ntile = (long long)a.gx * B;
  long long nwg;
  if (a.xmode == 0) {
    const int per = 8 / a.nrt;
    nwg = 8 * ((ntile + per - 1) / per);
  } else {
    nwg = 8LL * a.nrt * ((ntile + 7) / 8);
  }
  switch (opts().kernel_dbg) {
    case 0: return launch_s2tc_t<0>(a, nwg, stream);
    case 1: return launch_s2tc_t<1>(a, nwg, stream);
    case 2: return launch_s2tc_t<2>(a, nwg, stream);
    case 3: return launch_s2tc_t<3>(a, nwg, stream);
    case 4: return launch_s2tc_t<4>(a, nwg, stream);
    case 7: return launch_s2tc_t<7>(a, nwg, stream);
    case 8: return launch_s2tc_t<8>(a, nwg, stream);
    case 15: return launch_s2tc_t<15>(a, nwg, stream);
    case 13: return launch_s2tc_t<13>(a, nwg, stream);
    case 29: return launch_s2tc_t<29>(a, nwg, stream);
    case 45: return launch_s2tc_t<45>(a, nwg, stream);
    case 61: return launch_s2tc_t<61>(a, nwg, stream);
    case 9: return launch_s2tc_t<9>(a, nwg, stream);
    default: set_error("run_s2tc: no instance for s2tc_dbg = %d", opts().kernel_dbg); return DISSC_EINVAL;
  }
}


}  // namespace dissc
