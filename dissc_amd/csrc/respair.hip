// respair16_kernel / respair32_kernel: one residual pair of a ResBlock1 of the generator's narrow stages,
//
//     y = x + conv_1(lrelu(conv_d(lrelu(x))))            (k taps each; reference sr/models.py:34-41)
//
// in ONE launch, exact fp32 on the matrix pipe, followed by the same epilogue modes as the conv
// kernels (x_k update or the MRF accumulate of the stage).  These stages (C = 16 / 32 channels at
// 80 000 / 160 000 samples per 10 s) are HBM-bound when every conv is its own launch: a pair then
// moves 5 activation passes (read x, write t, read t, read x, write y) where this kernel moves 2.
//
// Why a PAIR and not the whole block: the halo of one pair is only (k-1)/2 columns of the
// intermediate (<= 5), so nothing worth mentioning is recomputed, the LDS footprint stays at
// ~40 KB (3-4 workgroups per CU overlap each other's staging, MFMA and store phases) and the three
// chains of a stage keep running concurrently on their streams (only a chain's LAST pair waits for
// the MRF accumulator).  The whole-block kernel (resblock_fused.hip, rounds 1-2, removed) recomputed a 60-column halo per
// side and was break-even.
//
// fp32 VALU work steals issue slots from fp32 MFMAs on gfx950 (tools/pipe_overlap.py), so the tap
// loops contain nothing but LDS fragment reads and MFMAs: leaky-ReLU, the utterance's zero padding
// and the ragged tail are applied ONCE per element -- when the staged window is written to LDS and in
// the epilogue of the first conv -- and all LDS offsets are compile-time constants.
//
// Bit-level: every output element sees the MFMA sequence of the unfused kernels (taps outer,
// 4-channel k-steps inner, bias / residual / MRF in the same order), so results are identical to the
// two-launch path; tests/test_gpu_generator.py holds the two against each other bitwise.
#include "common.h"

namespace dissc {

struct PairArgs {
  const float* x;     // [B][C][ld] pair input x_k
  float* out;         // EPI_RES: x_k' (may not alias x: neighbouring workgroups still read x's halo)
  float* acc;         // EPI_MRF_*: the stage accumulator
  const float* w1;    // packed conv weights (DevConv::wpack of the dilated conv / of the dil-1 conv)
  const float* w2;
  const float* b1;    // [C]
  const float* b2;
  const int32_t* lengths;
  int len_default, len_mul;
  int ld;
  long long bstride;
  float slope, mrf_div;
  int epi;
};

__device__ __forceinline__ float lrelu_p(float v, float slope) { return v > 0.f ? v : v * slope; }
constexpr int round32_16(int n) { return (n - 16 + 31) / 32 * 32 + 16; }  // smallest v >= n with v % 32 == 16

// rows of the 16 B-per-lane epilogue: v = conv + bias (+ residual) -> out / MRF accumulate
__device__ __forceinline__ void pair_store4(const PairArgs& a, size_t idx, f32x4 v, const f32x4& rv, int nv) {
  const int epi = a.epi;
  if (nv >= 4) {
    v[0] += rv[0]; v[1] += rv[1]; v[2] += rv[2]; v[3] += rv[3];
    if (epi == EPI_RES) {
      *reinterpret_cast<f32x4*>(a.out + idx) = v;
    } else if (epi == EPI_MRF_SET) {
      *reinterpret_cast<f32x4*>(a.acc + idx) = v;
    } else {
      const f32x4 ac = *reinterpret_cast<const f32x4*>(a.acc + idx);
      v[0] = ac[0] + v[0]; v[1] = ac[1] + v[1]; v[2] = ac[2] + v[2]; v[3] = ac[3] + v[3];
      if (epi == EPI_MRF_DIV) {
        v[0] = __fdiv_rn(v[0], a.mrf_div); v[1] = __fdiv_rn(v[1], a.mrf_div);
        v[2] = __fdiv_rn(v[2], a.mrf_div); v[3] = __fdiv_rn(v[3], a.mrf_div);
      }
      *reinterpret_cast<f32x4*>(a.acc + idx) = v;
    }
  } else {
    for (int e = 0; e < nv; ++e) {
      float x = v[e] + rv[e];
      if (epi == EPI_RES) {
        a.out[idx + e] = x;
      } else if (epi == EPI_MRF_SET) {
        a.acc[idx + e] = x;
      } else {
        x = a.acc[idx + e] + x;
        if (epi == EPI_MRF_DIV) x = __fdiv_rn(x, a.mrf_div);
        a.acc[idx + e] = x;
      }
    }
  }
}


// (b, first output column) of workgroup `lin` when only the tiles that EXIST are enumerated: utterance 0's
// ceil(len_0 / WOUT) tiles, then utterance 1's, ...  The grid still holds gridDim.x tiles for each of the B utterances;
// the workgroups beyond the last real tile all sit at the END of the dispatch order and return at once -- enumerating
// (tile, utterance) pairs and returning from the tiles beyond an utterance's end leaves the empty workgroups between the
// real ones (7-9 % on ragged batches for conv_wino_kernel, +0.65 ms per ragged forward for these stages).  Every wave
// finds its pair by a prefix sum of the tile counts over its lanes (the scheme of conv_wino.hip).
template <int WOUT>
__device__ __forceinline__ bool pair_tile(const PairArgs& a, int B, int& b, int& len, int& o0) {
  const int lin = blockIdx.y * gridDim.x + blockIdx.x;
  if (a.lengths == nullptr) {
    b = blockIdx.y;
    len = a.len_default;
    o0 = blockIdx.x * WOUT;
    return o0 < len;
  }
  const int lane = threadIdx.x & 63;
  int base = 0;
  for (int b0 = 0; b0 < B; b0 += 64) {
    const int l = b0 + lane < B ? a.lengths[b0 + lane] * a.len_mul : 0;
    const int nt = (l + WOUT - 1) / WOUT;
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
      b = __builtin_amdgcn_readfirstlane(b0 + lb);
      len = __builtin_amdgcn_readfirstlane(__shfl(l, lb, 64));
      o0 = __builtin_amdgcn_readfirstlane((lin - base - __shfl(incl - nt, lb, 64)) * WOUT);
      return true;
    }
    base += total;
  }
  return false;
}

// ---- C = 16: v_mfma_f32_16x16x4_f32, both convs' weights in registers -------------------------
template <int KS, int DIL, int NI>
__global__ void __launch_bounds__(256) respair16_kernel(const PairArgs a) {
  constexpr int C = 16, NW = 4, NT = 64 * NW;
  constexpr int SLOTS = 16 * NI * NW;           // columns computed per conv and workgroup
  constexpr int P2 = (KS - 1) / 2, P1 = P2 * DIL;
  constexpr int WOUT = (SLOTS - 2 * P2) & ~3;   // columns of y a workgroup owns
  constexpr int XW1 = round32_16(3 + SLOTS + 2 * P1);
  constexpr int XW2 = round32_16(SLOTS + 2 * P2);
  constexpr int NV = XW1 / 4;
  constexpr int SV = (C * NV + NT - 1) / NT;
  constexpr int CW = 16 * NI + 4;
  constexpr int LPR = 4 * NI, RPP = 64 / LPR;
  static_assert(NW * 16 * CW <= C * XW1, "epilogue patches alias the input window");
  __shared__ __attribute__((aligned(16))) float Xs[C * XW1];  // lrelu(x) window (later: epilogue patches)
  __shared__ __attribute__((aligned(16))) float Ts[C * XW2];  // lrelu(conv1 + b1), 0 outside the utterance

  int b, len, o0;
  if (!pair_tile<WOUT>(a, gridDim.y, b, len, o0)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int tin0 = o0 - P2 - P1;
  const int tb = tin0 & ~3, sh = tin0 - tb;
  const float slope = a.slope;
  const float* xb = a.x + (size_t)b * a.bstride;

  f32x4 wa[KS];
  {
    const __amdgpu_buffer_rsrc_t w1r = wave_rsrc(a.w1, 0x7ffffff0u);  // scalar-base loads (common.h)
#pragma unroll
    for (int j = 0; j < KS; ++j) wa[j] = rsrc_load16(w1r, lane * 16u, j * 1024u);
  }
  const f32x4 bias1 = *reinterpret_cast<const f32x4*>(a.b1 + 4 * g);

  // ---- stage lrelu(x) on [tb, tb + XW1) into LDS (16 B per lane, clamped unconditional loads) ----
  {
    f32x4 sv[SV];
    int r = tid / NV, v = tid - (tid / NV) * NV;
    constexpr int dr = NT / NV, dv = NT - dr * NV;
    int rr = r, vv = v;
#pragma unroll
    for (int i = 0; i < SV; ++i) {
      const int ci = rr < C ? rr : C - 1;
      int t = tb + 4 * vv;
      t = t < 0 ? 0 : (t > a.ld - 4 ? a.ld - 4 : t);
      sv[i] = *reinterpret_cast<const f32x4*>(xb + (size_t)ci * a.ld + t);
      vv += dv; rr += dr;
      if (vv >= NV) { vv -= NV; ++rr; }
    }
    rr = r; vv = v;
#pragma unroll
    for (int i = 0; i < SV; ++i) {
      if (rr < C) {
        const int t = tb + 4 * vv;
        f32x4 val = sv[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) val[e] = ((t + e) >= 0 && (t + e) < len) ? lrelu_p(val[e], slope) : 0.f;
        *reinterpret_cast<f32x4*>(Xs + rr * XW1 + 4 * vv) = val;
      }
      vv += dv; rr += dr;
      if (vv >= NV) { vv -= NV; ++rr; }
    }
  }
  __syncthreads();

  // ---- conv_d: t = conv(lrelu(x)) on slots [o0 - P2, o0 - P2 + SLOTS) ----
  f32x4 acc[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) acc[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const float* bj = Xs + g * XW1 + sh + wave * (16 * NI) + l15;
#pragma unroll
    for (int j = 0; j < KS; ++j) {
#pragma unroll
      for (int cq = 0; cq < 4; ++cq)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j][cq], bj[cq * 4 * XW1 + ni * 16 + j * DIL], acc[ni], 0, 0, 0);
      if (j & 1) __builtin_amdgcn_sched_barrier(0);  // bounds how far the LDS reads are hoisted (register pressure)
    }
  }
  // conv_1's weights take over the registers of conv_d's: fetched behind its last MFMAs / the barrier
  f32x4 wb[KS];
  {
    const __amdgpu_buffer_rsrc_t w2r = wave_rsrc(a.w2, 0x7ffffff0u);
#pragma unroll
    for (int j = 0; j < KS; ++j) wb[j] = rsrc_load16(w2r, lane * 16u, j * 1024u);
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int u = wave * (16 * NI) + ni * 16 + l15;
    const int t = o0 - P2 + u;
    const bool inside = t >= 0 && t < len;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = lrelu_p(acc[ni][r] + bias1[r], slope);
      Ts[(4 * g + r) * XW2 + u] = inside ? v : 0.f;
    }
  }
  __syncthreads();

  // ---- conv_1 on slots [o0, o0 + SLOTS); the first WOUT of them are this workgroup's output ----
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) acc[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  // residual rows of this wave's columns (raw x: L2-hot, this workgroup staged it a moment ago)
  const int prow = lane / LPR, pc4 = lane % LPR;
  const int ncol = wave * (16 * NI) + 4 * pc4;  // column within the slots
  const int tcol = o0 + ncol;
  const size_t ob = (size_t)b * a.bstride;
  f32x4 rv[NI];
#pragma unroll
  for (int p = 0; p < NI; ++p) {
    const int row = p * RPP + prow;
    const int tc = tcol > a.ld - 4 ? a.ld - 4 : tcol;
    rv[p] = *reinterpret_cast<const f32x4*>(a.x + ob + (size_t)row * a.ld + tc);
  }
  {
    const float* bt = Ts + g * XW2 + wave * (16 * NI) + l15;
#pragma unroll
    for (int j = 0; j < KS; ++j) {
#pragma unroll
      for (int cq = 0; cq < 4; ++cq)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[j][cq], bt[cq * 4 * XW2 + ni * 16 + j], acc[ni], 0, 0, 0);
      if (j & 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- epilogue: registers -> wave-private patch (aliases Xs: dead since the last barrier) -> 16 B per lane ----
  float* patch = Xs + wave * (16 * CW);
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int r = 0; r < 4; ++r) patch[(4 * g + r) * CW + ni * 16 + l15] = acc[ni][r];
  __builtin_amdgcn_wave_barrier();
  if (ncol < WOUT && tcol < len) {
#pragma unroll
    for (int p = 0; p < NI; ++p) {
      const int row = p * RPP + prow;
      f32x4 v = *reinterpret_cast<const f32x4*>(patch + row * CW + 4 * pc4);
      const float bz = a.b2[row];
      v[0] += bz; v[1] += bz; v[2] += bz; v[3] += bz;
      pair_store4(a, ob + (size_t)row * a.ld + tcol, v, rv[p], len - tcol);
    }
  }
}

// ---- C = 32: v_mfma_f32_32x32x2_f32 (64-cycle form), weights streamed tap by tap from L2 ---------
// A fragments in the layout of pack_conv_weights32: [chunk c][tap][half hf][lane][4], k-step ks = 4*hf + e
// of chunk c covers channels 16*c + 2*ks + (lane >> 5).  One step = one (chunk, tap) = 8 k-steps = 2 float4
// per lane, walked chunk-outer / tap-inner like conv_mfma32_kernel (identical accumulation order); the
// next step's fragments are fetched while the current 8*NI MFMAs run (sched_barrier pins the prefetch).
typedef float f32x16p __attribute__((ext_vector_type(16)));

// LM (LDS mode): 0 = separate windows for lrelu(x) and T (70-78 KB: two workgroups per CU); 1 (default) = T overwrites
// the x window after one more barrier, wave-private epilogue patches in their own 8.7 KB (44-52 KB: three per CU,
// -0.43 ms per forward).  Letting the patches overwrite the window too (35-43 KB, four per CU) measured the same.
template <int KS, int DIL, int NI, int LM>
__global__ void __launch_bounds__(256) respair32_kernel(const PairArgs a) {
  constexpr int C = 32, NW = 4, NT = 64 * NW;
  constexpr int SLOTS = 32 * NI * NW;
  constexpr int P2 = (KS - 1) / 2, P1 = P2 * DIL;
  constexpr int WOUT = (SLOTS - 2 * P2) & ~3;
  constexpr int XW1 = round32_16(3 + SLOTS + 2 * P1);
  constexpr int XW2 = round32_16(SLOTS + 2 * P2);
  constexpr int NV = XW1 / 4;
  constexpr int SV = (C * NV + NT - 1) / NT;
  constexpr int CW = 32 * NI + 4;
  constexpr int LPR = 8 * NI, RPP = 64 / LPR, NPASS = 8 / RPP;
  static_assert(NW * 8 * CW <= C * XW1, "epilogue patches alias the input window");
  static_assert(XW2 <= XW1, "T fits the x window");
  __shared__ __attribute__((aligned(16))) float Xs[C * XW1];
  __shared__ __attribute__((aligned(16))) float Tsep[LM == 0 ? C * XW2 : 4];
  __shared__ __attribute__((aligned(16))) float Psep[LM == 1 ? NW * 8 * CW : 4];
  float* const Ts = LM == 0 ? Tsep : Xs;

  int b, len, o0;
  if (!pair_tile<WOUT>(a, gridDim.y, b, len, o0)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int tin0 = o0 - P2 - P1;
  const int tb = tin0 & ~3, sh = tin0 - tb;
  const float slope = a.slope;
  const float* xb = a.x + (size_t)b * a.bstride;
  // scalar-base loads (common.h: wave_rsrc): the weights are the same for every wave, offsets are compile-time constants
  const __amdgpu_buffer_rsrc_t w1p = wave_rsrc(a.w1, 0x7ffffff0u), w2p = wave_rsrc(a.w2, 0x7ffffff0u);
  const unsigned lane16 = lane * 16u;
  // step q = c*KS + j (chunk outer, tap inner: the order of conv_mfma32_kernel) = 8 k-steps = 2 float4:
  // wp[(q*2 + hf)*64]
  f32x4 av[2], avn[2];
  av[0] = rsrc_load16(w1p, lane16, 0);
  av[1] = rsrc_load16(w1p, lane16, 1024u);

  {
    int rr = tid / NV, vv = tid - (tid / NV) * NV;
    constexpr int dr = NT / NV, dv = NT - dr * NV;
    // two half-batches keep the staging registers low (SV float4 would be ~40 VGPRs)
    constexpr int SVH = (SV + 1) / 2;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      f32x4 sv[SVH];
      int r1 = rr, v1 = vv;
#pragma unroll
      for (int i = 0; i < SVH; ++i) {
        const int ci = r1 < C ? r1 : C - 1;
        int t = tb + 4 * v1;
        t = t < 0 ? 0 : (t > a.ld - 4 ? a.ld - 4 : t);
        sv[i] = *reinterpret_cast<const f32x4*>(xb + (size_t)ci * a.ld + t);
        v1 += dv; r1 += dr;
        if (v1 >= NV) { v1 -= NV; ++r1; }
      }
#pragma unroll
      for (int i = 0; i < SVH; ++i) {
        if (rr < C) {
          const int t = tb + 4 * vv;
          f32x4 val = sv[i];
#pragma unroll
          for (int e = 0; e < 4; ++e) val[e] = ((t + e) >= 0 && (t + e) < len) ? lrelu_p(val[e], slope) : 0.f;
          *reinterpret_cast<f32x4*>(Xs + rr * XW1 + 4 * vv) = val;
        }
        vv += dv; rr += dr;
        if (vv >= NV) { vv -= NV; ++rr; }
      }
    }
  }
  __syncthreads();

  f32x16p acc[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[ni][e] = 0.f;

#define DISSC_PAIR32_TAPS(WP, BASE, XW, STEP, NEXT_WP)                                                     \
  _Pragma("unroll") for (int q = 0; q < 2 * KS; ++q) {                                                     \
    if (q + 1 < 2 * KS) {                                                                                  \
      avn[0] = rsrc_load16(WP, lane16, ((q + 1) * 2 + 0) * 1024u);                                         \
      avn[1] = rsrc_load16(WP, lane16, ((q + 1) * 2 + 1) * 1024u);                                         \
    } else {                                                                                               \
      avn[0] = rsrc_load16(NEXT_WP, lane16, 0);                                                            \
      avn[1] = rsrc_load16(NEXT_WP, lane16, 1024u);                                                        \
    }                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    _Pragma("unroll") for (int hf = 0; hf < 2; ++hf)                                                       \
      _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                        \
        _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                                  \
          acc[ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(                                                  \
              av[hf][e], BASE[((q / KS) * 16 + 2 * (4 * hf + e)) * XW + ni * 32 + (q % KS) * STEP], acc[ni], 0, 0, 0); \
    __builtin_amdgcn_sched_barrier(0);                                                                     \
    av[0] = avn[0];                                                                                        \
    av[1] = avn[1];                                                                                        \
  }

  {
    const float* bj = Xs + h * XW1 + sh + wave * (32 * NI) + l31;
    DISSC_PAIR32_TAPS(w1p, bj, XW1, DIL, w2p)  // leaves conv_1's first tap in av
  }
  // epilogue 1: T = lrelu(conv_d + b1), 0 outside the utterance.  D layout: col = lane & 31,
  // row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
  if constexpr (LM != 0) __syncthreads();  // every wave is done reading the x window T is about to overwrite
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int u = wave * (32 * NI) + ni * 32 + l31;
    const int t = o0 - P2 + u;
    const bool inside = t >= 0 && t < len;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      const float v = lrelu_p(acc[ni][r] + a.b1[row], slope);
      Ts[row * XW2 + u] = inside ? v : 0.f;
    }
  }
  __syncthreads();
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[ni][e] = 0.f;
  {
    const float* bt = Ts + h * XW2 + wave * (32 * NI) + l31;
    DISSC_PAIR32_TAPS(w2p, bt, XW2, 1, w2p)  // (the last prefetch re-reads tap 0: harmless)
  }
#undef DISSC_PAIR32_TAPS
  // epilogue 2: 8 rows at a time through a wave-private patch [8][CW] -> 16 B per lane
  float* ep = (LM == 1 ? Psep : Xs) + wave * (8 * CW);
  const int prow = lane / LPR, pc4 = lane % LPR;
  const int ncol = wave * (32 * NI) + 4 * pc4;
  const int tcol = o0 + ncol;
  const size_t ob = (size_t)b * a.bstride;
  const bool live = ncol < WOUT && tcol < len;
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {  // rows 8*qd .. 8*qd+7
    // residual rows of this pass (raw x, L2-hot), issued before the patch round trip
    f32x4 rv[NPASS];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int row = 8 * qd + p * RPP + prow;
      const int tc = tcol > a.ld - 4 ? a.ld - 4 : tcol;
      rv[p] = *reinterpret_cast<const f32x4*>(a.x + ob + (size_t)row * a.ld + tc);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) ep[(r + 4 * h) * CW + ni * 32 + l31] = acc[ni][4 * qd + r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int prw = p * RPP + prow;
      f32x4 v = *reinterpret_cast<const f32x4*>(ep + prw * CW + 4 * pc4);
      if (!live) continue;
      const int row = 8 * qd + prw;
      const float bz = a.b2[row];
      v[0] += bz; v[1] += bz; v[2] += bz; v[3] += bz;
      pair_store4(a, ob + (size_t)row * a.ld + tcol, v, rv[p], len - tcol);
    }
  }
}

template <int KS, int DIL>
static int launch_pair32(const PairArgs& a, int B, int Lmax, hipStream_t stream) {
  constexpr int NI = 2, SLOTS = 32 * NI * 4, WOUT = (SLOTS - (KS - 1)) & ~3;
  dim3 grid((Lmax + WOUT - 1) / WOUT, B);
  if (opts().pair_lds_mode == 0)
    hipLaunchKernelGGL((respair32_kernel<KS, DIL, NI, 0>), grid, dim3(256), (size_t)opts().pair_pad_lds, stream, a);
  else
    hipLaunchKernelGGL((respair32_kernel<KS, DIL, NI, 1>), grid, dim3(256), (size_t)opts().pair_pad_lds, stream, a);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

template <int KS, int DIL>
static int launch_pair16(const PairArgs& a, int B, int Lmax, hipStream_t stream) {
  constexpr int NI = 4, SLOTS = 16 * NI * 4, WOUT = (SLOTS - (KS - 1)) & ~3;
  dim3 grid((Lmax + WOUT - 1) / WOUT, B);
  hipLaunchKernelGGL((respair16_kernel<KS, DIL, NI>), grid, dim3(256), (size_t)opts().pair_pad_lds, stream, a);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

// option "pair_lds" (Options::pair_lds_mode, default 1): "pair_lds" option: LDS layout of respair32 (see LM)
// option "pair_pad_lds" (Options::pair_pad_lds, default 0): diagnostics: extra dynamic LDS bytes per workgroup (lowers occupancy)
// option "pair_max_c" (Options::pair_max_c, default 32): "pair_max_c" option: widest stage run as fused residual pairs (0 = off)

bool respair_supported(int C, int KS, int dil) {
  if (C > opts().pair_max_c) return false;
  if (C != 16 && C != 32) return false;
  return (KS == 3 || KS == 7 || KS == 11) && (dil == 1 || dil == 3 || dil == 5);
}

// c1 / c2: the DevConvs of the dilated conv and of the dil-1 conv (fp32, 16x16x4 packing)
int launch_respair(const DevConv& c1, const DevConv& c2, const float* x, float* out, float* acc,
                   const int32_t* lengths, int len_default, int len_mul, int B, int Lmax, int ld, float slope,
                   int epi, float mrf_div, hipStream_t stream) {
  const int C = c1.M;
  if (!respair_supported(C, c1.KS, c1.dil) || c2.KS != c1.KS || c2.dil != 1 || c1.m32 != (C == 32) ||
      c2.m32 != (C == 32) || c1.prec || c2.prec || c1.CIN != C || c2.CIN != C || c2.M != C || epi == EPI_STORE || x == out) {
    set_error("launch_respair: unsupported pair (C=%d k=%d d=%d)", C, c1.KS, c1.dil);
    return DISSC_EINVAL;
  }
  PairArgs a;
  a.x = x; a.out = out; a.acc = acc; a.w1 = c1.wpack; a.w2 = c2.wpack; a.b1 = c1.bias; a.b2 = c2.bias;
  a.lengths = lengths; a.len_default = len_default; a.len_mul = len_mul; a.ld = ld;
  a.bstride = (long long)C * ld; a.slope = slope; a.mrf_div = mrf_div; a.epi = epi;
#define DISSC_PAIR16(K, D)                                                        \
  if (c1.KS == K && c1.dil == D)                                                  \
    return C == 16 ? launch_pair16<K, D>(a, B, Lmax, stream) : launch_pair32<K, D>(a, B, Lmax, stream);
  DISSC_PAIR16(3, 1) DISSC_PAIR16(3, 3) DISSC_PAIR16(3, 5)
  DISSC_PAIR16(7, 1) DISSC_PAIR16(7, 3) DISSC_PAIR16(7, 5)
  DISSC_PAIR16(11, 1) DISSC_PAIR16(11, 3) DISSC_PAIR16(11, 5)
#undef DISSC_PAIR16
  set_error("launch_respair: no instance");
  return DISSC_EINVAL;
}

}  // namespace dissc
