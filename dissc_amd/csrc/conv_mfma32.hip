// conv_mfma32_kernel: the 32x32x2 (64-cycle) MFMA form of the implicit-GEMM conv, used for every
// layer with >= 32 output rows.  Staging, prefetching and epilogue semantics are those of
// conv_mfma_kernel (conv_mfma.hip); only the fragment geometry differs:
//   A: lane l holds W[row = l & 31][k = l >> 5], B: lane l holds X[k = l >> 5][col = l & 31],
//   D: 16 registers, col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5).
#include <string.h>

#include "common.h"
#include "conv_epilogue32.h"

#ifndef DISSC_LB32
#define DISSC_LB32 2
#endif
#ifndef DISSC_LB32_DMA
#define DISSC_LB32_DMA 4  // the LDS-DMA instances have no staging registers: 4 waves per SIMD
#endif

namespace dissc {


// 32x32x2 form of the kernel in conv_mfma.hip (64-cycle MFMAs sustain ~155 TFLOP/s on this part,
// the 16x16x4 form ~130-139): tile units are 32 rows x 32 columns, a k-step is 2 channels.
// STRIDE: input step per output position (1; 2 for HuBERT's strided feature convs).
// SPAN: largest (KS-1)*dil the staging registers are sized for.
// CPB: 16-channel chunks staged per barrier (4 for 1x1 convs, whose chunk is a single tap).
// DMA (1x1 convs whose input needs no activation): the window goes global -> LDS directly (global_load_lds_dwordx4,
// one wave instruction = 1 KB = four 64-column rows), no staging registers, no VALU.  Nothing is masked: a 1x1 conv's
// output column depends on its own input column only, and columns >= olen are never stored.
template <int MI, int NI, int WM, int WN, int STRIDE, int SPAN, int CPB = 1, bool DMA = false>
__global__ void __launch_bounds__(64 * WM * WN, DMA ? DISSC_LB32_DMA : DISSC_LB32) conv_mfma32_kernel(const ConvArgs a) {
  constexpr int NT = 64 * WM * WN;
  constexpr int BN = 32 * NI * WN;
  constexpr int KCB = KC * CPB;  // channels staged per barrier
  constexpr int XW_MAX = ((BN - 1) * STRIDE + 1 + SPAN + 3 + 31) / 32 * 32 + 16;
  constexpr int SV = (KCB * (XW_MAX / 4) + NT - 1) / NT;  // float4 staging slots per thread
  extern __shared__ __attribute__((aligned(16))) float xs[];  // 2 x [KCB][XW] | NW x [8][32*NI+4] (epilogue)

  int b = blockIdx.z;
  int bx = a.mfast ? blockIdx.y : blockIdx.x;  // time tile
  int by = a.mfast ? blockIdx.x : blockIdx.y;  // (group, M tile)
  int ntile_g = a.mfast ? gridDim.y : gridDim.x, nb_g = gridDim.z;
  if (a.xcd) {
    // XCD order (1-D grid): the hardware deals workgroup ids round-robin over the 8 XCDs, each with its own 4 MB L2.
    // The ids are cut into SWEEPS over all time tiles, one per group of xcd_mg M tiles (weight slabs that fit the L2
    // together); inside a sweep the xcd_mg M tiles of one time tile -- which read the same input window -- take
    // consecutive slots of ONE XCD.  So a window crosses the fabric once per sweep (not once per M tile) and a weight
    // slab once per XCD (not once per utterance).
    const int mt = a.mt_per_group * a.groups, mg = a.xcd_mg;
    const int sweep = blockIdx.x / a.xcd_span, r = blockIdx.x - sweep * a.xcd_span;
    const int s = r >> 3, sq = s / mg;
    const int tt = (r & 7) + 8 * sq;
    by = sweep * mg + (s - sq * mg);
    ntile_g = a.xcd_ntile;
    nb_g = a.xcd_nb;
    if (tt >= ntile_g * nb_g || by >= mt) return;
    b = tt / ntile_g;
    bx = tt - b * ntile_g;
  }
  if (a.ragged_enum) {
    // Ragged batch: (time tile, utterance) pairs are re-dealt so that only the tiles that EXIST are enumerated --
    // utterance 0's ceil(olen_0 / BN) tiles, then utterance 1's, ... -- and the workgroups left over all sit at the END of
    // the dispatch order (z slowest) and return at once, instead of lying between the real ones (conv_wino.hip has the
    // measurements).  Every wave finds its pair by a prefix sum of the tile counts over its lanes.  Not for
    // EPI_STORE_ACT, whose tiles beyond an utterance's end still have zero tails to write.
    const int ntile = ntile_g;
    const int lin = b * ntile + bx;
    const int lane_ = threadIdx.x & 63;
    const int nb = nb_g;
    int base = 0;
    b = -1;
    for (int b0 = 0; b0 < nb; b0 += 64) {
      int l = 0;
      if (b0 + lane_ < nb)
        l = a.lengths_out ? a.lengths_out[b0 + lane_]
                          : (a.olen_default >= 0 ? a.olen_default : (a.lengths ? a.lengths[b0 + lane_] * a.len_mul : a.len_default));
      const int nt = (l + BN - 1) / BN;
      int incl = nt;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane_ >= o) incl += v;
      }
      const int total = __shfl(incl, 63, 64);
      if (lin < base + total) {
        const unsigned long long m = __ballot(base + incl > lin);
        const int lb = __ffsll((long long)m) - 1;
        b = __builtin_amdgcn_readfirstlane(b0 + lb);
        bx = __builtin_amdgcn_readfirstlane(lin - base - __shfl(incl - nt, lb, 64));
        break;
      }
      base += total;
    }
    if (b < 0) return;
  }
  const int len = (a.lengths ? a.lengths[b] * a.len_mul : a.len_default);  // valid INPUT positions
  const int olen = a.lengths_out ? a.lengths_out[b] : (a.olen_default >= 0 ? a.olen_default : len);
  const int t0 = bx * BN;
  if (t0 >= olen) {
    if (a.epi == EPI_STORE_ACT && by == 0) zero_tail_tile<NT, BN>(a, b, t0, olen);
    return;
  }
  const int grp = by / a.mt_per_group;
  const int mt = by - grp * a.mt_per_group;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;
  const int XW = a.XW;
  const int NV = XW >> 2;
  const int tin0 = t0 * STRIDE - a.pad_left;
  const int tb = tin0 & ~3;   // 16-byte aligned window start (may be negative)
  const int sh = tin0 - tb;   // 0..3
  const int nq = a.nchunk * a.KS;
  const int ms0 = mt * (MI * WM) + wm * MI;  // 32-row subtile within the group
  const float slope = a.slope;
  const float* xb = a.x + (size_t)b * a.x_bstride + (size_t)grp * a.CIN * a.ldx;

  // this thread's float4 staging slots: slot i = tid + i*NT -> (row r, vec v)
  const int r0 = tid / NV, v0 = tid - r0 * NV;
  const int dr = NT / NV, dv = NT - dr * NV;

  // Staging: every slot issues an unconditional, in-bounds 16-byte load (addresses are
  // clamped, never predicated, so nothing waits at the load site); zero padding, the
  // ragged tail and leaky-ReLU are applied when the registers are written to LDS.
  f32x4 sv[DMA ? 1 : SV];
  static_assert(!DMA || (CPB * 16 * 64 % NT == 0 && ((SPAN == 0 && STRIDE == 1 && BN == 64) || CPB == 1)),
                "DMA staging: 1x1 convs (64 columns), the stride-2 valid convs of HuBERT, stride-1 convs on an EPI_STORE_ACT input");
  // DMA windows: row r of a chunk is the raw window [tb, tb + 4 * DW4) of channel r, rows packed back to back (row stride
  // XW = 4 * DW4 floats).  Nothing is clamped or masked:
  //   1x1:       64 columns from t0 (clamped to the row: the only variant that may touch the end of the allocation);
  //   stride 2:  valid convs, [2 t0, 2 t0 + 136) (k <= 9); fragment reads step two floats per lane;
  //   stride 1:  the input has the EPI_STORE_ACT layout -- activated values, zeros in [len, len + 8) and in the last 8
  //              columns of every row, so a row's tail is the left halo of the next row (and 8 zeroed floats precede
  //              the first row): the window may start before the row and run past its end.
  constexpr bool DMA1 = DMA && STRIDE == 1 && SPAN > 0;
  const int DW4 = DMA1 ? (XW >> 2) : ((STRIDE == 2) ? 34 : 16);  // float4 per staged row
  auto stage_dma = [&](float* buf, int c) {
    // slot e = tid + i*NT -> row e / DW4, float4 e % DW4; a wave's 64 slots are contiguous in LDS
    const int nslot = KCB * DW4;
    for (int i = 0; i * NT < nslot; ++i) {
      const int e = tid + i * NT;
      if (e >= nslot) continue;
      const int ci = c * KCB + e / DW4;
      int t = (DMA1 ? tb : tin0) + 4 * (e % DW4);
      if constexpr (!DMA1) t = t > a.ldx - 4 ? a.ldx - 4 : t;
      __builtin_amdgcn_global_load_lds(
          (const void __attribute__((address_space(1)))*)(xb + (ptrdiff_t)ci * a.ldx + t),
          (void __attribute__((address_space(3)))*)(buf + (i * NT + (tid & ~63)) * 4), 16, 0, 0);
    }
  };
  auto stage_load = [&](int c) {
    if constexpr (DMA) return;
    int r = r0, v = v0;
#pragma unroll
    for (int i = 0; i < SV; ++i) {
      int ci = c * KCB + (r < KCB ? r : KCB - 1);
      ci = ci < a.CIN ? ci : a.CIN - 1;
      int t = tb + 4 * v;
      t = t < 0 ? 0 : (t > a.ldx - 4 ? a.ldx - 4 : t);
      sv[i] = *reinterpret_cast<const f32x4*>(xb + (size_t)ci * a.ldx + t);
      v += dv;
      r += dr;
      if (v >= NV) { v -= NV; ++r; }
    }
  };
  auto stage_store = [&](float* buf, int c) {
    if constexpr (DMA) return;
    int r = r0, v = v0;
#pragma unroll
    for (int i = 0; i < SV; ++i) {
      if (r < KCB) {
        const int t = tb + 4 * v;
        const bool rowok = (c * KCB + r) < a.CIN;
        f32x4 val = sv[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = rowok && (t + e) >= 0 && (t + e) < len;
          val[e] = ok ? lrelu(val[e], slope) : 0.f;
        }
        if constexpr (STRIDE == 2) {
          // even / odd samples go to separate half-rows so the stride-2 fragment reads of the
          // strided conv become stride-1 (conflict-free) LDS reads
          float* rowp = buf + r * XW + 2 * v;
          *reinterpret_cast<float2*>(rowp) = make_float2(val[0], val[2]);
          *reinterpret_cast<float2*>(rowp + (XW >> 1)) = make_float2(val[1], val[3]);
        } else {
          *reinterpret_cast<f32x4*>(buf + r * XW + 4 * v) = val;
        }
      }
      v += dv;
      r += dr;
      if (v >= NV) { v -= NV; ++r; }
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  // A fragments: 8 k-steps (16 channels) of one tap = two float4 per lane: [q][half][lane]
  f32x4 av[MI][2], avn[MI][2];
  // scalar-base loads (common.h: wave_rsrc): this wave's MI subtiles are one contiguous slab [mi][block][half][lane]
  const int slab = __builtin_amdgcn_readfirstlane(grp * a.nsub_group + ms0);
  const __amdgpu_buffer_rsrc_t wrs = wave_rsrc(reinterpret_cast<const f32x4*>(a.wpack) + (size_t)slab * nq * 128, (unsigned)(MI * nq) * 2048u);
  const unsigned lane16 = lane * 16u;
  auto a_load = [&](int mi, int bl, int hf) __attribute__((always_inline)) {
    return rsrc_load16(wrs, lane16, (unsigned)((mi * nq + bl) * 2 + hf) * 1024u);
  };
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    av[mi][0] = a_load(mi, 0, 0);
    av[mi][1] = a_load(mi, 0, 1);
  }

  if constexpr (DMA) {
    stage_dma(xs, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  } else {
    stage_load(0);
    stage_store(xs, 0);
  }
  __syncthreads();

#define DISSC_MFMA_STEP(KS_, BV)                                                             \
  _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                          \
  _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                          \
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][(KS_) >> 2][(KS_) & 3], BV[ni], acc[mi][ni], 0, 0, 0);

  constexpr bool PLANES = (STRIDE == 2) && !DMA;  // even / odd sample planes (register staging only)
  const int boff = PLANES ? h * XW + wn * (32 * NI) + l31 : h * XW + sh + (wn * (32 * NI) + l31) * STRIDE;
  constexpr int CS = PLANES ? 1 : STRIDE;  // column step of the fragment reads
  const int XH = XW >> 1;
  const int XW2 = 2 * XW;
  int q = 0;
  const int nblk = (a.nchunk + CPB - 1) / CPB;
  for (int cb = 0; cb < nblk; ++cb) {
    const float* blk = xs + (cb & 1) * (KCB * XW) + boff;
    const bool more = cb + 1 < nblk;
#pragma unroll 1
    for (int sc = 0; sc < CPB; ++sc) {
      if (cb * CPB + sc >= a.nchunk) break;
      {
      const float* bch = blk + sc * (KC * XW);
        // tap j of a stride-2 conv reads plane (sh+j)&1 at column offset (sh+j)>>1
        const float* bj = PLANES ? bch + (sh & 1) * XH + (sh >> 1) : bch;
        float b0[NI], bk[7][NI], b0n[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b0[ni] = bj[ni * 32 * CS];
        for (int j = 0; j < a.KS; ++j, ++q) {
          const int qn = (q + 1 < nq) ? q + 1 : q;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            avn[mi][0] = a_load(mi, qn, 0);
            avn[mi][1] = a_load(mi, qn, 1);
          }
          if (j == 0 && sc == 0 && more) {  // in flight behind this block's MFMAs
            if constexpr (DMA) stage_dma(xs + ((cb + 1) & 1) * (KCB * XW), cb + 1);
            else stage_load(cb + 1);
          }
          __builtin_amdgcn_sched_barrier(0);      // keep the prefetches ahead of the MFMAs
#pragma unroll
          for (int s = 0; s < 7; ++s)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) bk[s][ni] = bj[(s + 1) * XW2 + ni * 32 * CS];
          DISSC_MFMA_STEP(0, b0)
          if constexpr (PLANES) {
            const int tj = sh + j + 1;
            bj = bch + (tj & 1) * XH + (tj >> 1);
          } else {
            bj += a.dil;
          }
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) b0n[ni] = bj[ni * 32 * CS];  // next tap's first k-step
          DISSC_MFMA_STEP(1, bk[0])
          DISSC_MFMA_STEP(2, bk[1])
          DISSC_MFMA_STEP(3, bk[2])
          DISSC_MFMA_STEP(4, bk[3])
          DISSC_MFMA_STEP(5, bk[4])
          DISSC_MFMA_STEP(6, bk[5])
          DISSC_MFMA_STEP(7, bk[6])
          __builtin_amdgcn_sched_barrier(0);  // the register rotation below must not creep upwards
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) b0[ni] = b0n[ni];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            av[mi][0] = avn[mi][0];
            av[mi][1] = avn[mi][1];
          }
        }
    }
      }
    if constexpr (DMA) __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the next block's window has landed
    else if (more) stage_store(xs + ((cb + 1) & 1) * (KCB * XW), cb + 1);
    __syncthreads();
  }
#undef DISSC_MFMA_STEP

  conv_epilogue32<MI, NI>(a, acc, xs, b, t0, olen, grp, ms0, wn);
}

// ---------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------
// Tile shapes (32-unit wave tiles): BM = 32*MI*WM, BN = 32*NI*WN.
struct TileCfg32 { int MI, NI, WM, WN; };
static const TileCfg32 kCfgs32[] = {
    {2, 2, 4, 1},  // 0: 256 x 64
    {2, 2, 2, 2},  // 1: 128 x 128
    {1, 2, 2, 2},  // 2:  64 x 128
    {1, 2, 1, 4},  // 3:  32 x 256
    {1, 4, 1, 4},  // 4:  32 x 512
    {2, 2, 1, 4},  // 5:  64 x 256
    {2, 2, 4, 2},  // 6: 256 x 128 (8 waves)
    {2, 2, 2, 4},  // 7: 128 x 256 (8 waves)
    {0, 0, 0, 0},  // 8, 9: unused
    {0, 0, 0, 0},
    {1, 1, 2, 2},  // 10:  64 x 64   (small grids only, see conv32_pick_cfg)
    {1, 1, 1, 4},  // 11:  32 x 128  (likewise)
};
// option "lin_tile" (Options::lin_tile, default 2): "lin_tile" option: channels per barrier / 16 of the 1x1 (linear) convs (2 or 4)
// option "cpb2" (Options::cpb2, default 0): "cpb2" option: kernels with KS <= this stage 32 channels per barrier
// option "lin_dma" (Options::lin_dma, default 1): "lin_dma" option: 1x1 convs stage their window with global_load_lds (1: 64, 2: 32 channels per barrier)
// option "conv_pad_lds" (Options::conv_pad_lds, default 0): "conv_pad_lds" option (diagnostics): extra LDS bytes per workgroup
// option "c64_wide" (Options::c64_wide, default 1): "c64_wide" option: 64 x 256 tile (64 x 64 wave tiles) for the DMA-staged second convs of the C = 64 stage
// option "conv2_dma" (Options::conv2_dma, default 1): "conv2_dma" option: stride-2 valid convs stage their window with global_load_lds
// option "xcd_order" (Options::xcd_order, default 11): XCD-aware workgroup order (bit 0: 1x1 convs, 1: stride-2 convs, 2: the rest,
//   3: the fused attention, hubert.hip).
//   Encoder, 32 x 10 s (profiles/r05/xcd_order.md): fabric traffic (PMC) 39.60 -> 31.25 GB per forward (conv1 7.98 -> 5.41,
//   fc1 1.217 -> 0.746, qkv 0.752 -> 0.569), time unchanged (27.66 vs 27.66 ms); the generator's few instances: no change
//   in either (bit 2 stays off).
// option "xcd_mg" (Options::xcd_mg, default 0 = automatic): M tiles per sweep of that order
// option "mfast" (Options::mfast, default 0): measured neutral on HuBERT linears (weights are L2/MALL resident either way)

void conv32_set_cfg(int bm_class, int cfg) {
  if (bm_class >= 0 && bm_class < 4 && cfg >= 0 && cfg < 8) g_defaults.cfg32_for_bm[bm_class] = cfg;
}

static int bm32_of(int M) { return M >= 256 ? 256 : (M >= 128 ? 128 : (M >= 64 ? 64 : 32)); }

int conv32_cfg(int M) {
  const int bm = bm32_of(M);
  int cls = 0;
  while ((32 << cls) < bm) ++cls;
  int cfg = opts().cfg32_for_bm[cls];
  if (32 * kCfgs32[cfg].MI * kCfgs32[cfg].WM > bm) cfg = 3 - cls;
  if (cfg == 4 && cls != 0) cfg = 3 - cls;
  return cfg;
}

int conv32_tile_bn(int M) {
  const TileCfg32& c = kCfgs32[conv32_cfg(M)];
  return 32 * c.NI * c.WN;
}

int conv32_cfg_bn(int cfg) { return 32 * kCfgs32[cfg].NI * kCfgs32[cfg].WN; }

// Tile shape for one launch.  The defaults are tuned for grids that fill the chip several times;
// a short or single utterance (the reference's one-at-a-time mode) gives a 256 x 64 tile only a
// few dozen workgroups on 256 CUs, each walking the whole K loop.  Below one workgroup per CU the
// launch steps down to smaller tiles.  Every output element still sees the same sequence of
// 32x32x2 MFMAs over (chunk, tap, k), so the result is bit-identical for every tile shape and an
// utterance's samples stay independent of the batch it runs in.
// option "small_grid" (Options::small_grid, default 1): "small_grid" option: workgroups per CU below which a launch steps down (0 = never)
int conv32_pick_cfg(int M, int B, int Lmax_out) {
  const int base = conv32_cfg(M);
  if (!opts().small_grid) return base;
  static const int kSteps[4][3] = {{3, 11, -1}, {2, 10, -1}, {1, 2, 10}, {0, 2, 10}};  // by BM class
  int cls = 0;
  while ((32 << cls) < bm32_of(M)) ++cls;
  if (base != kSteps[cls][0]) return base;  // a tuning override is in force: leave it alone
  int pick = base;
  for (int i = 0; i < 3 && kSteps[cls][i] >= 0; ++i) {
    pick = kSteps[cls][i];
    const TileCfg32& c = kCfgs32[pick];
    const int bm = 32 * c.MI * c.WM, bn = 32 * c.NI * c.WN;
    const long long nwg = (long long)((Lmax_out + bn - 1) / bn) * ((M + bm - 1) / bm) * B;
    if (nwg >= 256LL * opts().small_grid) break;
  }
  return pick;
}

// w: [Cout][Cin][KS] with Cin per group.  Layout [group][ms32][chunk][tap][half][lane][4]:
// lane l, half hf, component e -> k-step ks = 4*hf + e -> W[32*ms + (l & 31)][16*c + 2*ks + (l >> 5)][tap]
void pack_conv_weights32(const float* w, int Cout, int Cin, int KS, std::vector<float>& packed,
                         int& Mpad, int& nchunk, int groups) {
  const int Mg = Cout / groups;
  const int bm = bm32_of(Mg);
  Mpad = (Mg + bm - 1) / bm * bm;
  nchunk = (Cin + KC - 1) / KC;
  const int nsub = Mpad / 32;
  packed.assign((size_t)groups * nsub * nchunk * KS * 2 * 64 * 4, 0.f);
  for (int gi = 0; gi < groups; ++gi)
    for (int ms = 0; ms < nsub; ++ms)
      for (int c = 0; c < nchunk; ++c)
        for (int j = 0; j < KS; ++j)
          for (int hf = 0; hf < 2; ++hf)
            for (int lane = 0; lane < 64; ++lane)
              for (int e = 0; e < 4; ++e) {
                const int co = ms * 32 + (lane & 31);
                const int ci = c * KC + 2 * (4 * hf + e) + (lane >> 5);
                if (co < Mg && ci < Cin)
                  packed[((((((size_t)gi * nsub + ms) * nchunk + c) * KS + j) * 2 + hf) * 64 + lane) * 4 + e] =
                      w[((size_t)(gi * Mg + co) * Cin + ci) * KS + j];
              }
}

template <int MI, int NI, int WM, int WN, int STRIDE, int SPAN, int CPB = 1, bool DMA = false>
static int launch32_t(ConvArgs a, int B, int Lmax_out, hipStream_t stream) {
  constexpr int BM = 32 * MI * WM, BN = 32 * NI * WN, NW = WM * WN;
  if (DMA && STRIDE == 1 && SPAN > 0) a.XW = (BN + (a.KS - 1) * a.dil + 3 + 3) & ~3;  // this tile's own window
  if (DMA && !(STRIDE == 1 && SPAN > 0)) a.XW = (STRIDE == 2) ? 136 : BN;  // rows packed back to back (stride-1 taps keep conv_xw's XW)
  constexpr int CW = 32 * NI + 4;
  a.mt_per_group = (a.M + BM - 1) / BM;
  // EPI_STORE_ACT writes zero tails up to the end of the output rows: the grid covers ldo, not just the longest utterance
  dim3 grid(((a.epi == EPI_STORE_ACT ? a.ldo : Lmax_out) + BN - 1) / BN, a.mt_per_group * a.groups, B);
  // Many M tiles (HuBERT's 768..3072-row linears): let blockIdx.x walk them, so that the blocks an
  // XCD receives (id % 8) share a few M tiles and their weight slices stay L2-resident.
  a.mfast = (a.mt_per_group * a.groups >= 3 && opts().mfast) ? 1 : 0;
  if (a.mfast) grid = dim3(grid.y, grid.x, grid.z);
  a.ragged_enum = (opts().ragged_enum && (a.lengths || a.lengths_out) && a.epi != EPI_STORE_ACT && B > 1) ? 1 : 0;
  // option "xcd_order" (Options::xcd_order): bit 0: 1x1 convs (linears), bit 1: stride-2 convs, bit 2: every other
  // instance -- launches with >= 2 M tiles only (with one there is nothing to share)
  const int xcd_bit = (STRIDE == 2) ? 2 : (SPAN == 0 ? 1 : 4);
  a.xcd = ((opts().xcd_order & xcd_bit) && a.mt_per_group * a.groups >= 2 && !a.mfast) ? 1 : 0;
  if (a.xcd) {
    // M tiles per sweep: as many 32*MI*WM-row weight slabs as stay L2-resident next to the streamed windows (~3.2 MB of
    // the 4 MB, and a divisor of the M tile count); "xcd_mg" overrides.  Slabs of which fewer than two fit (fc2, K = 3072: 3 MB
    // each) are re-fetched whatever the order: such launches keep all their M tiles in one sweep (every window read once).
    const int mt = a.mt_per_group * a.groups;
    const double slab = (double)BM * a.CIN * a.KS * sizeof(float);
    int mg = (int)(3.2 * 1024 * 1024 / slab);
    if (mg < 2 || mg > mt) mg = mt;
    while (mt % mg) --mg;  // whole sweeps only: a short last sweep would add a partly filled round of workgroups (qkv, 9 M tiles
                           // in sweeps of 4 + 4 + 1: 467 -> 609 us)
    if (opts().xcd_mg > 0) mg = opts().xcd_mg < mt ? opts().xcd_mg : mt;
    const long long tt_pad = ((long long)grid.x * B + 7) / 8 * 8;
    if (tt_pad * mg * ((mt + mg - 1) / mg) > 0x7fffffffLL) {
      a.xcd = 0;  // beyond a 1-D grid (the 2^31-element probes): keep the 3-D grid, B on z
    } else {
      a.xcd_ntile = (int)grid.x;
      a.xcd_nb = B;
      a.xcd_mg = mg;
      a.xcd_span = (int)(tt_pad * mg);
      grid = dim3((unsigned)(tt_pad * mg * ((mt + mg - 1) / mg)), 1, 1);
    }
  }
  size_t lds_f = (size_t)2 * KC * CPB * a.XW;
  if (lds_f < (size_t)NW * 8 * CW) lds_f = (size_t)NW * 8 * CW;
  const size_t lds = lds_f * sizeof(float) + (size_t)opts().conv_pad_lds;  // (+ diagnostics: occupancy experiments)
  static DeviceOnce attr_once;  // per device (common.h)
  DISSC_HIP_CHECK(attr_once.max_lds(reinterpret_cast<const void*>(&conv_mfma32_kernel<MI, NI, WM, WN, STRIDE, SPAN, CPB, DMA>), 160 * 1024));
  hipLaunchKernelGGL((conv_mfma32_kernel<MI, NI, WM, WN, STRIDE, SPAN, CPB, DMA>), grid, dim3(64 * WM * WN), lds,
                     stream, a);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

int launch_conv32(const ConvArgs& a, int B, int Lmax_out, int stride, hipStream_t stream) {
  const int span = (a.KS - 1) * a.dil;
  const int cfg = a.cfg32 >= 0 ? a.cfg32 : conv32_cfg(a.M);
  if (stride == 2 && opts().conv2s128 && conv2s128_supported(a)) return launch_conv2s128(a, B, Lmax_out, stream);
  if (stride == 2 && span <= MAX_TAP_SPAN && a.up == 1) {
    // valid (unpadded) convs on an already-activated input: raw LDS-DMA window, nothing to mask -- every output column
    // below the utterance's output length reads inputs below its input length
    if (cfg == 0 && opts().conv2_dma && a.slope == 1.0f && a.pad_left == 0 && a.groups == 1 && a.KS <= 9 && a.dil == 1 &&
        a.CIN % KC == 0 && a.ldx >= 4 && a.ldx % 4 == 0)
      return launch32_t<2, 2, 4, 1, 2, MAX_TAP_SPAN, 1, true>(a, B, Lmax_out, stream);
    if (cfg == 0) return launch32_t<2, 2, 4, 1, 2, MAX_TAP_SPAN>(a, B, Lmax_out, stream);
    set_error("launch_conv32: stride 2 needs >= 256 output rows (got %d)", a.M);
    return DISSC_EINVAL;
  }
  if (stride != 1) {
    set_error("launch_conv32: stride %d unsupported", stride);
    return DISSC_EINVAL;
  }
  if (span > MAX_TAP_SPAN) {
    if (span <= WIDE_TAP_SPAN && cfg == 3) return launch32_t<1, 2, 1, 4, 1, WIDE_TAP_SPAN>(a, B, Lmax_out, stream);
    set_error("launch_conv32: kernel %d x dilation %d unsupported for %d rows", a.KS, a.dil, a.M);
    return DISSC_EINVAL;
  }
  if (span == 0 && stride == 1 && opts().lin128 && lin128_supported(a) && a.nchunk >= 8) return launch_lin128(a, B, Lmax_out, stream);
  if (span == 0 && bm32_of(a.M) == 256 && a.nchunk >= 8) {  // 1x1 convs: 64 channels per barrier
    // (the tile must stay the default 256 x 64: a.XW was sized for its BN)
    const bool dma = opts().lin_dma && a.slope == 1.0f && a.pad_left == 0 && a.up == 1 && a.groups == 1 &&
                     a.CIN % (2 * KC) == 0 && a.ldx >= 4 && a.ldx % 4 == 0;
    // without staging registers 64 channels per barrier fit 3 waves per SIMD (measured: 28.84 -> 28.67 ms per encode)
    if (dma && opts().lin_dma == 1 && a.CIN % (4 * KC) == 0) return launch32_t<2, 2, 4, 1, 1, 0, 4, true>(a, B, Lmax_out, stream);
    if (dma) return launch32_t<2, 2, 4, 1, 1, 0, 2, true>(a, B, Lmax_out, stream);
    if (opts().lin_tile == 4) return launch32_t<2, 2, 4, 1, 1, 0, 4>(a, B, Lmax_out, stream);  // 64 ch / barrier
    return launch32_t<2, 2, 4, 1, 1, 0, 2>(a, B, Lmax_out, stream);                        // 32 ch / barrier
  }
  if (a.prec == 1) return launch_conv_bf3(a, B, Lmax_out, stream);  // split-bf16 kernels (conv_bf3.hip)
  if (opts().cpb2 && a.KS <= opts().cpb2 && a.nchunk >= 4 && a.up == 1) {  // two chunks per barrier for short kernels
    if (cfg == 0) return launch32_t<2, 2, 4, 1, 1, MAX_TAP_SPAN, 2>(a, B, Lmax_out, stream);
    if (cfg == 1) return launch32_t<2, 2, 2, 2, 1, MAX_TAP_SPAN, 2>(a, B, Lmax_out, stream);
    if (cfg == 2) return launch32_t<1, 2, 2, 2, 1, MAX_TAP_SPAN, 2>(a, B, Lmax_out, stream);
  }
  if (a.dma_in && a.slope == 1.0f && a.up == 1 && a.groups == 1 && a.CIN % KC == 0 && a.KS > 1) {
    // input in the EPI_STORE_ACT layout (generator: the second conv of a residual pair): raw LDS-DMA windows
    if (cfg == 0) return launch32_t<2, 2, 4, 1, 1, MAX_TAP_SPAN, 1, true>(a, B, Lmax_out, stream);
    if (cfg == 1) return launch32_t<2, 2, 2, 2, 1, MAX_TAP_SPAN, 1, true>(a, B, Lmax_out, stream);
    if (cfg == 2 && opts().c64_wide) return launch32_t<2, 2, 1, 4, 1, MAX_TAP_SPAN, 1, true>(a, B, Lmax_out, stream);  // 64 x 256
    if (cfg == 2) return launch32_t<1, 2, 2, 2, 1, MAX_TAP_SPAN, 1, true>(a, B, Lmax_out, stream);
  }
  switch (cfg) {
    case 0: return launch32_t<2, 2, 4, 1, 1, MAX_TAP_SPAN>(a, B, Lmax_out, stream);
    case 1: return launch32_t<2, 2, 2, 2, 1, MAX_TAP_SPAN>(a, B, Lmax_out, stream);
    case 2: return launch32_t<1, 2, 2, 2, 1, MAX_TAP_SPAN>(a, B, Lmax_out, stream);
    case 3: return launch32_t<1, 2, 1, 4, 1, MAX_TAP_SPAN>(a, B, Lmax_out, stream);
    case 4: return launch32_t<1, 4, 1, 4, 1, MAX_TAP_SPAN>(a, B, Lmax_out, stream);
    case 6: return launch32_t<2, 2, 4, 2, 1, MAX_TAP_SPAN>(a, B, Lmax_out, stream);
    case 7: return launch32_t<2, 2, 2, 4, 1, MAX_TAP_SPAN>(a, B, Lmax_out, stream);
    case 10: return launch32_t<1, 1, 2, 2, 1, MAX_TAP_SPAN>(a, B, Lmax_out, stream);
    case 11: return launch32_t<1, 1, 1, 4, 1, MAX_TAP_SPAN>(a, B, Lmax_out, stream);
    default: return launch32_t<2, 2, 1, 4, 1, MAX_TAP_SPAN>(a, B, Lmax_out, stream);
  }
}

}  // namespace dissc
