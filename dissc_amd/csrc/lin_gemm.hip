// lin128_kernel: the fp32 GEMM of HuBERT's linears (1x1 convs, out[b][m][t] = sum_k W[m][k] x[b][k][t]) on 256 x 128 output
// tiles, two 4-wave workgroups per CU = TWO waves per SIMD (round 5 verdict, task 1; the wave count conv_wino8's tap loop
// was measured fastest at, tools/ubench/mfma_mix8.hip: 137-145 TFLOP/s at two waves per SIMD, 122-126 at three, 117 at four).
//
// What differs from conv_mfma32_kernel<2, 2, 4, 1, 1, 0, 4, true> (256 x 64 tiles, four workgroups per CU):
//   * a wave owns 64 rows x 128 columns (MI = 2 x NI = 4 accumulator blocks, 128 registers): one 16-byte A load feeds
//     4 column blocks x 4 k-steps (half the global -> VGPR traffic per MFMA, the costliest operand path: profiles/r05/
//     mfma_dma_ubench.txt), and the activation window crosses the fabric once per 256 x 128 tile instead of once per 256 x 64;
//   * the 32 columns of an MFMA are NOT 32 consecutive frames: column l of block ni is frame 4 l + ni of the wave's 128.  A
//     lane's four B operands of one k-step -- (channel 2 ks + (lane >> 5), frames 4 l .. 4 l + 3) -- are then ONE ds_read_b128
//     of the row-major window the LDS-DMA wrote (8 LDS instructions per 64 MFMAs; the 256 x 64 kernel: 16 ds_read_b32 per 32),
//     and the four accumulators a lane holds for one row are four consecutive frames: the epilogue stores 16 bytes per lane
//     straight from registers (512 contiguous bytes per half-wave and row) -- no transposition through LDS;
//   * every output element still sees the SAME sequence of v_mfma_f32_32x32x2_f32 over k (chunk, k-step; the packed weights
//     are the ones pack_conv_weights32 makes) and the same epilogue arithmetic, so results are bit-identical to the 256 x 64
//     kernel's: the tile shape is a schedule, not an arithmetic.
// Window staging: global_load_lds_dwordx4, KCB channels x 128 frames per barrier, two buffers (KCB = 64: 64 KB per workgroup).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "conv_epilogue32.h"

namespace dissc {

// The epilogue of one wave: rows (ms0 + mi) * 32 + 8 j + 4 h + i (register r = 4 j + i of block mi), four consecutive frames per lane.
// Every load is issued before anything depends on it: the 32 bias values (and the affine pair) as float4s -- rows 8 j + 4 h .. + 3 are
// adjacent in the [Mpad] arrays --, the residual four rows at a time.  (A first version loaded bias[row] inside the row loop: 32
// dependent round trips, 17-38 us per tile against a 164 us main loop -- profiles/r06/lin128_timeline_v1.txt.)
template <int MI, bool ACT, bool RES, bool AFF, bool FULL>
__device__ __forceinline__ void lin128_epilogue(const ConvArgs& a, f32x16 (&acc)[MI][4], int ms0, int h, size_t o0, int nv) {
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    f32x4 bz[4], sc[4], sf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rb = (ms0 + mi) * 32 + 8 * j + 4 * h;  // < Mpad (a multiple of 256 rows here)
      bz[j] = *reinterpret_cast<const f32x4*>(a.bias + rb);
      if constexpr (AFF) {
        sc[j] = *reinterpret_cast<const f32x4*>(a.scale + rb);
        sf[j] = *reinterpret_cast<const f32x4*>(a.shift + rb);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int rb = (ms0 + mi) * 32 + 8 * j + 4 * h;
      const size_t idx0 = o0 + (size_t)rb * a.ldo;
      f32x4 rs[4];
      if constexpr (RES) {
#pragma unroll
        for (int i = 0; i < 4; ++i)  // whole float4s: tcol + 3 < ldo, and lanes that are not stored are not used
          rs[i] = *reinterpret_cast<const f32x4*>(a.res + idx0 + (size_t)i * a.ldo);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * j + i;
        f32x4 v = {acc[mi][0][r], acc[mi][1][r], acc[mi][2][r], acc[mi][3][r]};
        const float b1 = bz[j][i];
        v[0] += b1; v[1] += b1; v[2] += b1; v[3] += b1;
        if constexpr (AFF) {
          const float s1 = sc[j][i], f1 = sf[j][i];
          v[0] = v[0] * s1 + f1; v[1] = v[1] * s1 + f1; v[2] = v[2] * s1 + f1; v[3] = v[3] * s1 + f1;
        }
        if constexpr (ACT) {
          v[0] = gelu_exact(v[0]); v[1] = gelu_exact(v[1]); v[2] = gelu_exact(v[2]); v[3] = gelu_exact(v[3]);
        }
        if constexpr (RES) {
          v[0] += rs[i][0]; v[1] += rs[i][1]; v[2] += rs[i][2]; v[3] += rs[i][3];
        }
        float* op = a.out + idx0 + (size_t)i * a.ldo;  // every row exists: M is a multiple of the 256-row tile (lin128_supported)
        if (FULL || nv >= 4) {
          *reinterpret_cast<f32x4*>(op) = v;
        } else {  // the last float4 of an utterance: 1 .. 3 frames
          op[0] = v[0];
          if (nv > 1) op[1] = v[1];
          if (nv > 2) op[2] = v[2];
        }
      }
    }
  }
}

// DBG (knock-outs for the gate record, option "lin128_dbg"; results are garbage): bit 0: no A loads in the loop, 1: no window DMA,
// 2: no B reads, 3: no epilogue, 4: no wait + barrier per stage
template <int MI, int KCB, int WGPC, int DBG = 0>
__global__ void __launch_bounds__(256, WGPC) lin128_kernel(const ConvArgs a) {
  constexpr int BN = 128, NT = 256, NI = 4;
  constexpr int CH = KCB / KC;        // 16-channel chunks per barrier
  constexpr int ND = KCB * 32 / NT;   // DMA instructions per thread and stage (one = 16 bytes per lane = two 128-frame rows per wave)
  extern __shared__ __attribute__((aligned(16))) float xs[];  // 2 x [KCB][128]

  int b = blockIdx.z, bx = blockIdx.x, by = blockIdx.y;
  int ntile_g = gridDim.x, nb_g = gridDim.z;
  if (a.stagger) {
    // The dispatcher deals workgroup ids round-robin over the 8 XCDs and, inside an XCD, breadth-first over its 32 CUs: ids 0 .. 255
    // take the first slot of every CU, ids 256 .. 511 the second.  Those start late, so that the two workgroups of a CU are
    // never in their epilogues (an HBM write burst with the matrix pipe idle) or prologues at the same time; every later
    // workgroup starts when a slot frees up and inherits the offset.
    const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (id >= 256u && id < 512u)
      for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(16);  // 16 x 64 cycles
  }
  if (a.xcd) {  // XCD order, as in conv_mfma32_kernel (conv_mfma32.hip has the description)
    const int mt = a.mt_per_group, mg = a.xcd_mg;
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
  if (a.ragged_enum) {  // only the (time tile, utterance) pairs that exist, as in conv_mfma32_kernel
    const int lin = b * ntile_g + bx;
    const int lane_ = threadIdx.x & 63;
    int base = 0;
    b = -1;
    for (int b0 = 0; b0 < nb_g; b0 += 64) {
      int l = 0;
      if (b0 + lane_ < nb_g)
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
  unsigned long long* tl = nullptr;  // DBG bit 5: 100 MHz wall-clock stamps of wave 0 (start, loop start, loop end, stores issued) + placement
  if constexpr ((DBG & 32) != 0) {
    const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    tl = reinterpret_cast<unsigned long long*>(a.acc) + (size_t)id * 8;
    if (threadIdx.x == 0) {
      tl[0] = __builtin_amdgcn_s_memrealtime();
      tl[4] = ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
      tl[1] = tl[2] = tl[3] = 0;
    }
  }
  if (a.stagger >= 0) __builtin_amdgcn_s_setprio(3);
  const int len = (a.lengths ? a.lengths[b] * a.len_mul : a.len_default);
  const int olen = a.lengths_out ? a.lengths_out[b] : (a.olen_default >= 0 ? a.olen_default : len);
  const int t0 = bx * BN;
  if (t0 >= olen) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int ms0 = by * (MI * 4) + wave * MI;  // this wave's first 32-row subtile
  const int nq = a.nchunk;
  const float* xb = a.x + (size_t)b * a.x_bstride;

  // DMA: slot e = tid + 256 i -> channel row (tid >> 5) + 8 i of the stage, frames t0 + 4 (tid & 31) .. + 3.  Frames beyond the row
  // are clamped into it (their columns are >= olen and never stored); nothing is masked: a 1x1 conv's column reads its own column.
  int tcl = t0 + 4 * (tid & 31);
  tcl = tcl > a.ldx - 4 ? a.ldx - 4 : tcl;
  const float* src = xb + (size_t)(tid >> 5) * a.ldx + tcl;
  const size_t row8 = (size_t)8 * a.ldx;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);  // scalar: the DMA's LDS address (M0) is made on the SALU
  auto stage_dma = [&](float* buf, int s) __attribute__((always_inline)) {
    const float* p = src + (size_t)s * KCB * a.ldx;
#pragma unroll
    for (int i = 0; i < ND; ++i)
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(p + i * row8),
                                       (void __attribute__((address_space(3)))*)(buf + (i * NT + wave_s * 64) * 4), 16, 0, 0);
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  // A fragments of one 16-channel chunk: two float4 per lane and 32-row subtile ([chunk][half][lane], pack_conv_weights32); two
  // register sets take turns (chunk q in set q & 1; CH is even, so the roles are the same at every stage's start)
  static_assert(CH % 2 == 0, "two A register sets take turns over the chunks of a stage");
  const f32x4* wp[MI];
  f32x4 av[2][MI][2];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    wp[mi] = reinterpret_cast<const f32x4*>(a.wpack) + (size_t)(ms0 + mi) * nq * 128 + lane;
    av[0][mi][0] = wp[mi][0];
    av[0][mi][1] = wp[mi][64];
  }
  stage_dma(xs, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __syncthreads();

  if constexpr ((DBG & 32) != 0) {
    if (threadIdx.x == 0) tl[1] = __builtin_amdgcn_s_memrealtime();
  }
  __builtin_amdgcn_s_setprio(0);
  const int boff = h * BN + 4 * l31;
  const int nstage = a.CIN / KCB;
  f32x4 bv[8];
  // One stage = CH chunks, fully unrolled.  Order of issue inside a chunk: the NEXT chunk's A fragments (global -> the other register
  // set), in the stage's first chunk then the next stage's window (LDS-DMA; issued AFTER the A loads: vmcnt retires in order, so
  // the wait for those A loads one chunk later leaves the DMA in flight -- it has two chunks = 128 MFMAs to land), then per k-step
  // 8 MFMAs followed by the ds_read_b128 that refills this k-step's B registers with the next chunk's values (one chunk ahead,
  // no second register set: the MFMAs that read them have been issued).  The last chunk of a stage refills nothing -- the next
  // stage's buffer is valid after the barrier only -- and the first chunk's eight reads follow the barrier.
  auto stage = [&](int s, auto last_tag) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_tag)::value;
    const float* blk = xs + (s & 1) * (KCB * BN) + boff;
    if (!(DBG & 4) || s == 0) {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) bv[ks] = *reinterpret_cast<const f32x4*>(blk + ks * (2 * BN));
    }
#pragma unroll
    for (int sc = 0; sc < CH; ++sc) {
      const int q = s * CH + sc;
      const int cur = sc & 1, nxt = cur ^ 1;
      if (!(LAST && sc == CH - 1) && !(DBG & 1)) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          av[nxt][mi][0] = wp[mi][(size_t)(q + 1) * 128];
          av[nxt][mi][1] = wp[mi][(size_t)(q + 1) * 128 + 64];
        }
      }
      if (!LAST && sc == 0 && !(DBG & 2)) stage_dma(xs + ((s + 1) & 1) * (KCB * BN), s + 1);
      __builtin_amdgcn_sched_barrier(0);
      const float* bnx = blk + (sc + 1) * (KC * BN);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(DBG & 1) ? 0 : cur][mi][ks >> 2][ks & 3], bv[ks][ni], acc[mi][ni], 0, 0, 0);
        if (sc + 1 < CH && !(DBG & 4)) bv[ks] = *reinterpret_cast<const f32x4*>(bnx + ks * (2 * BN));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!LAST && !(DBG & 16)) {
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the next stage's window (and the next chunk's A fragments) have landed
      __syncthreads();
    }
  };
#pragma unroll 1
  for (int s = 0; s + 1 < nstage; ++s) stage(s, std::false_type{});
  stage(nstage - 1, std::true_type{});
  if constexpr ((DBG & 32) != 0) {
    if (threadIdx.x == 0) tl[2] = __builtin_amdgcn_s_memrealtime();
  }

  // Epilogue straight from registers: C/D layout of 32x32x2: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); block
  // ni's column l31 is frame 4 l31 + ni, so {acc[mi][0..3][r]} are four consecutive frames of one row.  Arithmetic and order of
  // operations as conv_epilogue32 (bias, affine, GELU, residual).
  if (DBG & 8) {  // keep every accumulator alive without the epilogue's work
    float sum = 0.f;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) sum += acc[mi][ni][e];
    if (sum == 12345.678f) a.out[0] = sum;
    return;
  }
  // The epilogue (and the prologue) at raised priority: next to a wave that issues MFMAs back to back, this wave's ~500 VALU / store
  // instructions otherwise get an issue slot every 2-3 MFMAs (17-35 us instead of 3.5 alone, profiles/r06/lin128_timeline*.txt)
  if (a.stagger >= 0) __builtin_amdgcn_s_setprio(3);
  const int tcol = t0 + 4 * l31;
  if (tcol < olen) {
    const int fl = (a.act == 1 ? 1 : 0) | (a.epi == EPI_RES ? 2 : 0) | (a.scale ? 4 : 0);
    const size_t o0 = (size_t)b * a.o_bstride + tcol;
    const int nv = olen - tcol;
    // uniform choices: one straight-line body per combination (no per-row branches, no per-row dependent loads); FULL: no lane of
    // this wave holds the partial last float4 of an utterance (three tiles in four at T = 499)
    const bool full = __builtin_amdgcn_ballot_w64(nv < 4) == 0;
#define DISSC_LIN128_EPI(ACT, RES, AFF)                                                     \
  if (full) lin128_epilogue<MI, ACT, RES, AFF, true>(a, acc, ms0, h, o0, nv);                \
  else lin128_epilogue<MI, ACT, RES, AFF, false>(a, acc, ms0, h, o0, nv);                    \
  break;
    switch (fl) {
      case 0: DISSC_LIN128_EPI(false, false, false)
      case 1: DISSC_LIN128_EPI(true, false, false)
      case 2: DISSC_LIN128_EPI(false, true, false)
      case 3: DISSC_LIN128_EPI(true, true, false)
      case 4: DISSC_LIN128_EPI(false, false, true)
      case 5: DISSC_LIN128_EPI(true, false, true)
      case 6: DISSC_LIN128_EPI(false, true, true)
      default: DISSC_LIN128_EPI(true, true, true)
    }
#undef DISSC_LIN128_EPI
  }
  if constexpr ((DBG & 32) != 0) {
    if (threadIdx.x == 0) {
      tl[3] = __builtin_amdgcn_s_memrealtime();  // stores issued
      __builtin_amdgcn_s_waitcnt(0x0F70);
      tl[5] = __builtin_amdgcn_s_memrealtime();  // ... and acknowledged
    }
  }
}

// option "lin128" (Options::lin128, default 1): HuBERT's linears on lin128_kernel (0: conv_mfma32_kernel's 256 x 64 instances)
bool lin128_supported(const ConvArgs& a) {
  return a.KS == 1 && a.groups == 1 && a.up == 1 && a.slope == 1.0f && a.pad_left == 0 && a.prec == 0 && a.m32 == 1 &&
         (a.epi == EPI_STORE || a.epi == EPI_RES) && a.M >= 256 && a.M % 256 == 0 && a.CIN % 64 == 0 && a.ldx >= 4 && a.ldx % 4 == 0 &&
         a.ldo % 4 == 0;
}

template <int MI, int KCB, int WGPC, int DBG = 0>
static int launch_lin128_t(ConvArgs a, int B, int Lmax_out, hipStream_t stream) {
  constexpr int BM = 128 * MI, BN = 128;
  a.mt_per_group = (a.M + BM - 1) / BM;
  dim3 grid((Lmax_out + BN - 1) / BN, a.mt_per_group, B);
  a.mfast = 0;
  a.stagger = opts().lin128_stagger;
  a.ragged_enum = (opts().ragged_enum && (a.lengths || a.lengths_out) && B > 1) ? 1 : 0;
  const int mt = a.mt_per_group;
  const long long tt_pad = ((long long)grid.x * B + 7) / 8 * 8;
  a.xcd = ((opts().xcd_order & 1) && mt >= 2) ? 1 : 0;
  if (a.xcd) {
    const double slab = (double)BM * a.CIN * sizeof(float);
    int mg = (int)(3.2 * 1024 * 1024 / slab);
    if (mg < 2 || mg > mt) mg = mt;
    while (mt % mg) --mg;
    if (opts().xcd_mg > 0) mg = opts().xcd_mg < mt ? opts().xcd_mg : mt;
    if (tt_pad * mg * ((mt + mg - 1) / mg) > 0x7fffffffLL) {
      a.xcd = 0;  // beyond a 1-D grid: keep the 3-D one
    } else {
      a.xcd_ntile = (int)grid.x;
      a.xcd_nb = B;
      a.xcd_mg = mg;
      a.xcd_span = (int)(tt_pad * mg);
      grid = dim3((unsigned)(tt_pad * mg * ((mt + mg - 1) / mg)), 1, 1);
    }
  }
  size_t lds = (size_t)2 * KCB * BN * sizeof(float);
  if (opts().lin128_dbg == 64 || getenv("DISSC_LIN128_ONE")) lds = 100 * 1024;  // diagnostics: ONE workgroup per CU (one wave per SIMD)
  static DeviceOnce attr_once;
  DISSC_HIP_CHECK(attr_once.max_lds(reinterpret_cast<const void*>(&lin128_kernel<MI, KCB, WGPC, DBG>), 160 * 1024));
  hipLaunchKernelGGL((lin128_kernel<MI, KCB, WGPC, DBG>), grid, dim3(256), lds, stream, a);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

int launch_lin128(const ConvArgs& a, int B, int Lmax_out, hipStream_t stream) {
  switch (opts().lin128_dbg) {  // knock-outs of the KCB = 32 form
    case 0: case 64: break;
    case 1: return launch_lin128_t<2, 32, 2, 1>(a, B, Lmax_out, stream);
    case 2: return launch_lin128_t<2, 32, 2, 2>(a, B, Lmax_out, stream);
    case 4: return launch_lin128_t<2, 32, 2, 4>(a, B, Lmax_out, stream);
    case 8: return launch_lin128_t<2, 32, 2, 8>(a, B, Lmax_out, stream);
    case 16: return launch_lin128_t<2, 32, 2, 16>(a, B, Lmax_out, stream);
    case 18: return launch_lin128_t<2, 32, 2, 18>(a, B, Lmax_out, stream);
    case 23: return launch_lin128_t<2, 32, 2, 23>(a, B, Lmax_out, stream);
    case 31: return launch_lin128_t<2, 32, 2, 31>(a, B, Lmax_out, stream);
    case 32: return launch_lin128_t<2, 32, 2, 32>(a, B, Lmax_out, stream);
    default: set_error("lin128_dbg: no instance %d", opts().lin128_dbg); return DISSC_EINVAL;
  }
  // Tile shape per launch (option "lin128": 1 = this policy; 2 / 3 / 5 force a shape, for the gate records).  All shapes give the same
  // bits.  A CU works its tiles off two or three at a time; what decides is how evenly the tile count divides over 256 CUs
  // (profiles/r06/lin128_gate_v7.txt, B = 32 x T = 499, us, old kernel / 256 x 128 / 128 x 128 x3 per CU / 128 x 128 x2 per CU, 64
  // channels per barrier: fc1 619 / 581 / 628 / 594, fc2 629 / 748 / 592 / 579, qkv 470 / 491 / 474 / 448, out_proj 182 / 198 / 164 /
  // 192, proj 116 / 134 / 114 / 134):
  //   256 x 128 (64-row waves: half the A traffic per MFMA) when that leaves no CU with more row-tile work than 128-row tiles would;
  //   else 128 x 128: three workgroups per CU when a CU gets at most three tiles of a short K loop (one round, nobody alone),
  //   else two per CU with 64 channels per barrier.
  int mode = opts().lin128;
  if (mode == 1) {
    const long long ct = (long long)((Lmax_out + 127) / 128) * B;  // column tiles (an upper bound for ragged batches)
    const long long m1 = (ct * ((a.M + 127) / 128) + 255) / 256, m2 = (ct * ((a.M + 255) / 256) + 255) / 256;
    mode = (2 * m2 <= m1) ? 2 : ((m1 <= 3 && a.CIN < 2048) ? 3 : 5);
  }
  switch (mode) {
    case 2: return launch_lin128_t<2, 32, 2>(a, B, Lmax_out, stream);  // 256 x 128 tiles, two workgroups per CU
    case 3: return launch_lin128_t<1, 32, 3>(a, B, Lmax_out, stream);  // 128 x 128 tiles, three workgroups per CU
    default: return launch_lin128_t<1, 64, 2>(a, B, Lmax_out, stream); // 128 x 128 tiles, two per CU, 64 channels per barrier
  }
}

}  // namespace dissc
