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

// ---- hand-counted VMEM waits -------------------------------------------------------------------------------------------------
// hipcc (ROCm 7.2) waits vmcnt(0) at the next use of ANY global load's result while an LDS-DMA is in flight, and vmcnt retires in
// order: with compiler-issued loads a window DMA is drained by the first A fragment used after it -- one or two groups of MFMAs
// after it was issued -- and every A fragment behind it with it (knock-outs, profiles/r06/conv2s128_ko.txt: A loads 7.3 %, DMA
// 4.5 % of conv1).  These helpers issue the loads in inline asm, where the compiler keeps no score, and wait with exact counts:
//   vm_load16   one global_load_dwordx4 (the result is NOT valid until a vm_wait that covers it)
//   vm_dma16    one global_load_lds_dwordx4 (M0 = wave-uniform LDS byte address; lane l lands at + 16 l)
//   vm_wait<N>  s_waitcnt vmcnt(N), tied to the registers it makes valid so that their readers stay behind it
// (addresses: a wave-uniform 64-bit base in SGPRs + the lane's 32-bit byte offset -- the saddr form; the compiler's own loads carry
// 64-bit per-lane addresses made by v_lshl_add_u64, VALU work that is paid in matrix-pipe time)
__device__ __forceinline__ f32x4 vm_load16(const void* sbase, unsigned voff) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase));
  return v;
}
__device__ __forceinline__ void vm_dma16(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait(f32x4& r0, f32x4& r1) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(r0), "+v"(r1) : "n"(N));
}
template <int N>
__device__ __forceinline__ void vm_wait(f32x4& r0) {
  asm volatile("s_waitcnt vmcnt(%1)" : "+v"(r0) : "n"(N));
}

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

// Stores of one wave: uniform choice of one straight-line epilogue body per (activation, residual, affine) combination -- no per-row
// branches, no per-row dependent loads; FULL: no lane of this wave holds the partial last float4 of an utterance.
template <int MI>
__device__ __forceinline__ void lin128_store(const ConvArgs& a, f32x16 (&acc)[MI][4], int ms0, int h, int b, int tcol, int olen) {
  if (tcol >= olen) return;
  const int fl = (a.act == 1 ? 1 : 0) | (a.epi == EPI_RES ? 2 : 0);  // (layers with a per-row affine stay on conv_mfma32_kernel)
  const size_t o0 = (size_t)b * a.o_bstride + tcol;
  const int nv = olen - tcol;
  const bool full = __builtin_amdgcn_ballot_w64(nv < 4) == 0;
#define DISSC_LIN128_EPI(ACT, RES, AFF)                                                     \
  if (full) lin128_epilogue<MI, ACT, RES, AFF, true>(a, acc, ms0, h, o0, nv);                \
  else lin128_epilogue<MI, ACT, RES, AFF, false>(a, acc, ms0, h, o0, nv);                    \
  break;
  switch (fl) {
    case 0: DISSC_LIN128_EPI(false, false, false)
    case 1: DISSC_LIN128_EPI(true, false, false)
    case 2: DISSC_LIN128_EPI(false, true, false)
    default: DISSC_LIN128_EPI(true, true, false)
  }
#undef DISSC_LIN128_EPI
}

// Which tile is this workgroup's?  XCD order and ragged enumeration as in conv_mfma32_kernel (conv_mfma32.hip has the descriptions).
// false: nothing to do.
__device__ __forceinline__ bool tile128_of(const ConvArgs& a, int& b, int& bx, int& by) {
  constexpr int BN = 128;
  b = blockIdx.z, bx = blockIdx.x, by = blockIdx.y;
  int ntile_g = gridDim.x, nb_g = gridDim.z;
  if (a.xcd) {
    const int mt = a.mt_per_group, mg = a.xcd_mg;
    const int sweep = blockIdx.x / a.xcd_span, r = blockIdx.x - sweep * a.xcd_span;
    const int s = r >> 3, sq = s / mg;
    const int tt = (r & 7) + 8 * sq;
    by = sweep * mg + (s - sq * mg);
    ntile_g = a.xcd_ntile;
    nb_g = a.xcd_nb;
    if (tt >= ntile_g * nb_g || by >= mt) return false;
    b = tt / ntile_g;
    bx = tt - b * ntile_g;
  }
  if (a.ragged_enum) {
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
    if (b < 0) return false;
  }
  // wave-uniform by construction; say so (the ragged search's __shfl results look divergent to the compiler, and with them every
  // address derived from the tile: 64-bit per-lane address math instead of scalar bases)
  b = __builtin_amdgcn_readfirstlane(b);
  bx = __builtin_amdgcn_readfirstlane(bx);
  by = __builtin_amdgcn_readfirstlane(by);
  return true;
}

// DBG (diagnostics, option "kernel_dbg"): bit 3: no epilogue, bit 5: timeline stamps into a.acc (the knock-outs of the first,
// compiler-scheduled form -- A loads 5.7 %, epilogue 8 %, barrier 3 % of fc1 -- are on record in profiles/r06/lin128_gate_v3.txt)
template <int MI, int KCB, int WGPC, int DBG = 0>
__global__ void __launch_bounds__(256, WGPC) lin128_kernel(const ConvArgs a) {
  constexpr int BN = 128, NT = 256, NI = 4;
  constexpr int CH = KCB / KC;        // 16-channel chunks per barrier
  constexpr int ND = KCB * 32 / NT;   // DMA instructions per thread and stage (one = 16 bytes per lane = two 128-frame rows per wave)
  extern __shared__ __attribute__((aligned(16))) float xs[];  // 2 x [KCB][128]

  int b, bx, by;
  if (!tile128_of(a, b, bx, by)) return;
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
  __builtin_amdgcn_s_setprio(3);
  const int len = (a.lengths ? a.lengths[b] * a.len_mul : a.len_default);
  const int olen = a.lengths_out ? a.lengths_out[b] : (a.olen_default >= 0 ? a.olen_default : len);
  const int t0 = bx * BN;
  if (t0 >= olen) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int ms0 = by * (MI * 4) + wave * MI;  // this wave's first 32-row subtile
  const int nq = a.nchunk;
  const float* xb = a.x + (size_t)b * a.x_bstride;

  // DMA: slot e = tid + 256 i -> channel row (tid >> 5) + 8 i of the stage, frames t0 + 4 (tid & 31) .. + 3: one wave instruction moves
  // two rows, rows 2 w + 8 i (lanes 0 .. 31) and + 1 (lanes 32 .. 63).  Frames beyond the row are clamped into it (their columns are
  // >= olen and never stored); nothing is masked: a 1x1 conv's column reads its own column.  Every address is a wave-uniform base in
  // SGPRs + the lane's 32-bit byte offset (vm_load16 / vm_dma16 above).
  int tcl = t0 + 4 * l31;
  tcl = tcl > a.ldx - 4 ? a.ldx - 4 : tcl;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const float* xw = xb + (size_t)(2 * wave_s) * a.ldx;
  const unsigned vx_off = ((unsigned)h * (unsigned)a.ldx + (unsigned)tcl) * 4u;
  const size_t row8 = (size_t)8 * a.ldx;
  const unsigned xs_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)xs);
  auto stage_dma = [&](int bufi, int s) __attribute__((always_inline)) {
    const float* p = xw + (size_t)s * KCB * a.ldx;
    const unsigned baddr = xs_addr + bufi * (KCB * BN * 4);
#pragma unroll
    for (int i = 0; i < ND; ++i) vm_dma16(p + i * row8, vx_off, baddr + (i * NT + wave_s * 64) * 16);
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
  constexpr int LA = 2 * MI;  // A loads per chunk and wave
  const unsigned lane16 = lane * 16u;
  const f32x4* wp[MI];
  f32x4 av[2][MI][2];
  auto wait_a = [&](int set, auto cnt_tag) __attribute__((always_inline)) {
    constexpr int CNT = decltype(cnt_tag)::value;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) vm_wait<CNT>(av[set][mi][0], av[set][mi][1]);
  };
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    wp[mi] = reinterpret_cast<const f32x4*>(a.wpack) + (size_t)(by * (MI * 4) + wave_s * MI + mi) * nq * 128;
    av[0][mi][0] = vm_load16(wp[mi], lane16);
    av[0][mi][1] = vm_load16(wp[mi] + 64, lane16);
  }
  stage_dma(0, 0);
  wait_a(0, std::integral_constant<int, 0>{});
  __syncthreads();

  if constexpr ((DBG & 32) != 0) {
    if (threadIdx.x == 0) tl[1] = __builtin_amdgcn_s_memrealtime();
  }
  __builtin_amdgcn_s_setprio(0);
  const int boff = h * BN + 4 * l31;
  const int nstage = a.CIN / KCB;
  f32x4 bv[8];
  // One stage = CH chunks, fully unrolled.  Order of issue inside a chunk: the NEXT chunk's A fragments (into the other register set),
  // in the stage's first chunk then the next stage's window (LDS-DMA), then the hand-counted wait for THIS chunk's fragments --
  // vmcnt retires in order: what may stay in flight is the next chunk's LA loads and, in the stage's first two chunks, the ND DMA
  // instructions issued behind this chunk's fragments (the window has two chunks = 128 MFMAs per wave to land) --, then per k-step 8
  // MFMAs followed by the ds_read_b128 that refills this k-step's B registers with the next chunk's values (one chunk ahead, no
  // second register set: the MFMAs that read them have been issued).  The last chunk of a stage refills nothing -- the next stage's
  // buffer is valid after the barrier only -- and the first chunk's eight reads follow the barrier.
  auto stage = [&](int s, auto last_tag) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_tag)::value;
    const float* blk = xs + (s & 1) * (KCB * BN) + boff;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) bv[ks] = *reinterpret_cast<const f32x4*>(blk + ks * (2 * BN));
#pragma unroll
    for (int sc = 0; sc < CH; ++sc) {
      const int q = s * CH + sc;
      const int cur = sc & 1, nxt = cur ^ 1;
      const bool pre = !(LAST && sc == CH - 1);
      if (pre) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          av[nxt][mi][0] = vm_load16(wp[mi] + (size_t)(q + 1) * 128, lane16);
          av[nxt][mi][1] = vm_load16(wp[mi] + (size_t)(q + 1) * 128 + 64, lane16);
        }
      }
      if (!LAST && sc == 0) stage_dma((s + 1) & 1, s + 1);
      if (!LAST && sc <= 1) wait_a(cur, std::integral_constant<int, LA + ND>{});
      else if (pre) wait_a(cur, std::integral_constant<int, LA>{});
      else wait_a(cur, std::integral_constant<int, 0>{});
      __builtin_amdgcn_sched_barrier(0);
      const float* bnx = blk + (sc + 1) * (KC * BN);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cur][mi][ks >> 2][ks & 3], bv[ks][ni], acc[mi][ni], 0, 0, 0);
        if (sc + 1 < CH) bv[ks] = *reinterpret_cast<const f32x4*>(bnx + ks * (2 * BN));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!LAST) {
      // every wave's DMA instructions must have landed before anyone reads the next buffer: with three or more chunks per stage
      // the third chunk's wait has retired them (they are older than its fragments); with two, wait here -- the next chunk's
      // fragments may stay in flight
      if constexpr (CH <= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LA) : "memory");
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
  __builtin_amdgcn_s_setprio(3);
  lin128_store<MI>(a, acc, ms0, h, b, t0 + 4 * l31, olen);
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
         (a.epi == EPI_STORE || a.epi == EPI_RES) && !a.scale && a.M >= 256 && a.M % 256 == 0 && a.CIN % 64 == 0 && a.ldx >= 4 && a.ldx % 4 == 0 &&
         a.ldo % 4 == 0;
}

template <int MI, int KCB, int WGPC, int DBG = 0>
static int launch_lin128_t(ConvArgs a, int B, int Lmax_out, hipStream_t stream) {
  constexpr int BM = 128 * MI, BN = 128;
  a.mt_per_group = (a.M + BM - 1) / BM;
  dim3 grid((Lmax_out + BN - 1) / BN, a.mt_per_group, B);
  a.mfast = 0;
  a.ragged_enum = (opts().ragged_enum && (a.lengths || a.lengths_out) && B > 1) ? 1 : 0;
  const int mt = a.mt_per_group;
  const long long tt_pad = ((long long)grid.x * B + 7) / 8 * 8;
  a.xcd = ((opts().xcd_order & 1) && mt >= 2) ? 1 : 0;
  if (a.xcd) {
    // M tiles per sweep: the divisor of the M tile count with the least modelled fabric traffic (whole sweeps only: a short last sweep
    // adds a partly filled round of workgroups).  Per sweep every XCD reads its windows once (sweeps x the activation bytes in all) and
    // the sweep's weight slabs once -- if they stay L2-resident next to the streamed windows (<= 3.2 MB of the 4) -- or once per round
    // of co-resident workgroups if they do not.  Measured (profiles/r06/lin_xcd_mg.txt, 32 x 499 frames): fc2 (K = 3072: 1.5 MB slabs
    // AND 1.5 MB windows) 0.776 GB per launch in sweeps of 2 -> 0.523 in one sweep of 6, which this rule now picks; qkv (6 of 18) and
    // fc1 (8 of 24) stay where the "slabs that fit" rule had them; times do not move (the linears are matrix-pipe bound).
    const double slab = (double)BM * a.CIN * sizeof(float);
    const double xbytes = (double)tt_pad * BN * a.CIN * sizeof(float), wbytes = slab * mt;
    int mg = mt;
    double best = 0.0;
    for (int c = 1; c <= mt; ++c) {
      if (mt % c) continue;
      const double rounds = c * slab <= 3.2 * 1024 * 1024 ? 1.0 : (double)((tt_pad / 8 * c + WGPC * 32 - 1) / (WGPC * 32));
      const double cost = (double)(mt / c) * xbytes + 8.0 * wbytes * rounds;
      if (c == 1 || cost <= best) best = cost, mg = c;
    }
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
  if (opts().kernel_dbg == 64 || getenv("DISSC_LIN128_ONE")) lds = 100 * 1024;  // diagnostics: ONE workgroup per CU (one wave per SIMD)
  static DeviceOnce attr_once;
  DISSC_HIP_CHECK(attr_once.max_lds(reinterpret_cast<const void*>(&lin128_kernel<MI, KCB, WGPC, DBG>), 160 * 1024));
  hipLaunchKernelGGL((lin128_kernel<MI, KCB, WGPC, DBG>), grid, dim3(256), lds, stream, a);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

int launch_lin128(const ConvArgs& a, int B, int Lmax_out, hipStream_t stream) {
  if (opts().kernel_dbg == 32) return launch_lin128_t<2, 32, 2, 32>(a, B, Lmax_out, stream);  // timeline stamps (tools/lin128_timeline.py)
  // Tile shape per launch (option "lin128": 1 = this policy; 2 / 3 / 5 force a shape, for the gate records).  All shapes give the same
  // bits.  With scalar-base loads and hand-counted waits (profiles/r06/lin128_gate_v8.txt; B = 32 x T = 499, us: old kernel / 256 x 128
  // two per CU / 128 x 128 three per CU / 128 x 128 two per CU with 64 channels per barrier): fc1 618 / 617 / 558 / 550, fc2 628 / 731 /
  // 536 / 708, qkv 468 / 503 / 419 / 453, out_proj 182 / 212 / 146 / 189, proj 116 / 143 / 100 / 125 -- 128 x 128 tiles, three
  // workgroups per CU (three waves per SIMD, 768 slots: every HuBERT shape fills whole rounds) win everywhere but on the
  // largest grid of short K loops, where the 64-channel stages of the two-per-CU form are 1.5 % ahead.
  int mode = opts().lin128;
  if (mode == 1) {
    const long long ct = (long long)((Lmax_out + 127) / 128) * B;  // column tiles (an upper bound for ragged batches)
    const long long m1 = (ct * ((a.M + 127) / 128) + 255) / 256;   // 128-row tiles per CU
    mode = (m1 >= 12 && a.CIN <= 1024) ? 5 : 3;
  }
  switch (mode) {
    case 2: return launch_lin128_t<2, 32, 2>(a, B, Lmax_out, stream);  // 256 x 128 tiles, two workgroups per CU
    case 3: return launch_lin128_t<1, 32, 3>(a, B, Lmax_out, stream);  // 128 x 128 tiles, three workgroups per CU
    default: return launch_lin128_t<1, 64, 2>(a, B, Lmax_out, stream); // 128 x 128 tiles, two per CU, 64 channels per barrier
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv2s128_kernel: HuBERT's stride-2, k = 3 feature convs (512 -> 512, valid, GELU; conv1 .. conv4 = 43 % of the encoder) on the
// same 256 x 128 tiles, two workgroups per CU.  out[m][t] = sum_c sum_j W[m][c][j] x[c][2 t + j].
//   * B operand: with column l of block ni = frame 4 l + ni, a lane's four columns of tap j read x[c][8 l + 2 ni + j]: ALL twelve B
//     operands of one channel (3 taps x 4 blocks) are the nine consecutive floats w[0 .. 8] = x[c][8 l .. 8 l + 8] of the raw window:
//     two ds_read_b128 + one ds_read_b32 per 24 MFMAs (the 256 x 64 kernel: 2 strided ds_read_b32 per 4), no data movement -- tap j,
//     block ni is register w[2 ni + j].
//   * K order: chunk (16 channels) -> channel pair -> tap (conv_mfma32_kernel: chunk -> tap -> channel pair, which would keep eight
//     windows = 72 registers live across the taps).  Same products, same fp32 MFMA chain length, another summation order: NOT
//     bit-identical to the 256 x 64 kernel (feature error against the float64 / HF goldens unchanged, tests/test_gpu_hubert.py).  The
//     weights are packed for this order (pack_s2_weights128): [32-row subtile][chunk][group g][lane][4], k-step 4 g + e = 3 ks + j.
//   * window in LDS: per channel 64 float4 = frames [2 t0, 2 t0 + 256) in a main block [KCB][256] plus ONE more float4 (the 257th
//     float, w[8] of lane 31) in a tail block [KCB][4]: rows of a power-of-two length (DMA slot -> (row, float4) by shifts), and lane
//     31 reads its w[8] from the tail instead of from its neighbour's w[0].
// DBG (option "kernel_dbg", diagnostics): bit 0: no A loads in the loop, 1: no window DMA, 2: no window reads, 3: no epilogue, 4: no wait +
// barrier per stage, 5: timeline stamps into a.acc (as lin128_kernel)
// ASM: loads and waits through vm_load16 / vm_dma16 / vm_wait (hand-counted vmcnt); false: compiler-scheduled builtins
template <int KCB, int WGPC, int DBG = 0, bool ASM = true, int MI = 2>
__global__ void __launch_bounds__(256, WGPC) conv2s128_kernel(const ConvArgs a) {
  constexpr int BN = 128, NT = 256, NI = 4, KS = 3;
  constexpr int CH = KCB / KC;          // 16-channel chunks per barrier
  constexpr int G = KC / 2 * KS / 4;    // float4 groups of four k-steps per chunk and 32-row subtile (6)
  constexpr int ND = KCB * 64 / NT;     // main-block DMA instructions per thread and stage
  constexpr int STAGE_F = KCB * 256 + KCB * 4;  // floats per stage buffer (main + tail)
  extern __shared__ __attribute__((aligned(16))) float xs[];  // 2 x STAGE_F
  static_assert(G == 6 && KCB % 16 == 0, "k = 3 only");

  int b, bx, by;
  if (!tile128_of(a, b, bx, by)) return;
  unsigned long long* tl = nullptr;
  if constexpr ((DBG & 32) != 0) {
    const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (id < 4096 && a.acc) tl = reinterpret_cast<unsigned long long*>(a.acc) + (size_t)id * 8;
    if (tl && threadIdx.x == 0) {
      tl[0] = __builtin_amdgcn_s_memrealtime();
      tl[4] = ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
      tl[1] = tl[2] = tl[3] = tl[5] = 0;
    }
  }
  const int olen = a.lengths_out ? a.lengths_out[b] : a.olen_default;
  const int t0 = bx * BN;
  if (t0 >= olen) return;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int ms0 = by * (MI * 4) + wave * MI;
  const float* xb = a.x + (size_t)b * a.x_bstride;
  const int tin0 = 2 * t0;
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);

  // DMA, main block: slot e = tid + 256 i -> row (tid >> 6) + 4 i, float4 tid & 63; tail block: thread r < KCB moves row r's 65th float4.
  // Positions beyond the row are clamped into it: they feed columns >= olen only (a valid conv's last output reads x[2 olen] < ldx).
  int tm = tin0 + 4 * (tid & 63);
  tm = tm > a.ldx - 4 ? a.ldx - 4 : tm;
  int tt = tin0 + 256;
  tt = tt > a.ldx - 4 ? a.ldx - 4 : tt;
  // tail: wave w moves rows w KCB/4 .. + KCB/4 - 1 with its first KCB/4 lanes: EVERY wave issues ND + 1 DMA instructions per stage
  // (the hand-counted waits need one count for all waves).  Addresses: wave-uniform row base + the lane's byte offset.
  constexpr int TR = KCB / 4;
  const float* xw = xb + (size_t)wave_s * a.ldx;            // main block: this wave's row of every group of four (uniform)
  const unsigned vm_off = (unsigned)tm * 4u;                // ... and the lane's float4 in it
  const float* xt = xb + (size_t)(wave_s * TR) * a.ldx + tt;  // tail block: rows wave TR + lane (lane < TR)
  const unsigned vt_off = (unsigned)((lane < TR ? lane : 0) * a.ldx) * 4u;
  const size_t row4 = (size_t)4 * a.ldx;
  const unsigned xs_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)xs);
  auto stage_dma = [&](int bufi, int s) __attribute__((always_inline)) {
    const float* p = xw + (size_t)s * KCB * a.ldx;
    float* buf = xs + bufi * STAGE_F;
    const unsigned baddr = xs_addr + bufi * (STAGE_F * 4);
#pragma unroll
    for (int i = 0; i < ND; ++i) {
      if constexpr (ASM) vm_dma16(p + i * row4, vm_off, baddr + (i * NT + wave_s * 64) * 16);
      else
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(p + i * row4 + tm),
                                         (void __attribute__((address_space(3)))*)(buf + (i * NT + wave_s * 64) * 4), 16, 0, 0);
    }
    if (lane < TR) {
      const float* q = xt + (size_t)s * KCB * a.ldx;
      if constexpr (ASM) vm_dma16(q, vt_off, baddr + (KCB * 256 + wave_s * TR * 4) * 4);
      else
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(reinterpret_cast<const char*>(q) + vt_off),
                                         (void __attribute__((address_space(3)))*)(buf + KCB * 256 + wave_s * TR * 4), 16, 0, 0);
    }
  };
  constexpr int D = ND + 1;  // DMA instructions per wave and stage

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  // A fragments: groups of four k-steps, THREE register sets taking turns (group g in set g % 3; G = 6), loaded two groups ahead
  const int ngrp = a.nchunk * G;
  const unsigned lane16 = lane * 16u;
  const f32x4* wp[MI];
  f32x4 av[3][MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    // wave-uniform base (SGPR pair) + the lane's 32-bit byte offset: the saddr form of global_load, no 64-bit VALU address math
    wp[mi] = reinterpret_cast<const f32x4*>(a.wpack2) + (size_t)(by * (MI * 4) + wave_s * MI + mi) * ngrp * 64;
    av[0][mi] = ASM ? vm_load16(wp[mi], lane16) : *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(wp[mi]) + lane16);
    av[1][mi] = ASM ? vm_load16(wp[mi] + 64, lane16)
                    : *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(wp[mi] + 64) + lane16);
  }
  stage_dma(0, 0);
  if constexpr (ASM) {
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      vm_wait<0>(av[0][mi]);
      vm_wait<0>(av[1][mi]);
    }
  } else {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  }
  __syncthreads();

  if constexpr ((DBG & 32) != 0) {
    if (tl && threadIdx.x == 0) tl[1] = __builtin_amdgcn_s_memrealtime();
  }
  // per-lane LDS offsets (floats) inside a stage buffer: w[0 .. 7] at row * 256 + 8 l31; w[8] = the next lane's w[0], or (lane 31) the
  // row's tail float4 -- two affine sequences in the channel pair with different strides
  const int off_w = h * 256 + 8 * l31;
  const int off_8 = l31 < 31 ? off_w + 8 : KCB * 256 + h * 4;
  const int str_8 = l31 < 31 ? 2 * 256 : 2 * 4;
  const int nstage = a.CIN / KCB;
  f32x4 wa[2], wb[2];
  float wc[2];
  auto stage = [&](int s, auto last_tag) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_tag)::value;
    const float* blk = xs + (s & 1) * STAGE_F;
    wa[0] = *reinterpret_cast<const f32x4*>(blk + off_w);
    wb[0] = *reinterpret_cast<const f32x4*>(blk + off_w + 4);
    wc[0] = blk[off_8];
#pragma unroll
    for (int pr = 0; pr < CH * 8; ++pr) {  // channel pairs of the stage
      const int cur = pr & 1, nxt = cur ^ 1;
      if (pr + 1 < CH * 8 && !(DBG & 4)) {  // the next pair's window, behind this pair's 24 MFMAs
        const float* nw = blk + (pr + 1) * (2 * 256);
        wa[nxt] = *reinterpret_cast<const f32x4*>(nw + off_w);
        wb[nxt] = *reinterpret_cast<const f32x4*>(nw + off_w + 4);
        wc[nxt] = blk[off_8 + (pr + 1) * str_8];
      }
      const float w9[9] = {wa[cur][0], wa[cur][1], wa[cur][2], wa[cur][3], wb[cur][0], wb[cur][1], wb[cur][2], wb[cur][3], wc[cur]};
#pragma unroll
      for (int j = 0; j < KS; ++j) {
        const int kk = pr * KS + j;            // k-step inside the stage
        const int g = kk >> 2, e = kk & 3;     // group (of this stage) and component
        if (e == 0) {
          // two groups ahead into the set that group g - 1 has just released; in the stage's first group then the next stage's window
          const int gi = s * (CH * G) + g + 2;
          const int gl = gi < ngrp ? gi : ngrp - 1;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            if (!(DBG & 1))
              av[(g + 2) % 3][mi] = ASM ? vm_load16(wp[mi] + (size_t)gl * 64, lane16)
                                        : *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(wp[mi] + (size_t)gl * 64) + lane16);
          if (!LAST && g == 0 && !(DBG & 2)) stage_dma((s + 1) & 1, s + 1);
          if constexpr (ASM && !(DBG & 3)) {
            // group g's fragments: younger and possibly in flight are groups g + 1, g + 2 (4 loads) and, for the groups loaded before
            // this stage's DMA was issued (g <= 2), its D instructions: the window has three groups (96 MFMAs) to land
            if constexpr (MI == 2) {
              if (!LAST && g <= 2) vm_wait<4 + D>(av[g % 3][0], av[g % 3][1]);
              else vm_wait<4>(av[g % 3][0], av[g % 3][1]);
            } else {
              if (!LAST && g <= 2) vm_wait<2 + D>(av[g % 3][0]);
              else vm_wait<2>(av[g % 3][0]);
            }
          } else if constexpr (ASM) {
            vm_wait<0>(av[g % 3][0]);  // knock-outs change the counts: drain
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g % 3][mi][e], w9[2 * ni + j], acc[mi][ni], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!LAST && !(DBG & 16)) {
      // ASM: this wave's DMA instructions were retired by group 3's wait (they are older than that group's fragments); the barrier
      // makes every wave's part of the window visible.  Compiler-scheduled form: vmcnt(0) here.
      if constexpr (!ASM) __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
    }
  };
  static_assert(CH * G >= 4, "the DMA is retired by the wait of the stage's fourth group");
  static_assert((CH * G) % 3 == 0, "the A register sets must be in the same roles at every stage's start");
#pragma unroll 1
  for (int s = 0; s + 1 < nstage; ++s) stage(s, std::false_type{});
  stage(nstage - 1, std::true_type{});
  if constexpr (ASM) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped prefetches beyond the last group
  if constexpr ((DBG & 32) != 0) {
    if (tl && threadIdx.x == 0) tl[2] = __builtin_amdgcn_s_memrealtime();
  }
  if (DBG & 8) {
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
  lin128_store<MI>(a, acc, ms0, h, b, t0 + 4 * l31, olen);
  if constexpr ((DBG & 32) != 0) {
    if (tl && threadIdx.x == 0) {
      tl[3] = __builtin_amdgcn_s_memrealtime();
      __builtin_amdgcn_s_waitcnt(0x0F70);
      tl[5] = __builtin_amdgcn_s_memrealtime();
    }
  }
}

// w: [Cout][Cin][3] -> [32-row subtile][chunk][group g][lane][4]: lane l, component e of group g = k-step kk = 4 g + e of the chunk =
// (channel pair ks = kk / 3, tap j = kk % 3) -> W[32 ms + (l & 31)][16 c + 2 ks + (l >> 5)][j]
void pack_s2_weights128(const float* w, int Cout, int Cin, std::vector<float>& packed) {
  const int nsub = Cout / 32, nchunk = Cin / KC, G = 6;
  packed.assign((size_t)nsub * nchunk * G * 64 * 4, 0.f);
  for (int ms = 0; ms < nsub; ++ms)
    for (int c = 0; c < nchunk; ++c)
      for (int g = 0; g < G; ++g)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 4; ++e) {
            const int kk = 4 * g + e, ks = kk / 3, j = kk % 3;
            const int co = ms * 32 + (lane & 31), ci = c * KC + 2 * ks + (lane >> 5);
            packed[((((size_t)ms * nchunk + c) * G + g) * 64 + lane) * 4 + e] = w[((size_t)co * Cin + ci) * 3 + j];
          }
}

bool conv2s128_shape(int Cout, int Cin, int KS, int stride, int groups) {
  return stride == 2 && KS == 3 && groups == 1 && Cout % 256 == 0 && Cin % 32 == 0;
}

// option "conv2s128" (Options::conv2s128, default 1): stride-2 k = 3 convs on conv2s128_kernel (0: conv_mfma32_kernel; 2: 32 channels per barrier)
bool conv2s128_supported(const ConvArgs& a) {
  return a.wpack2 && a.KS == 3 && a.dil == 1 && a.groups == 1 && a.up == 1 && a.slope == 1.0f && a.pad_left == 0 && a.prec == 0 &&
         a.epi == EPI_STORE && !a.scale && a.M % 256 == 0 && a.CIN % 32 == 0 && a.ldx >= 4 && a.ldx % 4 == 0 && a.ldo % 4 == 0 &&
         (a.lengths_out || a.olen_default >= 0);
}

template <int KCB, int WGPC, int DBG = 0, bool ASM = true, int MI = 2>
static int launch_conv2s128_t(ConvArgs a, int B, int Lmax_out, hipStream_t stream) {
  constexpr int BM = 128 * MI, BN = 128;
  a.mt_per_group = a.M / BM;
  dim3 grid((Lmax_out + BN - 1) / BN, a.mt_per_group, B);
  a.mfast = 0;
  a.ragged_enum = (opts().ragged_enum && a.lengths_out && B > 1) ? 1 : 0;
  const int mt = a.mt_per_group;
  const long long tt_pad = ((long long)grid.x * B + 7) / 8 * 8;
  a.xcd = ((opts().xcd_order & 2) && mt >= 2) ? 1 : 0;
  if (a.xcd) {
    const double slab = (double)BM * a.CIN * a.KS * sizeof(float);
    int mg = (int)(3.2 * 1024 * 1024 / slab);
    if (mg < 2 || mg > mt) mg = mt;
    while (mt % mg) --mg;
    if (opts().xcd_mg > 0) mg = opts().xcd_mg < mt ? opts().xcd_mg : mt;
    if (tt_pad * mg * ((mt + mg - 1) / mg) > 0x7fffffffLL) {
      a.xcd = 0;
    } else {
      a.xcd_ntile = (int)grid.x;
      a.xcd_nb = B;
      a.xcd_mg = mg;
      a.xcd_span = (int)(tt_pad * mg);
      grid = dim3((unsigned)(tt_pad * mg * ((mt + mg - 1) / mg)), 1, 1);
    }
  }
  const size_t lds = (size_t)2 * (KCB * 256 + KCB * 4) * sizeof(float);
  static DeviceOnce attr_once;
  DISSC_HIP_CHECK(attr_once.max_lds(reinterpret_cast<const void*>(&conv2s128_kernel<KCB, WGPC, DBG, ASM, MI>), 160 * 1024));
  hipLaunchKernelGGL((conv2s128_kernel<KCB, WGPC, DBG, ASM, MI>), grid, dim3(256), lds, stream, a);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

int launch_conv2s128(const ConvArgs& a, int B, int Lmax_out, hipStream_t stream) {
  if (opts().kernel_dbg == 32) return launch_conv2s128_t<32, 2, 32>(a, B, Lmax_out, stream);  // timeline stamps (tools/conv2s128_gate.py)
  // "conv2s128": 1 = 16 channels per barrier (default: conv1..4 10.76 ms against 10.80 with 32, profiles/r06/conv2s128_gate_v3.txt), 3 = 32.
  // (2, the compiler-scheduled form with per-lane 64-bit addresses -- conv1 6 146 us where this one takes 5 715 -- is on record in
  //  the same file and no longer instantiated)
  if (opts().conv2s128 == 3) return launch_conv2s128_t<32, 2, 0, true>(a, B, Lmax_out, stream);
  // (128 x 128 tiles on three workgroups per CU -- launch_conv2s128_t<16, 3, 0, true, 1>, the shape that won for the linears -- is a tie
  //  here: conv1..4 10.82 ms against 10.78, profiles/r06/conv2s128_mi1.txt; not instantiated)
  return launch_conv2s128_t<16, 2, 0, true>(a, B, Lmax_out, stream);
}

}  // namespace dissc
