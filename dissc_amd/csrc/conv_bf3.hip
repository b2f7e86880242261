// conv_bf3_kernel: the implicit-GEMM conv in split-bf16 ("bf16x3") arithmetic, for the generator
// when the "precision" option is 1 (dissc_set_option).  The default fp32 path never comes here.
//
// Every fp32 operand v is split into bf16 halves hi = bf16(v), lo = bf16(v - hi) and each product is
// formed as hi*hi + hi*lo + lo*hi on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16, fp32
// accumulate): ~2^-17 relative product error instead of exact fp32 (waveform RMS error vs the
// reference ~4e-6, bar 1e-4), at several times the fp32-MFMA rate.
//
// Same tile geometry, weight streaming and epilogue as conv_mfma32_kernel (conv_mfma32.hip); what
// differs is the staging: the fp32 -> (hi, lo) split is done ONCE per staged value, when the
// prefetched registers are written to LDS, into four planes [channel octet][hi|lo][column][8 bf16]
// (same footprint as the fp32 tile).  A B fragment of the 32x32x16 MFMA -- 8 consecutive channels
// of one time step -- is then a single conflict-free ds_read_b128 at any tap offset, and the tap
// loop holds nothing but weight loads, LDS reads and MFMAs.
// (A producer/consumer wave split of this kernel was measured slower on every layer shape.)
#include <string.h>

#include "common.h"
#include "conv_epilogue32.h"

namespace dissc {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// window width for a BN-column tile and taps spanning `span` inputs (multiple of 4; +3 for the
// 16-byte alignment of the window start)
__host__ __device__ constexpr int bf3_xw(int BN, int span) { return (BN + span + 3 + 3) & ~3; }

template <int MI, int NI, int WM, int WN>
__global__ void __launch_bounds__(64 * WM * WN, 2) conv_bf3_kernel(const ConvArgs a) {
  constexpr int NT = 64 * WM * WN;
  constexpr int BN = 32 * NI * WN;
  static_assert(2 * (bf3_xw(BN, MAX_TAP_SPAN) / 4) <= NT, "one staging slot per thread");
  extern __shared__ __attribute__((aligned(16))) float xs[];  // 2 x 4 planes [XW] x 16 B | epilogue patches

  const int b = blockIdx.z;
  const int len = (a.lengths ? a.lengths[b] * a.len_mul : a.len_default);  // valid INPUT positions
  const int olen = a.lengths_out ? a.lengths_out[b] : (a.olen_default >= 0 ? a.olen_default : len);
  const int t0 = blockIdx.x * BN;
  if (t0 >= olen) return;
  const int mt = blockIdx.y;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, h = lane >> 5;
  const int XW = a.XW, NV = XW >> 2;
  const int tin0 = t0 - a.pad_left;
  const int tb = tin0 & ~3;  // 16-byte aligned window start (may be negative)
  const int sh = tin0 - tb;  // 0..3
  const int nq = a.nchunk * a.KS;
  const int ms0 = mt * (MI * WM) + wm * MI;  // 32-row subtile
  const float slope = a.slope;
  const float* xb = a.x + (size_t)b * a.x_bstride;
  bf16x8* const lds8 = reinterpret_cast<bf16x8*>(xs);
  const int BUF = 4 * XW;  // one buffer, in 16-byte units

  // Staging slot of this thread: (channel octet ph, time quad pv) -> 8 unconditional, clamped
  // 16-byte loads; padding, ragged tail, leaky-ReLU and the split are applied at the LDS store,
  // one 16-byte store of 8 hi halves and one of 8 lo halves per time step.
  f32x4 sp[8];
  const int ph = tid / NV, pv = tid - ph * NV;
  auto stage_load = [&](int c) {
    if (ph < 2) {
      int t = tb + 4 * pv;
      t = t < 0 ? 0 : (t > a.ldx - 4 ? a.ldx - 4 : t);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int ci = c * KC + 8 * ph + e;
        ci = ci < a.CIN ? ci : a.CIN - 1;
        sp[e] = *reinterpret_cast<const f32x4*>(xb + (size_t)ci * a.ldx + t);
      }
    }
  };
  auto stage_store = [&](int c) {
    if (ph < 2) {
      const int t = tb + 4 * pv;
      bf16x8* pl = lds8 + (c & 1) * BUF + (2 * ph) * XW + 4 * pv;
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const bool tok = (t + tt) >= 0 && (t + tt) < len;
        bf16x8 vh, vl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const bool ok = tok && (c * KC + 8 * ph + e) < a.CIN;
          const float x = ok ? lrelu(sp[e][tt], slope) : 0.f;
          const __bf16 xh = (__bf16)x;
          vh[e] = xh;
          vl[e] = (__bf16)(x - (float)xh);
        }
        pl[tt] = vh;
        pl[tt + XW] = vl;
      }
    }
  };

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

  // A fragments of step q (= chunk * KS + tap): [ms32][q][hi|lo][lane] x 8 bf16
  const f32x4* wp[MI];
  f32x4 av[MI][2], avn[MI][2];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    wp[mi] = reinterpret_cast<const f32x4*>(a.wpack) + (size_t)(ms0 + mi) * nq * 128 + lane;
    av[mi][0] = wp[mi][0];
    av[mi][1] = wp[mi][64];
  }

  stage_load(0);
  stage_store(0);
  __syncthreads();

  const int boff8 = (2 * h) * XW + sh + wn * (32 * NI) + l31;
  int q = 0;
  for (int c = 0; c < a.nchunk; ++c) {
    const bool more = c + 1 < a.nchunk;
    const bf16x8* bj = lds8 + (c & 1) * BUF + boff8;
    for (int j = 0; j < a.KS; ++j, ++q) {
      const int qn = (q + 1 < nq) ? q + 1 : q;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        avn[mi][0] = wp[mi][(size_t)qn * 128];
        avn[mi][1] = wp[mi][(size_t)qn * 128 + 64];
      }
      if (j == 0 && more) stage_load(c + 1);  // in flight behind this chunk's MFMAs
      __builtin_amdgcn_sched_barrier(0);      // keep the prefetches ahead of the MFMAs
      bf16x8 bh[NI], bl[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        bh[ni] = bj[ni * 32];
        bl[ni] = bj[ni * 32 + XW];
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const bf16x8 ah = __builtin_bit_cast(bf16x8, av[mi][0]);
        const bf16x8 al = __builtin_bit_cast(bf16x8, av[mi][1]);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[ni], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[ni], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[ni], acc[mi][ni], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // the register rotation below must not creep upwards
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        av[mi][0] = avn[mi][0];
        av[mi][1] = avn[mi][1];
      }
      bj += a.dil;
    }
    if (more) stage_store(c + 1);
    __syncthreads();
  }

  conv_epilogue32<MI, NI>(a, acc, xs, b, t0, olen, 0, ms0, wn);
}

// ---- host side ---------------------------------------------------------------------------

static inline uint16_t f32_to_bf16_rne(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t h) {
  const uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// Tile shapes, by output-row class (same classes as the fp32 kernel): BM = 32*MI*WM, BN = 32*NI*WN.
struct TileBf3 { int MI, NI, WM, WN; };
static const TileBf3 kBf3[4] = {
    {1, 2, 1, 4},  // M <  64:  32 x 256
    {1, 2, 2, 2},  // M < 128:  64 x 128
    {2, 4, 2, 2},  // M < 256: 128 x 256  (128 accumulator registers per lane: the short bf16 MFMAs
    {2, 4, 4, 1},  // else   : 256 x 128   need more work per weight fragment than the fp32 ones)
};
static int bf3_class(int M) { return M >= 256 ? 3 : (M >= 128 ? 2 : (M >= 64 ? 1 : 0)); }
static int bf3_bm(int M) { return 32 << bf3_class(M); }

int conv_bf3_tile_bn(int M) {
  const TileBf3& t = kBf3[bf3_class(M)];
  return 32 * t.NI * t.WN;
}

// w: [Cout][Cin][KS].  Layout [ms32][step q = chunk*KS + tap][hi|lo][lane] x 8 bf16, lane l,
// element e -> W[32*ms + (l & 31)][16*chunk + 8*(l >> 5) + e][tap].
void pack_conv_weights_bf3(const float* w, int Cout, int Cin, int KS, std::vector<float>& packed,
                           int& Mpad, int& nchunk) {
  const int bm = bf3_bm(Cout);
  Mpad = (Cout + bm - 1) / bm * bm;
  nchunk = (Cin + KC - 1) / KC;
  const int nsub = Mpad / 32;
  const int nq = nchunk * KS;
  packed.assign((size_t)nsub * nq * 2 * 64 * 4, 0.f);
  uint16_t* p16 = reinterpret_cast<uint16_t*>(packed.data());
  for (int ms = 0; ms < nsub; ++ms)
    for (int c = 0; c < nchunk; ++c)
      for (int j = 0; j < KS; ++j)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 8; ++e) {
            const int co = ms * 32 + (lane & 31);
            const int ci = c * KC + 8 * (lane >> 5) + e;
            if (co >= Cout || ci >= Cin) continue;
            const float v = w[((size_t)co * Cin + ci) * KS + j];
            const uint16_t hi = f32_to_bf16_rne(v);
            const uint16_t lo = f32_to_bf16_rne(v - bf16_to_f32(hi));
            const size_t slot = (((size_t)ms * nq + (size_t)c * KS + j) * 2) * 64 + lane;
            p16[slot * 8 + e] = hi;
            p16[(slot + 64) * 8 + e] = lo;
          }
}

template <int MI, int NI, int WM, int WN>
static int launch_bf3_t(ConvArgs a, int B, int Lmax_out, hipStream_t stream) {
  constexpr int BM = 32 * MI * WM, BN = 32 * NI * WN;
  constexpr int CW = 32 * NI + 4;
  a.XW = bf3_xw(BN, (a.KS - 1) * a.dil);
  a.mt_per_group = (a.M + BM - 1) / BM;
  dim3 grid((Lmax_out + BN - 1) / BN, a.mt_per_group, B);
  size_t lds = (size_t)2 * 4 * a.XW * 16;
  if (lds < (size_t)WM * WN * 8 * CW * sizeof(float)) lds = (size_t)WM * WN * 8 * CW * sizeof(float);
  static DeviceOnce attr_once;  // per device (common.h)
  DISSC_HIP_CHECK(attr_once.max_lds(reinterpret_cast<const void*>(&conv_bf3_kernel<MI, NI, WM, WN>), 160 * 1024));
  hipLaunchKernelGGL((conv_bf3_kernel<MI, NI, WM, WN>), grid, dim3(64 * WM * WN), lds, stream, a);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

int launch_conv_bf3(const ConvArgs& a, int B, int Lmax_out, hipStream_t stream) {
  if (a.groups != 1 || (a.KS - 1) * a.dil > MAX_TAP_SPAN || a.M < 32) {
    set_error("launch_conv_bf3: unsupported layer (groups %d, k %d, dilation %d, rows %d)", a.groups, a.KS,
              a.dil, a.M);
    return DISSC_EINVAL;
  }
  const int cls = bf3_class(a.M);
  if (cls >= 2 && opts().small_grid) {
    // small grids (short or single utterances): 64 x 128 tiles instead of the 128-accumulator ones,
    // same MFMA sequence per output element, bit-identical result (cf. conv32_pick_cfg)
    const TileBf3& t = kBf3[cls];
    const int bm = 32 * t.MI * t.WM, bn = 32 * t.NI * t.WN;
    const long long nwg = (long long)((Lmax_out + bn - 1) / bn) * ((a.M + bm - 1) / bm) * B;
    if (nwg < 256LL * opts().small_grid) return launch_bf3_t<1, 2, 2, 2>(a, B, Lmax_out, stream);
  }
  switch (cls) {
    case 0: return launch_bf3_t<1, 2, 1, 4>(a, B, Lmax_out, stream);
    case 1: return launch_bf3_t<1, 2, 2, 2>(a, B, Lmax_out, stream);
    case 2: return launch_bf3_t<2, 4, 2, 2>(a, B, Lmax_out, stream);
    default: return launch_bf3_t<2, 4, 4, 1>(a, B, Lmax_out, stream);
  }
}

}  // namespace dissc
