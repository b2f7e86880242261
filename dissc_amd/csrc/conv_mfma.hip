// Dilated "same" Conv1d / phase-decomposed ConvTranspose1d as an implicit GEMM on the
// gfx950 fp32 matrix pipe (v_mfma_f32_16x16x4_f32: exact f32, 157 TFLOP/s peak).
//
//   D[row, t] = bias[row] + sum_{ci, j} W[row, ci, j] * act(x[ci, t + (j - (KS-1)/2) * dil])
//
// GEMM view: M = output rows (Cout, or Cout*stride for ConvTranspose), N = time,
// K = Cin*KS.  One workgroup owns a BM x BN output tile of ONE utterance:
//   - the input window [16 channels][BN + (KS-1)*dil] of chunk c+1 is fetched with
//     16-byte aligned loads into registers while chunk c is on the matrix pipe, then
//     written (leaky-ReLU + the utterance's zero padding applied) to the other half of a
//     double-buffered LDS tile: one barrier per chunk, HBM latency hidden behind MFMAs;
//   - the weight (A) fragments are pre-packed on the host in exactly the MFMA lane
//     order, so every wave streams them L2 -> VGPR with one coalesced dwordx4 per lane
//     per (16 rows x 16 channels x 1 tap), prefetched one tap ahead;
//   - B fragments are ds_read_b32 (lanes 0-15 / 16-31 hit rows XW apart, XW%32==16
//     => conflict free);
//   - the epilogue transposes each wave's accumulators through a private LDS patch so
//     bias/residual/MRF traffic is 16 B per lane in >=256 B runs per output row.
// Replaces the MIOpen/cuDNN conv1d / conv_transpose1d calls behind reference
// sr/models.py:34-41 (ResBlock1), :99-102 (conv_pre, ups).
#include "common.h"

namespace dissc {

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

__device__ __forceinline__ float gelu_exact(float v) { return 0.5f * v * (1.f + erf_1ulp(v * 0.70710678118654752440f)); }

// STRIDE: input step per output position (1; 2 for HuBERT's strided feature convs).
// SPAN: largest (KS-1)*dil the staging registers are sized for.
// CPB: 16-channel chunks staged per barrier (4 for 1x1 convs, whose chunk is a single tap).
template <int MI, int NI, int WM, int WN, int STRIDE, int SPAN, int CPB = 1>
__global__ void __launch_bounds__(64 * WM * WN, 2) conv_mfma_kernel(const ConvArgs a) {
  constexpr int NT = 64 * WM * WN;
  constexpr int BN = 16 * NI * WN;
  constexpr int KCB = KC * CPB;  // channels staged per barrier
  constexpr int XW_MAX = ((BN - 1) * STRIDE + 1 + SPAN + 3 + 31) / 32 * 32 + 16;
  constexpr int SV = (KCB * (XW_MAX / 4) + NT - 1) / NT;  // float4 staging slots per thread
  constexpr int CW = 16 * NI + 4;                        // epilogue patch row stride
  extern __shared__ __attribute__((aligned(16))) float xs[];  // 2 x [KCB][XW] | NW x [16][CW]

  const int b = blockIdx.z;
  const int len = (a.lengths ? a.lengths[b] * a.len_mul : a.len_default);  // valid INPUT positions
  const int olen = a.lengths_out ? a.lengths_out[b] : (a.olen_default >= 0 ? a.olen_default : len);
  const int t0 = blockIdx.x * BN;
  if (t0 >= olen) return;
  const int grp = blockIdx.y / a.mt_per_group;
  const int mt = blockIdx.y - grp * a.mt_per_group;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l15 = lane & 15, g = lane >> 4;
  const int XW = a.XW;
  const int NV = XW >> 2;
  const int tin0 = t0 * STRIDE - a.pad_left;
  const int tb = tin0 & ~3;   // 16-byte aligned window start (may be negative)
  const int sh = tin0 - tb;   // 0..3
  const int nq = a.nchunk * a.KS;
  const int ms0 = mt * (MI * WM) + wm * MI;  // 16-row subtile within the group
  const float slope = a.slope;
  const float* xb = a.x + (size_t)b * a.x_bstride + (size_t)grp * a.CIN * a.ldx;

  // this thread's float4 staging slots: slot i = tid + i*NT -> (row r, vec v)
  const int r0 = tid / NV, v0 = tid - r0 * NV;
  const int dr = NT / NV, dv = NT - dr * NV;

  // Staging: every slot issues an unconditional, in-bounds 16-byte load (addresses are
  // clamped, never predicated, so nothing waits at the load site); zero padding, the
  // ragged tail and leaky-ReLU are applied when the registers are written to LDS.
  f32x4 sv[SV];
  auto stage_load = [&](int c) {
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
        *reinterpret_cast<f32x4*>(buf + r * XW + 4 * v) = val;
      }
      v += dv;
      r += dr;
      if (v >= NV) { v -= NV; ++r; }
    }
  };

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 av[MI], avn[MI];
  // scalar-base loads (common.h: wave_rsrc): this wave's MI subtiles are one contiguous slab [mi][block][lane]
  const int slab = __builtin_amdgcn_readfirstlane(grp * a.nsub_group + ms0);
  const __amdgpu_buffer_rsrc_t wrs = wave_rsrc(reinterpret_cast<const f32x4*>(a.wpack) + (size_t)slab * nq * 64, (unsigned)(MI * nq) * 1024u);
  const unsigned lane16 = lane * 16u;
  auto a_load = [&](int mi, int bl) __attribute__((always_inline)) { return rsrc_load16(wrs, lane16, (unsigned)(mi * nq + bl) * 1024u); };
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) av[mi] = a_load(mi, 0);

  stage_load(0);
  stage_store(xs, 0);
  __syncthreads();

#define DISSC_MFMA_STEP(CQ, BV)                                                              \
  _Pragma("unroll") for (int mi = 0; mi < MI; ++mi)                                          \
  _Pragma("unroll") for (int ni = 0; ni < NI; ++ni)                                          \
      acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi][CQ], BV[ni], acc[mi][ni], 0, 0, 0);

  const int boff = g * XW + sh + (wn * (16 * NI) + l15) * STRIDE;
  const int XW4 = 4 * XW;
  int q = 0;
  const int nblk = (a.nchunk + CPB - 1) / CPB;
  for (int cb = 0; cb < nblk; ++cb) {
    const float* blk = xs + (cb & 1) * (KCB * XW) + boff;
    const bool more = cb + 1 < nblk;
#pragma unroll 1
    for (int sc = 0; sc < CPB; ++sc) {
      if (cb * CPB + sc >= a.nchunk) break;
      const float* bj = blk + sc * (KC * XW);
      float b0[NI], b1[NI], b2[NI], b3[NI], b0n[NI];
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b0[ni] = bj[ni * 16 * STRIDE];
      for (int j = 0; j < a.KS; ++j, ++q) {
        const int qn = (q + 1 < nq) ? q + 1 : q;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) avn[mi] = a_load(mi, qn);
        if (j == 0 && sc == 0 && more) stage_load(cb + 1);  // in flight behind this block's MFMAs
        __builtin_amdgcn_sched_barrier(0);      // keep the prefetches ahead of the MFMAs
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          b1[ni] = bj[XW4 + ni * 16 * STRIDE];
          b2[ni] = bj[2 * XW4 + ni * 16 * STRIDE];
          b3[ni] = bj[3 * XW4 + ni * 16 * STRIDE];
        }
        DISSC_MFMA_STEP(0, b0)
        bj += a.dil;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b0n[ni] = bj[ni * 16 * STRIDE];  // next tap's first k-step
        DISSC_MFMA_STEP(1, b1)
        DISSC_MFMA_STEP(2, b2)
        DISSC_MFMA_STEP(3, b3)
        __builtin_amdgcn_sched_barrier(0);  // the register rotation below must not creep upwards
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b0[ni] = b0n[ni];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) av[mi] = avn[mi];
      }
    }
    if (more) stage_store(xs + ((cb + 1) & 1) * (KCB * XW), cb + 1);
    __syncthreads();
  }
#undef DISSC_MFMA_STEP

  const int epi = a.epi;
  const size_t ob = (size_t)b * a.o_bstride;
  if (a.up != 1) {
    // ConvTranspose pixel shuffle straight from registers: row = co*up + p -> out[co][t*up + p].
    // C/D layout of 16x16x4: col = lane & 15 (time), row = 4*(lane>>4) + r.
    const int tbase = t0 + wn * (16 * NI) + l15;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = (ms0 + mi) * 16 + g * 4 + r;
        if (row >= a.M) continue;
        const float bz = a.bias[row];  // ConvTranspose path: groups == 1
        const int co = row / a.up_np;
        const int p = a.up_p0 + row - co * a.up_np;
        const size_t rowoff = ob + (size_t)co * a.ldo + p;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int t = tbase + ni * 16;
          if (t < olen) a.out[rowoff + (size_t)t * a.up] = acc[mi][ni][r] + bz;
        }
      }
    }
    return;
  }

  // Transposed epilogue: registers -> wave-private LDS patch [16][CW] -> 16 B per lane.
  float* ep = xs + wave * (16 * CW);
  constexpr int LPR = 4 * NI;        // float4 lanes per output row
  constexpr int RPP = 64 / LPR;      // rows per pass
  const int prow = lane / LPR, pc4 = lane % LPR;
  const int tcol = t0 + wn * (16 * NI) + 4 * pc4;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) ep[(4 * g + r) * CW + ni * 16 + l15] = acc[mi][ni][r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < NI; ++p) {
      const int rl = p * RPP + prow;
      const int row = (ms0 + mi) * 16 + rl;  // within the group
      f32x4 v = *reinterpret_cast<const f32x4*>(ep + rl * CW + 4 * pc4);
      if (row >= a.M || tcol >= olen) continue;
      const int prow_idx = grp * a.nsub_group * 16 + row;  // bias/scale/shift are [groups][Mpad]
      const float bz = a.bias[prow_idx];
      v[0] += bz; v[1] += bz; v[2] += bz; v[3] += bz;
      if (a.scale) {  // eval-mode BatchNorm1d as PyTorch evaluates it: x * alpha + beta
        const float sc = a.scale[prow_idx], sf = a.shift[prow_idx];
        v[0] = v[0] * sc + sf; v[1] = v[1] * sc + sf; v[2] = v[2] * sc + sf; v[3] = v[3] * sc + sf;
      }
      if (a.act == 1) {
        v[0] = gelu_exact(v[0]); v[1] = gelu_exact(v[1]); v[2] = gelu_exact(v[2]); v[3] = gelu_exact(v[3]);
      }
      const size_t idx = ob + (size_t)(grp * a.M + row) * a.ldo + tcol;
      const int nv = olen - tcol;  // >= 1
      if (nv >= 4) {
        if (epi == EPI_STORE) {
          *reinterpret_cast<f32x4*>(a.out + idx) = v;
        } else {
          const f32x4 rs = *reinterpret_cast<const f32x4*>(a.res + idx);
          v[0] += rs[0]; v[1] += rs[1]; v[2] += rs[2]; v[3] += rs[3];
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
        }
      } else {
        for (int e = 0; e < nv; ++e) {
          float x = v[e];
          if (epi == EPI_STORE) {
            a.out[idx + e] = x;
          } else {
            x += a.res[idx + e];
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
    }
  }
}

// ---------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------
static int pick_bm(int M) {
  if (M == 48) return 48;  // HuBERT's positional conv: 48 rows per group = three 16-row tiles, nothing padded
  if (M >= 256) return 256;
  if (M >= 128) return 128;
  if (M >= 64) return 64;
  if (M >= 32) return 32;
  return 16;
}

// Tile shapes: wave tile = (16*MI) x (16*NI), WM x WN waves, BM = 16*MI*WM, BN = 16*NI*WN.
struct TileCfg { int MI, NI, WM, WN; };
static const TileCfg kCfgs[] = {
    {4, 4, 4, 1},  // 0: 256 x 64
    {4, 4, 2, 2},  // 1: 128 x 128
    {4, 4, 1, 4},  // 2:  64 x 256
    {2, 8, 1, 4},  // 3:  32 x 512
    {1, 8, 1, 4},  // 4:  16 x 512
    {2, 4, 1, 4},  // 5:  32 x 256
    {1, 4, 1, 4},  // 6:  16 x 256
    {2, 4, 2, 2},  // 7:  64 x 128
    {2, 4, 4, 1},  // 8: 128 x 64
    {4, 2, 4, 2},  // 9: 256 x 64, 8 waves
    {3, 4, 1, 4},  // 10: 48 x 256 (pick_bm(48) only)
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

void conv_set_cfg(int bm_class, int cfg) {
  if (bm_class >= 0 && bm_class < 5 && cfg >= 0 && cfg < kNumCfgs) g_defaults.cfg_for_bm[bm_class] = cfg;
}

int conv_cfg(int M) {
  const int bm = pick_bm(M);
  if (bm == 48) return 10;
  int cls = 0;
  while ((16 << cls) < bm) ++cls;
  int cfg = opts().cfg_for_bm[cls];
  if (16 * kCfgs[cfg].MI * kCfgs[cfg].WM > bm) cfg = 4 - cls;  // override must divide the padded M
  return cfg;
}

int conv_tile_bn(int M) {
  const TileCfg& c = kCfgs[conv_cfg(M)];
  return 16 * c.NI * c.WN;
}

int conv_xw(int M, int KS, int dil, int stride, int m32, int bn_) {
  const int bn = bn_ > 0 ? bn_ : (m32 ? conv32_tile_bn(M) : conv_tile_bn(M));
  const int need = (bn - 1) * stride + 1 + (KS - 1) * dil + 3;  // +3: aligned window start
  int xw = (need + 31) / 32 * 32 + 16;
  if (xw - 32 >= need) xw -= 32;
  return xw;
}

// w: [Cout][Cin][KS] with Cin = input channels PER GROUP; rows of group g are
// [g*Cout/groups, (g+1)*Cout/groups).  Mpad = padded rows PER GROUP.
void pack_conv_weights(const float* w, int Cout, int Cin, int KS, std::vector<float>& packed,
                       int& Mpad, int& nchunk, int groups) {
  const int Mg = Cout / groups;
  const int bm = pick_bm(Mg);
  Mpad = (Mg + bm - 1) / bm * bm;
  nchunk = (Cin + KC - 1) / KC;
  const int nsub = Mpad / 16;
  packed.assign((size_t)groups * nsub * nchunk * KS * 64 * 4, 0.f);
  for (int gi = 0; gi < groups; ++gi)
    for (int ms = 0; ms < nsub; ++ms)
      for (int c = 0; c < nchunk; ++c)
        for (int j = 0; j < KS; ++j)
          for (int lane = 0; lane < 64; ++lane)
            for (int cq = 0; cq < 4; ++cq) {
              const int co = ms * 16 + (lane & 15);
              const int ci = c * KC + cq * 4 + (lane >> 4);
              if (co < Mg && ci < Cin)
                packed[(((((size_t)gi * nsub + ms) * nchunk + c) * KS + j) * 64 + lane) * 4 + cq] =
                    w[((size_t)(gi * Mg + co) * Cin + ci) * KS + j];
            }
}

void convT_phase_weights(const float* w, int Cin, int Cout, int k, int s, int p0, int np, int dlo,
                         int ntap, std::vector<float>& wc) {
  const int pad = (k - s) / 2;
  wc.assign((size_t)Cout * np * Cin * ntap, 0.f);
  for (int co = 0; co < Cout; ++co)
    for (int pi = 0; pi < np; ++pi)
      for (int ci = 0; ci < Cin; ++ci)
        for (int j = 0; j < ntap; ++j) {
          const int kk = (p0 + pi) + pad - s * (dlo + j);
          if (kk >= 0 && kk < k)
            wc[((size_t)(co * np + pi) * Cin + ci) * ntap + j] = w[((size_t)ci * Cout + co) * k + kk];
        }
}

template <int MI, int NI, int WM, int WN, int STRIDE, int SPAN, int CPB = 1>
static int launch_t(ConvArgs a, int B, int Lmax_out, hipStream_t stream) {
  constexpr int BM = 16 * MI * WM, BN = 16 * NI * WN, NW = WM * WN;
  constexpr int CW = 16 * NI + 4;
  a.mt_per_group = (a.M + BM - 1) / BM;
  dim3 grid((Lmax_out + BN - 1) / BN, a.mt_per_group * a.groups, B);
  size_t lds_f = (size_t)2 * KC * CPB * a.XW;
  if (lds_f < (size_t)NW * 16 * CW) lds_f = (size_t)NW * 16 * CW;
  const size_t lds = lds_f * sizeof(float);
  static DeviceOnce attr_once;  // per device (common.h)
  DISSC_HIP_CHECK(attr_once.max_lds(reinterpret_cast<const void*>(&conv_mfma_kernel<MI, NI, WM, WN, STRIDE, SPAN, CPB>), 160 * 1024));
  hipLaunchKernelGGL((conv_mfma_kernel<MI, NI, WM, WN, STRIDE, SPAN, CPB>), grid, dim3(64 * WM * WN), lds,
                     stream, a);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

int launch_conv(const ConvArgs& a, int B, int Lmax_out, int stride, hipStream_t stream) {
  if (B <= 0 || Lmax_out <= 0) return DISSC_OK;
  if (B > 65535) {
    set_error("launch_conv: batch %d exceeds grid.z limit", B);
    return DISSC_EINVAL;
  }
  const int span = (a.KS - 1) * a.dil;
  if ((a.ldx & 3) || (a.ldo & 3) || ((uintptr_t)a.x & 15) || ((uintptr_t)a.out & 15 && a.out) ||
      ((uintptr_t)a.res & 15) || ((uintptr_t)a.acc & 15)) {
    set_error("launch_conv: activations must be 16-byte aligned with row strides %% 4 == 0");
    return DISSC_EINVAL;
  }
  if (a.m32) return launch_conv32(a, B, Lmax_out, stride, stream);
  // (the 16x16x4 kernel ignores a.mfast)
  const int cfg = conv_cfg(a.M);
  if (stride == 2 && span <= MAX_TAP_SPAN && a.up == 1) {
    // HuBERT feature convs (512 -> 512, k3/k2 s2): only the 256x64 tile is instantiated
    if (cfg == 0) return launch_t<4, 4, 4, 1, 2, MAX_TAP_SPAN>(a, B, Lmax_out, stream);
    set_error("launch_conv: stride 2 needs >= 256 output rows (got %d)", a.M);
    return DISSC_EINVAL;
  }
  if (stride != 1) {
    set_error("launch_conv: stride %d unsupported", stride);
    return DISSC_EINVAL;
  }
  if (span > MAX_TAP_SPAN) {
    // wide taps (HuBERT positional conv k=128, 48 rows per group): 32x256 tile only
    if (span <= WIDE_TAP_SPAN && cfg == 5) return launch_t<2, 4, 1, 4, 1, WIDE_TAP_SPAN>(a, B, Lmax_out, stream);
    if (span <= WIDE_TAP_SPAN && cfg == 10) return launch_t<3, 4, 1, 4, 1, WIDE_TAP_SPAN>(a, B, Lmax_out, stream);
    set_error("launch_conv: kernel %d x dilation %d unsupported for %d rows", a.KS, a.dil, a.M);
    return DISSC_EINVAL;
  }
  if (span == 0 && cfg == 0 && a.nchunk >= 8)  // 1x1 convs (HuBERT linears): 64 channels per barrier
    return launch_t<4, 4, 4, 1, 1, 0, 4>(a, B, Lmax_out, stream);
  switch (cfg) {
    case 0: return launch_t<4, 4, 4, 1, 1, MAX_TAP_SPAN>(a, B, Lmax_out, stream);
    case 1: return launch_t<4, 4, 2, 2, 1, MAX_TAP_SPAN>(a, B, Lmax_out, stream);
    case 2: return launch_t<4, 4, 1, 4, 1, MAX_TAP_SPAN>(a, B, Lmax_out, stream);
    case 3: return launch_t<2, 8, 1, 4, 1, MAX_TAP_SPAN>(a, B, Lmax_out, stream);
    case 4: return launch_t<1, 8, 1, 4, 1, MAX_TAP_SPAN>(a, B, Lmax_out, stream);
    case 5: return launch_t<2, 4, 1, 4, 1, MAX_TAP_SPAN>(a, B, Lmax_out, stream);
    case 6: return launch_t<1, 4, 1, 4, 1, MAX_TAP_SPAN>(a, B, Lmax_out, stream);
    case 7: return launch_t<2, 4, 2, 2, 1, MAX_TAP_SPAN>(a, B, Lmax_out, stream);
    case 8: return launch_t<2, 4, 4, 1, 1, MAX_TAP_SPAN>(a, B, Lmax_out, stream);
    case 10: return launch_t<3, 4, 1, 4, 1, MAX_TAP_SPAN>(a, B, Lmax_out, stream);
    default: return launch_t<4, 2, 4, 2, 1, MAX_TAP_SPAN>(a, B, Lmax_out, stream);
  }
}

}  // namespace dissc
