// conv16_stream_kernel: the 16-channel -> 16-channel convs of the generator's last stage
// (C = 16, L = 320 x frames: 6.4 % of the FLOPs but bandwidth/latency sensitive).
//
// Same math and epilogue modes as conv_mfma_kernel (16x16x4 fp32 MFMA, one 16-channel chunk),
// but one workgroup streams several consecutive time tiles through a software pipeline:
//   - all taps of the weights stay in registers (KS float4 per lane, loaded once per workgroup);
//   - the input window of tile i+1 and the residual of tile i are fetched into registers while
//     tile i is on the matrix pipe; the window goes to the other half of a double-buffered LDS tile;
//   - one barrier per tile; the epilogue (wave-private LDS transpose -> 16 B per lane) of tile i
//     overlaps with the other waves' MFMAs.
// This removes the per-workgroup load -> barrier -> compute -> store serialisation that limits the
// general kernel when a workgroup's whole K loop is a single chunk.
#include "common.h"

namespace dissc {

__device__ __forceinline__ float lrelu16(float v, float slope) { return v > 0.f ? v : v * slope; }

template <int KS, int NI, int WN>
__global__ void __launch_bounds__(64 * WN) conv16_stream_kernel(const ConvArgs a, int tiles_per_block) {
  constexpr int NT = 64 * WN;
  constexpr int BN = 16 * NI * WN;
  constexpr int XW_MAX = (BN + MAX_TAP_SPAN + 3 + 31) / 32 * 32 + 16;
  constexpr int SV = (KC * (XW_MAX / 4) + NT - 1) / NT;  // input float4 slots per thread
  constexpr int CW = 16 * NI + 4;
  constexpr int LPR = 4 * NI;   // float4 lanes per output row of a wave
  constexpr int RPP = 64 / LPR; // rows per pass
  extern __shared__ __attribute__((aligned(16))) float xs[];  // 2 x [16][XW] | WN x [16][CW]

  const int b = blockIdx.z;
  const int len = (a.lengths ? a.lengths[b] * a.len_mul : a.len_default);
  const int tile0 = blockIdx.x * tiles_per_block;
  if (tile0 * BN >= len) return;
  int ntile = (len - tile0 * BN + BN - 1) / BN;
  ntile = ntile < tiles_per_block ? ntile : tiles_per_block;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int XW = a.XW, NV = XW >> 2;
  const int halo = a.pad_left;
  const float slope = a.slope;
  const float* xb = a.x + (size_t)b * a.x_bstride;
  const size_t ob = (size_t)b * a.o_bstride;
  float* patch = xs + 2 * KC * XW + wave * (16 * CW);

  f32x4 wa[KS];
  {
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.wpack) + lane;
#pragma unroll
    for (int j = 0; j < KS; ++j) wa[j] = wp[j * 64];
  }
  const int r0 = tid / NV, v0 = tid - r0 * NV;
  const int dr = NT / NV, dv = NT - dr * NV;
  f32x4 sv[SV];
  auto stage_load = [&](int t0) {
    const int tb = (t0 - halo) & ~3;
    int r = r0, v = v0;
#pragma unroll
    for (int i = 0; i < SV; ++i) {
      int ci = r < KC ? r : KC - 1;
      ci = ci < a.CIN ? ci : a.CIN - 1;
      int t = tb + 4 * v;
      t = t < 0 ? 0 : (t > a.ldx - 4 ? a.ldx - 4 : t);
      sv[i] = *reinterpret_cast<const f32x4*>(xb + (size_t)ci * a.ldx + t);
      v += dv; r += dr;
      if (v >= NV) { v -= NV; ++r; }
    }
  };
  auto stage_store = [&](float* buf, int t0) {
    const int tb = (t0 - halo) & ~3;
    int r = r0, v = v0;
#pragma unroll
    for (int i = 0; i < SV; ++i) {
      if (r < KC) {
        const int t = tb + 4 * v;
        const bool rowok = r < a.CIN;
        f32x4 val = sv[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = rowok && (t + e) >= 0 && (t + e) < len;
          val[e] = ok ? lrelu16(val[e], slope) : 0.f;
        }
        *reinterpret_cast<f32x4*>(buf + r * XW + 4 * v) = val;
      }
      v += dv; r += dr;
      if (v >= NV) { v -= NV; ++r; }
    }
  };

  // epilogue geometry (constant per thread)
  const int prow = lane / LPR, pc4 = lane % LPR;
  const int epi = a.epi;
  const bool need_res = epi != EPI_STORE;

  stage_load(tile0 * BN);
  stage_store(xs, tile0 * BN);
  __syncthreads();

#pragma unroll 1
  for (int i = 0; i < ntile; ++i) {
    const int t0 = (tile0 + i) * BN;
    const bool more = i + 1 < ntile;
    if (more) stage_load(t0 + BN);
    // residual rows of this tile, fetched behind the MFMAs
    const int tcol = t0 + wave * (16 * NI) + 4 * pc4;
    f32x4 rv[NI];
    if (need_res) {
#pragma unroll
      for (int p = 0; p < NI; ++p) {
        const int row = p * RPP + prow;
        int tc = tcol > a.ldo - 4 ? a.ldo - 4 : tcol;
        rv[p] = *reinterpret_cast<const f32x4*>(a.res + ob + (size_t)(row < a.M ? row : 0) * a.ldo + tc);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- MFMAs: acc[ni] over KS taps x 4 k-steps -------------------------------------------
    const int sh = (t0 - halo) - ((t0 - halo) & ~3);
    const float* bj = xs + (i & 1) * (KC * XW) + g * XW + sh + wave * (16 * NI) + l15;
    f32x4 acc[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KS; ++j) {
#pragma unroll
      for (int cq = 0; cq < 4; ++cq)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j][cq], bj[cq * 4 * XW + ni * 16], acc[ni], 0, 0, 0);
      bj += a.dil;
    }
    // ---- epilogue: registers -> wave-private patch -> 16 B per lane ---------------------------
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) patch[(4 * g + r) * CW + ni * 16 + l15] = acc[ni][r];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < NI; ++p) {
      const int row = p * RPP + prow;
      f32x4 v = *reinterpret_cast<const f32x4*>(patch + row * CW + 4 * pc4);
      if (row >= a.M || tcol >= len) continue;
      const float bz = a.bias[row];
      v[0] += bz; v[1] += bz; v[2] += bz; v[3] += bz;
      const size_t idx = ob + (size_t)row * a.ldo + tcol;
      const int nv = len - tcol;
      if (nv >= 4) {
        if (epi == EPI_STORE) {
          *reinterpret_cast<f32x4*>(a.out + idx) = v;
        } else {
          v[0] += rv[p][0]; v[1] += rv[p][1]; v[2] += rv[p][2]; v[3] += rv[p][3];
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
    if (more) stage_store(xs + ((i + 1) & 1) * (KC * XW), t0 + BN);
    __syncthreads();
  }
}

int g_stream16 = 4;  // "stream16" option: time tiles per workgroup of conv16_stream_kernel (0 = off)

template <int KS>
static int launch_stream16(const ConvArgs& a, int B, int Lmax, hipStream_t stream) {
  constexpr int NI = 4, WN = 4, BN = 16 * NI * WN, CW = 16 * NI + 4;
  const int tpb = g_stream16;
  const int ntile = (Lmax + BN - 1) / BN;
  dim3 grid((ntile + tpb - 1) / tpb, 1, B);
  const size_t lds = ((size_t)2 * KC * a.XW + (size_t)WN * 16 * CW) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    DISSC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv16_stream_kernel<KS, NI, WN>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done = true;
  }
  hipLaunchKernelGGL((conv16_stream_kernel<KS, NI, WN>), grid, dim3(64 * WN), lds, stream, a, tpb);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

// returns 1 when the launch was taken by this kernel, 0 when the caller should use the general
// one, < 0 on error
int try_launch_conv16_stream(const ConvArgs& a, int B, int Lmax, int stride, hipStream_t stream) {
  if (g_stream16 <= 0 || a.m32 || stride != 1 || a.up != 1 || a.groups != 1 || a.nchunk != 1 || a.M > 16 ||
      a.CIN > 16 || a.scale || a.act != 0 || a.lengths_out || a.olen_default >= 0 ||
      (a.KS - 1) * a.dil > MAX_TAP_SPAN || a.pad_left != ((a.KS - 1) * a.dil) / 2 || conv_tile_bn(a.M) != 256)
    return 0;
  int rc;
  switch (a.KS) {
    case 3: rc = launch_stream16<3>(a, B, Lmax, stream); break;
    case 7: rc = launch_stream16<7>(a, B, Lmax, stream); break;
    case 11: rc = launch_stream16<11>(a, B, Lmax, stream); break;
    default: return 0;
  }
  return rc == DISSC_OK ? 1 : rc;
}

}  // namespace dissc
