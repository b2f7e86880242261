// Epilogue shared by the 32-row-subtile conv kernels (conv_mfma32.hip, conv_bf3.hip): accumulators in
// the 32x32 MFMA C/D layout -> bias / affine / activation / residual / MRF update -> HBM.
#pragma once
#include "common.h"

namespace dissc {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }
__device__ __forceinline__ float gelu_exact(float v) { return 0.5f * v * (1.f + erf_1ulp(v * 0.70710678118654752440f)); }

// EPI_STORE_ACT: zeros for the columns tcol .. tcol+3 of one row that lie in [olen, olen + ZERO_TAIL) or [ldo - ZERO_TAIL, ldo)
__device__ __forceinline__ void zero_tail4(const ConvArgs& a, size_t rowoff, int tcol, int olen) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int t = tcol + e;
    if (t >= olen && t < a.ldo && (t < olen + ZERO_TAIL || t >= a.ldo - ZERO_TAIL)) a.out[rowoff + t] = 0.f;
  }
}

// a whole tile beyond the utterance (t0 >= olen): only its part of the zero tails is written
template <int NT, int BN>
__device__ __forceinline__ void zero_tail_tile(const ConvArgs& a, int b, int t0, int olen) {
  if (!(t0 < olen + ZERO_TAIL || t0 + BN > a.ldo - ZERO_TAIL) || t0 >= a.ldo) return;
  const size_t ob = (size_t)b * a.o_bstride;
  for (int row = threadIdx.x; row < a.M * a.groups; row += NT)
    for (int t = t0; t < t0 + BN && t < a.ldo; ++t)
      if (t < olen + ZERO_TAIL || t >= a.ldo - ZERO_TAIL) a.out[ob + (size_t)row * a.ldo + t] = 0.f;
}

// acc[mi][ni]: rows (ms0 + mi) * 32 .. +31 of group grp, columns t0 + wn * 32 * NI + ni * 32 .. +31.
// xs: the workgroup's LDS (free at this point): one [8][32 * NI + 4] patch per wave.  All lanes of
// the calling wave must take part; waves are told apart by threadIdx.x >> 6.
template <int MI, int NI>
__device__ __forceinline__ void conv_epilogue32(const ConvArgs& a, f32x16 (&acc)[MI][NI], float* xs, int b,
                                                int t0, int olen, int grp, int ms0, int wn) {
  constexpr int CW = 32 * NI + 4;  // patch row stride
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int epi = a.epi;
  const size_t ob = (size_t)b * a.o_bstride;
  // C/D layout of 32x32x2: col = lane & 31 (time), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
  if (a.up != 1) {
    // ConvTranspose pixel shuffle straight from registers: row = co*np + pi -> out[co][t*up + p0 + pi]
    const int tbase = t0 + wn * (32 * NI) + l31;
    if ((a.up_np & 1) == 0 && (a.up & 1) == 0 && (a.up_p0 & 1) == 0 && (a.M & 1) == 0) {
      // two adjacent phases of one output channel sit in registers r, r+1 of the same lane (rows 2i, 2i+1):
      // one 8-byte store per pair, contiguous across the wave's lanes when up == 2
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const int row = (ms0 + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;  // even
          if (row >= a.M) continue;
          const float bz0 = a.bias[row], bz1 = a.bias[row + 1];
          const int co = row / a.up_np;
          const int p = a.up_p0 + row - co * a.up_np;
          const size_t rowoff = ob + (size_t)co * a.ldo + p;
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const int t = tbase + ni * 32;
            if (t < olen)
              *reinterpret_cast<float2*>(a.out + rowoff + (size_t)t * a.up) =
                  make_float2(acc[mi][ni][r] + bz0, acc[mi][ni][r + 1] + bz1);
          }
        }
      }
      return;
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (ms0 + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row >= a.M) continue;
        const float bz = a.bias[row];  // ConvTranspose path: groups == 1
        const int co = row / a.up_np;
        const int p = a.up_p0 + row - co * a.up_np;
        const size_t rowoff = ob + (size_t)co * a.ldo + p;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const int t = tbase + ni * 32;
          if (t < olen) a.out[rowoff + (size_t)t * a.up] = acc[mi][ni][r] + bz;
        }
      }
    }
    return;
  }

  // Transposed epilogue: 8 rows at a time through a wave-private LDS patch [8][CW] -> 16 B per lane.
  float* ep = xs + wave * (8 * CW);
  constexpr int LPR = 8 * NI;        // float4 lanes per output row (32*NI columns)
  constexpr int RPP = 64 / LPR;      // rows per pass
  constexpr int NPASS = 8 / RPP;
  const int prow = lane / LPR, pc4 = lane % LPR;
  const int tcol = t0 + wn * (32 * NI) + 4 * pc4;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {  // rows 8*qd .. 8*qd+7 of the 32-row subtile
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) ep[(4 * h + r) * CW + ni * 32 + l31] = acc[mi][ni][qd * 4 + r];
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        const int rl = p * RPP + prow;
        const int row = (ms0 + mi) * 32 + qd * 8 + rl;  // within the group
        f32x4 v = *reinterpret_cast<const f32x4*>(ep + rl * CW + 4 * pc4);
        if (row >= a.M) continue;
        if (tcol >= olen) {
          if (epi == EPI_STORE_ACT) zero_tail4(a, ob + (size_t)(grp * a.M + row) * a.ldo, tcol, olen);
          continue;
        }
        const int prow_idx = grp * a.nsub_group * 32 + row;  // bias/scale/shift are [groups][Mpad]
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
        if (epi == EPI_STORE_ACT) {
          const float sl = a.out_slope;
          v[0] = lrelu(v[0], sl); v[1] = lrelu(v[1], sl); v[2] = lrelu(v[2], sl); v[3] = lrelu(v[3], sl);
          if (nv >= 4) {
            *reinterpret_cast<f32x4*>(a.out + idx) = v;
          } else {
            for (int e = 0; e < nv; ++e) a.out[idx + e] = v[e];
            zero_tail4(a, idx - tcol, tcol, olen);  // the rest of this float4 lies in the zero tail
          }
          continue;
        }
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
}

}  // namespace dissc
