// Dilated "same" Conv1d / phase-decomposed ConvTranspose1d as an implicit GEMM on the
// gfx950 fp32 matrix pipe (v_mfma_f32_16x16x4_f32: exact f32, 157 TFLOP/s peak).
//
//   D[row, t] = bias[row] + sum_{ci, j} W[row, ci, j] * act(x[ci, t + (j - (KS-1)/2) * dil])
//
// GEMM view: M = output rows (Cout, or Cout*stride for ConvTranspose), N = time,
// K = Cin*KS.  One workgroup owns a BM x BN output tile of ONE utterance:
//   - the input window [KC channels][BN + (KS-1)*dil] is staged through LDS once per
//     16-channel chunk (coalesced along time, leaky-ReLU and the utterance's zero
//     padding applied on the way in), and is shared by all waves / all taps;
//   - the weight (A) fragments are pre-packed on the host in exactly the MFMA lane
//     order, so every wave streams them L2 -> VGPR with one coalesced dwordx4 per
//     lane per (16 rows x 16 channels x 1 tap), prefetched one tap ahead;
//   - B fragments are ds_read_b32 (lanes 0-15 / 16-31 hit rows XW apart, XW%32==16
//     => conflict free).
// Replaces the MIOpen/cuDNN conv1d / conv_transpose1d calls behind reference
// sr/models.py:34-41 (ResBlock1), :99-102 (conv_pre, ups).
#include "common.h"

namespace dissc {

template <int MI, int NI, int WM, int WN>
__global__ void __launch_bounds__(64 * WM * WN) conv_mfma_kernel(const ConvArgs a) {
  constexpr int NT = 64 * WM * WN;
  constexpr int BN = 16 * NI * WN;
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [KC][XW]

  const int b = blockIdx.z;
  const int len = (a.lengths ? a.lengths[b] * a.len_mul : a.len_default);
  const int t0 = blockIdx.x * BN;
  if (t0 >= len) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int l15 = lane & 15, g = lane >> 4;
  const int XW = a.XW;
  const int halo = ((a.KS - 1) * a.dil) >> 1;
  const int span = BN + (a.KS - 1) * a.dil;
  const int nq = a.nchunk * a.KS;
  const int ms0 = blockIdx.y * (MI * WM) + wm * MI;

  f32x4 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

  const f32x4* wp[MI];
  f32x4 av[MI], avn[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    wp[mi] = reinterpret_cast<const f32x4*>(a.wpack) + (size_t)(ms0 + mi) * nq * 64 + lane;
    av[mi] = wp[mi][0];
  }

  const float* xb = a.x + (size_t)b * a.x_bstride;
  const float slope = a.slope;
  const float* bbase = xs + g * XW + wn * (16 * NI) + l15;

  int q = 0;
  for (int c = 0; c < a.nchunk; ++c) {
    __syncthreads();  // everyone is done reading the previous chunk
    for (int r = wave; r < KC; r += NT / 64) {
      const int ci = c * KC + r;
      const bool crow = ci < a.CIN;
      const float* row = xb + (size_t)ci * a.ldx;
      float* dst = xs + r * XW;
      for (int u = lane; u < span; u += 64) {
        const int t = t0 - halo + u;
        float v = 0.f;
        if (crow && t >= 0 && t < len) {
          v = row[t];
          v = v > 0.f ? v : v * slope;
        }
        dst[u] = v;
      }
    }
    __syncthreads();

    for (int j = 0; j < a.KS; ++j, ++q) {
      const int qn = (q + 1 < nq) ? q + 1 : q;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) avn[mi] = wp[mi][(size_t)qn * 64];
      const float* bj = bbase + j * a.dil;
#pragma unroll
      for (int cq = 0; cq < 4; ++cq) {
        float bv[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) bv[ni] = bj[cq * 4 * XW + ni * 16];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mi][cq], bv[ni], acc[mi][ni], 0, 0, 0);
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) av[mi] = avn[mi];
    }
  }

  // Epilogue.  C/D layout of 16x16x4: col = lane & 15 (time), row = 4*(lane>>4) + r.
  const int tbase = t0 + wn * (16 * NI) + l15;
  const size_t boff = (size_t)b * a.o_bstride;
  const int epi = a.epi;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = (ms0 + mi) * 16 + g * 4 + r;
      if (row >= a.M) continue;
      const float bz = a.bias[row];
      size_t rowoff;
      int tmul = 1, tadd = 0;
      if (a.up == 1) {
        rowoff = boff + (size_t)row * a.ldo;
      } else {
        const int co = row / a.up;
        tadd = row - co * a.up;
        tmul = a.up;
        rowoff = boff + (size_t)co * a.ldo;
      }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const int t = tbase + ni * 16;
        if (t >= len) continue;
        const float v = acc[mi][ni][r] + bz;
        const size_t idx = rowoff + (size_t)t * tmul + tadd;
        if (epi == EPI_STORE) {
          a.out[idx] = v;
        } else if (epi == EPI_RES) {
          a.out[idx] = v + a.res[idx];
        } else {
          const float rr = v + a.res[idx];
          if (epi == EPI_MRF_SET) {
            a.acc[idx] = rr;
          } else if (epi == EPI_MRF_ADD) {
            a.acc[idx] = a.acc[idx] + rr;
          } else {
            a.acc[idx] = __fdiv_rn(a.acc[idx] + rr, a.mrf_div);
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
  if (M >= 256) return 256;
  if (M >= 128) return 128;
  if (M >= 64) return 64;
  if (M >= 32) return 32;
  return 16;
}

int conv_tile_bn(int M) {
  switch (pick_bm(M)) {
    case 256: return 64;
    case 128: return 128;
    case 64: return 256;
    default: return 512;
  }
}

int conv_xw(int M, int KS, int dil) {
  const int span = conv_tile_bn(M) + (KS - 1) * dil;
  int xw = (span + 31) / 32 * 32 + 16;
  if (xw - 32 >= span) xw -= 32;
  return xw;
}

void pack_conv_weights(const float* w, int Cout, int Cin, int KS, std::vector<float>& packed,
                       int& Mpad, int& nchunk) {
  const int bm = pick_bm(Cout);
  Mpad = (Cout + bm - 1) / bm * bm;
  nchunk = (Cin + KC - 1) / KC;
  const int nsub = Mpad / 16;
  packed.assign((size_t)nsub * nchunk * KS * 64 * 4, 0.f);
  for (int ms = 0; ms < nsub; ++ms)
    for (int c = 0; c < nchunk; ++c)
      for (int j = 0; j < KS; ++j)
        for (int lane = 0; lane < 64; ++lane)
          for (int cq = 0; cq < 4; ++cq) {
            const int co = ms * 16 + (lane & 15);
            const int ci = c * KC + cq * 4 + (lane >> 4);
            if (co < Cout && ci < Cin)
              packed[((((size_t)ms * nchunk + c) * KS + j) * 64 + lane) * 4 + cq] =
                  w[((size_t)co * Cin + ci) * KS + j];
          }
}

void convT_to_conv(const float* w, int Cin, int Cout, int k, int s, std::vector<float>& w3) {
  const int pad = (k - s) / 2;
  const int M = Cout * s;
  w3.assign((size_t)M * Cin * 3, 0.f);
  for (int co = 0; co < Cout; ++co)
    for (int p = 0; p < s; ++p)
      for (int ci = 0; ci < Cin; ++ci)
        for (int j = 0; j < 3; ++j) {
          const int kk = p + pad - s * (j - 1);
          if (kk >= 0 && kk < k)
            w3[((size_t)(co * s + p) * Cin + ci) * 3 + j] = w[((size_t)ci * Cout + co) * k + kk];
        }
}

template <int MI, int NI, int WM, int WN>
static int launch_t(const ConvArgs& a, int B, int Lmax, hipStream_t stream) {
  constexpr int BM = 16 * MI * WM, BN = 16 * NI * WN;
  const int Mpad = (a.M + BM - 1) / BM * BM;
  dim3 grid((Lmax + BN - 1) / BN, Mpad / BM, B);
  const size_t lds = (size_t)KC * a.XW * sizeof(float);
  if (lds > 65536) {
    static bool done = false;
    if (!done) {
      DISSC_HIP_CHECK(hipFuncSetAttribute(
          reinterpret_cast<const void*>(&conv_mfma_kernel<MI, NI, WM, WN>),
          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      done = true;
    }
  }
  hipLaunchKernelGGL((conv_mfma_kernel<MI, NI, WM, WN>), grid, dim3(64 * WM * WN), lds, stream, a);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

int launch_conv(const ConvArgs& a, int B, int Lmax, hipStream_t stream) {
  if (B <= 0 || Lmax <= 0) return DISSC_OK;
  if (B > 65535) {
    set_error("launch_conv: batch %d exceeds grid.z limit", B);
    return DISSC_EINVAL;
  }
  switch (pick_bm(a.M)) {
    case 256: return launch_t<4, 4, 4, 1>(a, B, Lmax, stream);
    case 128: return launch_t<4, 4, 2, 2>(a, B, Lmax, stream);
    case 64: return launch_t<4, 4, 1, 4>(a, B, Lmax, stream);
    case 32: return launch_t<2, 8, 1, 4>(a, B, Lmax, stream);
    default: return launch_t<1, 8, 1, 4>(a, B, Lmax, stream);
  }
}

}  // namespace dissc
