// respair16_f23_kernel: the k = 11 residual pairs of the 16-channel stage as register-only Toom-Cook F(2,3) -- respair_f23.hip's
// scheme (four points held by one wave, B^T one addition per B operand, A^T three additions per output in registers, no V
// tiles, no exchange) on v_mfma_f32_16x16x4_f32: M = 16 rows, a k-step is 4 channels, the 16 channels of a conv are 4 k-steps.
// Both convs' transform-domain weights live in registers like respair16_kernel's taps do: 4 sub-filters x 4 points = 16 float4
// per conv (64 registers, loaded per conv) where the direct form holds 11; 8 products per output instead of 11.
// A wave's 4 column tiles x 16 columns = 64 output PAIRS = 128 outputs; geometry as in respair_f23.hip.
#include <string.h>

#include <vector>

#include "common.h"
#include "respair_f23.h"

namespace dissc {

template <int KS_, int DIL>
struct F23Geo16 {
  static constexpr int KS = KS_, NS = (KS_ + 2) / 3, C = 16, NW = 4;
  static constexpr int P2 = (KS - 1) / 2, P1 = P2 * DIL;
  static constexpr int D1 = DIL * NS, D2 = NS;
  static constexpr int NCOLS = 64 * NW;
  static constexpr int NU1 = NCOLS / D1, NC1 = NU1 * D1, W1 = 2 * NC1;
  static constexpr int NU2 = NCOLS / D2, NC2 = NU2 * D2, W2 = 2 * NC2;
  static constexpr int WOUT = ((W1 - 2 * P2) < W2 ? (W1 - 2 * P2) : W2) & ~3;
  static constexpr int REACH1 = (3 * NS - 1) * DIL, REACH2 = 3 * NS - 1;
  static constexpr int XW1 = f23_round32_16(3 + W1 + REACH1);
  static constexpr int XW2 = f23_round32_16(W2 + REACH2 + 1);
  static constexpr int XW = XW1 > XW2 ? XW1 : XW2;
  static constexpr int PW = 128 + 4;  // patch row: a wave's 128 outputs
  static_assert(NW * 16 * PW <= C * XW, "the epilogue patches fit the buffer");
  static_assert(W1 <= XW && W2 + REACH2 < XW, "T fits the buffer");
};

template <int KS_, int DIL>
__global__ void __launch_bounds__(256, 3) respair16_f23_kernel(const PairFArgs a) {
  using G = F23Geo16<KS_, DIL>;
  constexpr int C = G::C, NW = G::NW, NT = 64 * NW, NS = G::NS, P2 = G::P2, P1 = G::P1, D1 = G::D1, D2 = G::D2, XW = G::XW,
                W1 = G::W1, NC1 = G::NC1, NC2 = G::NC2, WOUT = G::WOUT, PW = G::PW, NI = 4;
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [C][XW]

  int b, len, o0;
  if (!f23_tile<WOUT>(a, gridDim.y, b, len, o0)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int tin0 = o0 - P2 - P1;
  const int tb = tin0 & ~3, sh = tin0 - tb;
  const float slope = a.slope;
  const float* xb = a.x + (size_t)b * a.bstride;

  {  // lrelu(x) on [tb, tb + XW) into LDS: every load first, then activation / zeros outside the utterance / stores
    constexpr int NV = XW / 4, NIT = (C * NV + NT - 1) / NT;
    f32x4 sv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * NT;
      const int r = i / NV < C ? i / NV : C - 1, v = i - (i / NV) * NV;
      const int t = tb + 4 * v;
      const int tc = t < 0 ? 0 : (t > a.ld - 4 ? a.ld - 4 : t);
      sv[it] = *reinterpret_cast<const f32x4*>(xb + (size_t)r * a.ld + tc);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * NT;
      if (i >= C * NV) continue;
      const int r = i / NV, v = i - r * NV;
      const int t = tb + 4 * v;
      f32x4 val = sv[it];
#pragma unroll
      for (int e = 0; e < 4; ++e) val[e] = ((t + e) >= 0 && (t + e) < len) ? (val[e] > 0.f ? val[e] : val[e] * slope) : 0.f;
      *reinterpret_cast<f32x4*>(xs + r * XW + 4 * v) = val;
    }
  }

  // this lane's four columns of each conv: column -> (unit tau, phase rho) -> first sample 2 D tau + rho
  int base1[NI], base2[NI];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int c = wave * 64 + ni * 16 + l15;
    const int c1 = c < NC1 ? c : NC1 - 1, c2 = c < NC2 ? c : NC2 - 1;
    base1[ni] = 2 * D1 * (c1 / D1) + (c1 % D1);
    base2[ni] = 2 * D2 * (c2 / D2) + (c2 % D2);
  }
  f32x4 acc[4][NI];
  auto clear = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[p][ni][e] = 0.f;
  };
  // one conv: 4 sub-filters x 4 k-steps x 4 column tiles x 4 points = 256 MFMAs fed by 256 fragment reads and 256 additions;
  // the conv's 16 weight fragments [sub-filter][point] (one float4 = the 4 k-steps) stay in registers
  auto taps = [&](const float* wq, const float* src, const int (&base)[NI], int off0, int dstep, int dunit) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t wr = wave_rsrc(wq, 0x7ffffff0u);  // scalar-base loads (common.h), constant offsets
    f32x4 u[NS][4];
#pragma unroll
    for (int j = 0; j < NS; ++j)
#pragma unroll
      for (int p = 0; p < 4; ++p) u[j][p] = rsrc_load16(wr, lane * 16u, (j * 4 + p) * 1024u);
#pragma unroll
    for (int j = 0; j < NS; ++j) {
#pragma unroll
      for (int cq = 0; cq < 4; ++cq) {
        const int row = 4 * cq + g;
        float bq[4][NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          const float* q = src + row * XW + off0 + base[ni] + j * dstep;
          const float x0 = q[0], x1 = q[dunit], x2 = q[2 * dunit], x3 = q[3 * dunit];
          bq[0][ni] = x0 - x2;
          bq[1][ni] = x1 + x2;
          bq[2][ni] = x2 - x1;
          bq[3][ni] = x1 - x3;
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni)
            acc[p][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[j][p][cq], bq[p][ni], acc[p][ni], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  __syncthreads();
  clear();
  if (!(a.dbg & 1)) taps(a.w1, xs, base1, sh, DIL, D1);

  // ---- T = lrelu(conv_d + b1) inside the utterance, 0 outside, into the same buffer.  D layout: col = lane & 15, row = 4 g + r ----
  __syncthreads();
  for (int i = tid; i < C * (XW - W1); i += NT) {
    const int r = i / (XW - W1), v = i - r * (XW - W1);
    xs[r * XW + W1 + v] = 0.f;
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int c = wave * 64 + ni * 16 + l15;
    if (c < NC1 && !(a.dbg & 2)) {
      const int pe = 2 * D1 * (c / D1) + (c % D1), po = pe + D1;
      const int te = o0 - P2 + pe, to = te + D1;
      const bool ine = te >= 0 && te < len, ino = to >= 0 && to < len;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 4 * g + r;
        const float bz = a.b1[row];
        const float y0 = acc[0][ni][r], y1 = acc[1][ni][r], y2 = acc[2][ni][r], y3 = acc[3][ni][r];
        float ve = (y0 + y1) + y2 + bz, vo = (y1 - y2) - y3 + bz;
        ve = ve > 0.f ? ve : ve * slope;
        vo = vo > 0.f ? vo : vo * slope;
        xs[row * XW + pe] = ine ? ve : 0.f;
        xs[row * XW + po] = ino ? vo : 0.f;
      }
    }
  }
  __syncthreads();
  clear();
  if (!(a.dbg & 1)) taps(a.w2, xs, base2, 0, 1, D2);

  // ---- epilogue: y = x + conv_1 + b2 (or an MRF mode) through a wave-private patch [16][PW] -> 16 B per lane ----
  __syncthreads();
  float* ep = xs + wave * (16 * PW);
  const int prow = lane >> 5, pc4 = lane & 31;
  const int ncol = wave * 128 + 4 * pc4;
  const int tcol = o0 + ncol;
  const size_t ob = (size_t)b * a.bstride;
  const bool live = ncol < WOUT && tcol < len;
  const int epi = a.epi;
  const bool rmw = epi != EPI_RES && epi != EPI_MRF_SET;
  if (a.dbg & 4) {
    if (acc[0][0][0] == 123.f) a.out[0] = 1.f;
    return;
  }
  f32x4 rv[8], pa[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int row = 2 * p + prow;
    const int tc = tcol > a.ld - 4 ? a.ld - 4 : tcol;
    rv[p] = *reinterpret_cast<const f32x4*>(a.x + ob + (size_t)row * a.ld + tc);
    if (rmw && live && tcol + 4 <= len) pa[p] = *reinterpret_cast<const f32x4*>(a.acc + ob + (size_t)row * a.ld + tcol);
  }
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int cl = ni * 16 + l15;
    const int pe = 2 * D2 * (cl / D2) + (cl % D2);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float y0 = acc[0][ni][r], y1 = acc[1][ni][r], y2 = acc[2][ni][r], y3 = acc[3][ni][r];
      ep[(4 * g + r) * PW + pe] = (y0 + y1) + y2;
      ep[(4 * g + r) * PW + pe + D2] = (y1 - y2) - y3;
    }
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int row = 2 * p + prow;
    f32x4 v = *reinterpret_cast<const f32x4*>(ep + row * PW + 4 * pc4);
    if (!live) continue;
    const float bz = a.b2[row];
    const size_t idx = ob + (size_t)row * a.ld + tcol;
    const int nv = len - tcol;
    if (nv >= 4) {
      const f32x4 r4 = rv[p];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (v[e] + bz) + r4[e];
      if (epi == EPI_RES) {
        *reinterpret_cast<f32x4*>(a.out + idx) = v;
      } else if (epi == EPI_MRF_SET) {
        *reinterpret_cast<f32x4*>(a.acc + idx) = v;
      } else {
        const f32x4 ac = pa[p];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = ac[e] + v[e];
          if (epi == EPI_MRF_DIV) v[e] = __fdiv_rn(v[e], a.mrf_div);
        }
        *reinterpret_cast<f32x4*>(a.acc + idx) = v;
      }
    } else {
      for (int e = 0; e < nv; ++e) {
        float x = (v[e] + bz) + a.x[idx + e];
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

// w: [16][16][11] -> U_p[co][ci][j] = sum_i G[p][i] w[co][ci][j + 4 i] in A-fragment order [sub-filter][point][lane][k-step]:
// lane l, k-step cq -> U_p[co = l & 15][ci = 4 cq + (l >> 4)][j]
int pack_pair16_f23(const float* w, float** dev, int KS) {
  static const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  constexpr int C = 16;
  const int NS = (KS + 2) / 3;
  std::vector<float> packed((size_t)NS * 4 * 64 * 4);
  size_t o = 0;
  for (int j = 0; j < NS; ++j)
    for (int p = 0; p < 4; ++p)
      for (int lane = 0; lane < 64; ++lane)
        for (int cq = 0; cq < 4; ++cq) {
          const int co = lane & 15, ci = 4 * cq + (lane >> 4);
          double u = 0.0;
          for (int i = 0; i < 3; ++i) {
            const int tap = j + NS * i;
            if (tap < KS) u += G[p][i] * (double)w[((size_t)co * C + ci) * KS + tap];
          }
          packed[o++] = (float)u;
        }
  return upload(packed, dev);
}

template <int KS_, int DIL>
static int launch_f23_16_t(const PairFArgs& a, int B, int Lmax, hipStream_t stream) {
  using G = F23Geo16<KS_, DIL>;
  dim3 grid((Lmax + G::WOUT - 1) / G::WOUT, B);
  hipLaunchKernelGGL((respair16_f23_kernel<KS_, DIL>), grid, dim3(256), sizeof(float) * G::C * G::XW, stream, a);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

int launch_pair16_f23(const PairFArgs& a, int KS, int dil, int B, int Lmax, hipStream_t stream) {
#define DISSC_F23(K_, D_) \
  if (KS == K_ && dil == D_) return launch_f23_16_t<K_, D_>(a, B, Lmax, stream);
  DISSC_F23(11, 1) DISSC_F23(11, 3) DISSC_F23(11, 5)
#if DISSC_EXPERIMENTAL  // k = 3 through these kernels measured neutral in the forward (NOTES round 4): not in the default build
  DISSC_F23(3, 1) DISSC_F23(3, 3) DISSC_F23(3, 5)
#endif
#undef DISSC_F23
  set_error("launch_pair16_f23: no instance for k = %d, dilation %d", KS, dil);
  return DISSC_EINVAL;
}

}  // namespace dissc
