// respair32_f23_kernel: one residual pair y = x + conv_1(lrelu(conv_d(lrelu(x)))) of the 32-channel stage, k = 11 (reference
// sr/models.py:34-41), with BOTH convs in the Toom-Cook F(2,3) transform domain and NOTHING exchanged between waves.
//
// Why F(2,3) here when the wider stages use six and eight points (conv_wino.hip, conv_wino8.hip): what makes the transform
// domain lose at C <= 32 is not its arithmetic but the traffic around it -- V tiles written and re-read through LDS, Y_p
// exchanged between the waves that own the points.  With FOUR points (0, 1, -1, inf) one wave can hold all of them for its
// own columns (4 points x 2 column tiles x 16 accumulators = 128 registers at M = 32 rows), so
//   * B^T (entries 0 / +-1) is one subtraction or addition per B operand on the way from LDS to the MFMA:
//       b0 = x0 - x2, b1 = x1 + x2, b2 = x2 - x1, b3 = x1 - x3           (x_q = window sample at + q D)
//   * A^T is three additions per output, in registers: y_even = Y0 + Y1 + Y2, y_odd = Y1 - Y2 - Y3
//   * G (host): U_p = G w per triple of taps, rows (1, 0, 0), (1/2, 1/2, 1/2), (1/2, -1/2, 1/2), (0, 0, 1)
// and the kernel keeps the shape of respair32_kernel (respair.hip): window of lrelu(x) in LDS, conv_d, T = lrelu(. + b1) back
// into the same LDS, conv_1, wave-private epilogue patches.  The k = 11 taps are four sub-filters of three taps with tap
// stride NS = 4 (tap 11 is a zero: its samples still enter the transforms, so the window holds real data there); an MFMA
// column is an output PAIR (t, t + D), D = d NS: 4 products per 2 outputs and sub-filter = 8 per output instead of 11, and 4
// LDS fragment reads where the direct form makes 6.  A wave's 64 columns are 128 outputs, the workgroup's 256 columns 512
// (480 / 504 for d = 5 / 3, whose units of 2 D outputs do not divide 512).
// fp32 operands, fp32 products, fp32 accumulation on v_mfma_f32_32x32x2_f32; not bit-identical to the direct pair.
#include <string.h>

#include <vector>

#include "common.h"
#include "respair_f23.h"

namespace dissc {

// option "pair_f23" (Options::pair_f23, default 3): "pair_f23" option (read at dissc_gen_create), a bit mask: 1 = the C = 32, k = 11 pairs run on this kernel -- per launch
                     // 857 / 894 / 924 us at d = 1 / 3 / 5 against 1 042 / 1 037 / 1 052 for the direct pair (B = 32 x 10 s) --, 2 = the
                     // C = 16, k = 11 pairs on respair16_f23.hip (525 against 604 us at d = 1); default both.  4 / 8 = the k = 3 pairs of
                     // the two stages (C = 32: 365 against 417 us, C = 16: 262 against 263; forward 33.11 -> 33.09 ms: off)

template <int KS_, int DIL>
struct F23Geo {
  static constexpr int KS = KS_, NS = (KS_ + 2) / 3, C = 32, NW = 4;
  static constexpr int P2 = (KS - 1) / 2, P1 = P2 * DIL;
  static constexpr int D1 = DIL * NS, D2 = NS;
  static constexpr int NCOLS = 64 * NW;                               // pair-columns per conv and workgroup
  static constexpr int NU1 = NCOLS / D1, NC1 = NU1 * D1, W1 = 2 * NC1;  // positions of T conv_d produces: [o0 - P2, o0 - P2 + W1)
  static constexpr int NU2 = NCOLS / D2, NC2 = NU2 * D2, W2 = 2 * NC2;  // outputs conv_1 computes: [o0, o0 + W2)
  static constexpr int WOUT = ((W1 - 2 * P2) < W2 ? (W1 - 2 * P2) : W2) & ~3;  // outputs a workgroup owns
  static constexpr int REACH1 = (3 * NS - 1) * DIL, REACH2 = 3 * NS - 1;       // samples read beyond the last column's first
  static constexpr int XW1 = f23_round32_16(3 + W1 + REACH1);
  static constexpr int XW2 = f23_round32_16(W2 + REACH2 + 1);
  static constexpr int XW = XW1 > XW2 ? XW1 : XW2;  // row stride of the one LDS buffer (x window, then T, then the patches)
  static constexpr int PW = 128 + 4;                // patch row: a wave's 128 outputs
  static_assert(NW * 8 * PW <= C * XW, "the epilogue patches fit the buffer");
  static_assert(W1 <= XW && W2 + REACH2 < XW, "T fits the buffer");
};

template <int KS_, int DIL>
__global__ void __launch_bounds__(256, 2) respair32_f23_kernel(const PairFArgs a) {
  using G = F23Geo<KS_, DIL>;
  constexpr int C = G::C, NW = G::NW, NT = 64 * NW, NS = G::NS, P2 = G::P2, P1 = G::P1, D1 = G::D1, D2 = G::D2, XW = G::XW,
                W1 = G::W1, NC1 = G::NC1, NC2 = G::NC2, WOUT = G::WOUT, PW = G::PW;
  extern __shared__ __attribute__((aligned(16))) float xs[];  // [C][XW]

  int b, len, o0;
  if (!f23_tile<WOUT>(a, gridDim.y, b, len, o0)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int tin0 = o0 - P2 - P1;
  const int tb = tin0 & ~3, sh = tin0 - tb;
  const float slope = a.slope;
  const float* xb = a.x + (size_t)b * a.bstride;

  // ---- lrelu(x) on [tb, tb + XW) into LDS: every 16-byte load of the thread first (clamped, unconditional), then activation,
  // zeros outside the utterance and the stores (loads issued one per loop trip would be as many dependent round trips) ----
  {
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
      // (t is a multiple of 4 and so is ld: a clamped quad lies wholly outside [0, len) and is zeroed here)
      f32x4 val = sv[it];
#pragma unroll
      for (int e = 0; e < 4; ++e) val[e] = ((t + e) >= 0 && (t + e) < len) ? (val[e] > 0.f ? val[e] : val[e] * slope) : 0.f;
      *reinterpret_cast<f32x4*>(xs + r * XW + 4 * v) = val;
    }
  }

  // this lane's two columns of each conv: column -> (unit tau, phase rho) -> first sample 2 D tau + rho
  int base1[2], base2[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    int c = wave * 64 + ni * 32 + l31;
    const int c1 = c < NC1 ? c : NC1 - 1, c2 = c < NC2 ? c : NC2 - 1;
    base1[ni] = 2 * D1 * (c1 / D1) + (c1 % D1);
    base2[ni] = 2 * D2 * (c2 / D2) + (c2 % D2);
  }
  typedef float f32x16f __attribute__((ext_vector_type(16)));
  f32x16f acc[4][2];
  auto clear = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[p][ni][e] = 0.f;
  };
  // one conv: chunks of 16 channels x 4 sub-filters; per (chunk, sub-filter) 8 k-steps x 2 column tiles x 4 points = 64 MFMAs fed
  // by 64 fragment reads and 64 additions
  auto taps = [&](const float* wq, const float* src, const int (&base)[2], int off0, int dstep, int dunit) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t wr = wave_rsrc(wq, 0x7ffffff0u);  // scalar-base loads (common.h), constant offsets
    const unsigned lane16 = lane * 16u;
    f32x4 av[4][2], avn[4][2];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      av[p][0] = rsrc_load16(wr, lane16, (p * 2 + 0) * 1024u);
      av[p][1] = rsrc_load16(wr, lane16, (p * 2 + 1) * 1024u);
    }
#pragma unroll
    for (int s = 0; s < 2 * NS; ++s) {  // s = chunk * NS + sub-filter
      const int sn = s + 1 < 2 * NS ? s + 1 : s;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        avn[p][0] = rsrc_load16(wr, lane16, ((sn * 4 + p) * 2 + 0) * 1024u);
        avn[p][1] = rsrc_load16(wr, lane16, ((sn * 4 + p) * 2 + 1) * 1024u);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int chunk = s / NS, j = s % NS;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 16 * chunk + 2 * (4 * hf + e) + h;
          float bq[4][2];
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) {
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
            for (int ni = 0; ni < 2; ++ni)
              acc[p][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[p][hf][e], bq[p][ni], acc[p][ni], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        av[p][0] = avn[p][0];
        av[p][1] = avn[p][1];
      }
    }
  };

  __syncthreads();
  clear();
  if (!(a.dbg & 1)) taps(a.w1, xs, base1, sh, DIL, D1);

  // ---- T = lrelu(conv_d + b1) inside the utterance, 0 outside, into the same buffer: positions [0, W1) <-> times o0 - P2 + . ----
  __syncthreads();  // every wave is done reading the x window
  for (int i = tid; i < C * (XW - W1); i += NT) {
    const int r = i / (XW - W1), v = i - r * (XW - W1);
    xs[r * XW + W1 + v] = 0.f;  // what conv_1's last columns read beyond T
  }
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int c = wave * 64 + ni * 32 + l31;
    if (c < NC1 && !(a.dbg & 2)) {
      const int pe = 2 * D1 * (c / D1) + (c % D1), po = pe + D1;
      const int te = o0 - P2 + pe, to = te + D1;
      const bool ine = te >= 0 && te < len, ino = to >= 0 && to < len;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
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

  // ---- epilogue: y = x + conv_1 + b2 (or an MRF mode), 8 rows at a time through a wave-private patch [8][PW] ----
  __syncthreads();  // every wave is done reading T, which the patches overwrite
  float* ep = xs + wave * (8 * PW);
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
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {  // rows 8 qd .. 8 qd + 7
    f32x4 rv[4], pa[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = 8 * qd + 2 * p + prow;
      const int tc = tcol > a.ld - 4 ? a.ld - 4 : tcol;
      rv[p] = *reinterpret_cast<const f32x4*>(a.x + ob + (size_t)row * a.ld + tc);
      if (rmw && live && tcol + 4 <= len) pa[p] = *reinterpret_cast<const f32x4*>(a.acc + ob + (size_t)row * a.ld + tcol);
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int cl = ni * 32 + l31;
      const int pe = 2 * D2 * (cl / D2) + (cl % D2);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int rr = 4 * qd + r;
        const float y0 = acc[0][ni][rr], y1 = acc[1][ni][rr], y2 = acc[2][ni][rr], y3 = acc[3][ni][rr];
        ep[(r + 4 * h) * PW + pe] = (y0 + y1) + y2;
        ep[(r + 4 * h) * PW + pe + D2] = (y1 - y2) - y3;
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int prw = 2 * p + prow;
      f32x4 v = *reinterpret_cast<const f32x4*>(ep + prw * PW + 4 * pc4);
      if (!live) continue;
      const int row = 8 * qd + prw;
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
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// "pair_f23" is a bit mask: 1 = the 32-channel stage (this file), 2 = the 16-channel stage (respair16_f23.hip)
// (bits 2 / 3: the k = 3 pairs of the two stages -- one sub-filter, 2 products per output instead of 3)
bool pair_f23_supported(int C, int KS, int dil) {
  if (!((KS == 11 || (KS == 3 && DISSC_EXPERIMENTAL)) && (dil == 1 || dil == 3 || dil == 5))) return false;
  const int sh = KS == 3 ? 2 : 0;
  return (C == 32 && (opts().pair_f23 & (1 << sh))) || (C == 16 && (opts().pair_f23 & (2 << sh)));
}

// w: [32][32][11] -> U_p[co][ci][j] = sum_i G[p][i] w[co][ci][j + 4 i] in A-fragment order [chunk][sub-filter][point][half][lane][4]
int pack_pair_f23(const float* w, float** dev, int C_, int KS) {
  if (C_ == 16) return pack_pair16_f23(w, dev, KS);
  static const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
  constexpr int C = 32;
  const int NS = (KS + 2) / 3;
  std::vector<float> packed((size_t)2 * NS * 4 * 2 * 64 * 4);
  size_t o = 0;
  for (int c = 0; c < 2; ++c)
    for (int j = 0; j < NS; ++j)
      for (int p = 0; p < 4; ++p)
        for (int hf = 0; hf < 2; ++hf)
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 4; ++e) {
              const int co = lane & 31, ci = 16 * c + 2 * (4 * hf + e) + (lane >> 5);
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
static int launch_f23_t(const PairFArgs& a, int B, int Lmax, hipStream_t stream) {
  using G = F23Geo<KS_, DIL>;
  static DeviceOnce attr_once;  // per device (common.h)
  DISSC_HIP_CHECK(attr_once.max_lds(reinterpret_cast<const void*>(&respair32_f23_kernel<KS_, DIL>), 160 * 1024));
  dim3 grid((Lmax + G::WOUT - 1) / G::WOUT, B);
  hipLaunchKernelGGL((respair32_f23_kernel<KS_, DIL>), grid, dim3(256), sizeof(float) * G::C * G::XW, stream, a);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

int launch_pair_f23(const DevPairW& pw, const float* x, float* out, float* acc, const int32_t* lengths, int len_default,
                    int len_mul, int B, int Lmax, int ld, float slope, int epi, float mrf_div, hipStream_t stream) {
  PairFArgs a;
  a.x = x; a.out = out; a.acc = acc; a.w1 = pw.w1; a.w2 = pw.w2; a.b1 = pw.b1; a.b2 = pw.b2;
  a.lengths = lengths; a.len_default = len_default; a.len_mul = len_mul; a.ld = ld;
  a.bstride = (long long)pw.C * ld; a.slope = slope; a.mrf_div = mrf_div; a.epi = epi; a.dbg = opts().kernel_dbg;
  if (pw.C == 16) return launch_pair16_f23(a, pw.KS, pw.dil, B, Lmax, stream);
#define DISSC_F23(K_, D_) \
  if (pw.KS == K_ && pw.dil == D_) return launch_f23_t<K_, D_>(a, B, Lmax, stream);
  DISSC_F23(11, 1) DISSC_F23(11, 3) DISSC_F23(11, 5)
#if DISSC_EXPERIMENTAL  // k = 3 through these kernels measured neutral in the forward (NOTES round 4): not in the default build
  DISSC_F23(3, 1) DISSC_F23(3, 3) DISSC_F23(3, 5)
#endif
#undef DISSC_F23
  set_error("launch_pair_f23: no instance for k = %d, dilation %d", pw.KS, pw.dil);
  return DISSC_EINVAL;
}

}  // namespace dissc
