// YAAPT's sequential stages on the device: one workgroup per utterance, fp64 like the numpy restatement
// (oracle/yaapt_ref.py: spec_track / merge / dynamic-programming passes; parity with amfm_decompy unpinned, see there).
//   yaapt_spec_track_kernel : NLFER normalisation + voicing, SHC-candidate smoothing (median), Viterbi over the 4 SHC
//                             candidates, pchip interpolation over the unvoiced gaps, 3-tap mean, NCCF lag ranges
//   yaapt_final_track_kernel: merit re-weighting + stable sort of the 6 NCCF candidates, median-smoothed "best" track,
//                             the 8-row candidate table, Viterbi with voiced/unvoiced transition costs -> F0
// The frame loops of the two Viterbi passes are inherently serial (a 4x4 / 8x8 min-plus step per frame); one wave
// walks them with the candidate axis on the lanes and the back-pointers in LDS.  32 utterances = 32 CUs busy for
// ~1 ms -- against 0.40 s of per-utterance numpy loops on the host before.
#include "common.h"

namespace dissc {

namespace {

constexpr int DP_NT = 256;
constexpr int MAXMED = 9;  // largest median window

struct TrackCfg {
  double nlfer_thresh1, nlfer_thresh2, dp5_k1, merit_boost, merit_pivot, merit_extra, dp_w1, dp_w2, dp_w3, dp_w4,
      min_std, f0_min, f0_max;
  int median_value, nccf_pwidth, fs;
};

__device__ inline double block_sum(double v, double* red) {
  // fixed-order tree over the 256 threads
  const int tid = threadIdx.x;
  __syncthreads();
  red[tid] = v;
  __syncthreads();
  for (int s = DP_NT / 2; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

// exclusive scan of per-thread counts; returns this thread's offset, *total = sum
__device__ inline int block_scan(int v, int* sc, int* total) {
  const int tid = threadIdx.x;
  __syncthreads();
  sc[tid] = v;
  __syncthreads();
  for (int off = 1; off < DP_NT; off <<= 1) {
    const int t = tid >= off ? sc[tid - off] : 0;
    __syncthreads();
    sc[tid] += t;
    __syncthreads();
  }
  const int incl = sc[tid];
  *total = sc[DP_NT - 1];
  __syncthreads();
  return incl - v;
}

// scipy.signal.medfilt: odd window, zeros beyond the ends
__device__ inline double median_at(const double* a, int n, int c, int ks) {
  double w[MAXMED];
  const int h = ks / 2;
  for (int j = 0; j < ks; ++j) {
    const int i = c - h + j;
    const double v = (i >= 0 && i < n) ? a[i] : 0.0;
    int k = j;
    while (k > 0 && w[k - 1] > v) { w[k] = w[k - 1]; --k; }
    w[k] = v;
  }
  return w[h];
}

// a lane's double as a wave-uniform value (two v_readlane_b32; `lane` is a compile-time constant at every call site --
// a ds_bpermute shuffle here put ~100 cycles of LDS latency, sixteen times, on every step of the serial chain)
__device__ inline double readlane_d(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// One wave: min-plus recursion over n frames with the K candidates on the lanes (lane % K = candidate).
//   val(i, k): the candidate's pitch, local(i, k): its own cost, scal(i): a per-step scalar of the step i-1 -> i,
//   cost(cur, prv, s): transition cost.  The per-frame operands are fetched one frame ahead, so that only shuffles
//   and fp64 arithmetic sit on the serial chain.  back: [n][K] bytes (LDS, or global for very long utterances);
//   lane 0 walks it backwards into `path`.
template <int K, class Val, class Local, class Scal, class Cost>
__device__ inline void viterbi_wave(int n, Val val, Local local, Scal scal, Cost cost, uint8_t* back, int* path) {
  const int lane = threadIdx.x & 63;
  const int to = lane % K;
  double v_prev = val(0, to);
  double cum = local(0, to);
  double nv = 0.0, nl = 0.0, ns = 0.0;
  if (n > 1) { nv = val(1, to); nl = local(1, to); ns = scal(1); }
  for (int i = 1; i < n; ++i) {
    const double v_cur = nv, l_cur = nl, s_cur = ns;
    if (i + 1 < n) { nv = val(i + 1, to); nl = local(i + 1, to); ns = scal(i + 1); }
    double best = 0.0;
    int arg = 0;
#pragma unroll
    for (int fr = 0; fr < K; ++fr) {
      const double c = cost(v_cur, readlane_d(v_prev, fr), s_cur) + readlane_d(cum, fr);
      if (fr == 0 || c < best) { best = c; arg = fr; }  // first minimum on ties (np.argmin)
    }
    cum = best + l_cur;
    v_prev = v_cur;
    if (lane < K) back[(size_t)i * K + to] = (uint8_t)arg;
  }
  double bestc = 0.0;
  int last = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double c = readlane_d(cum, k);
    if (k == 0 || c < bestc) { bestc = c; last = k; }
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  if (lane == 0) {
    int p = last;
    path[n - 1] = p;
    for (int i = n - 1; i > 0; --i) {
      p = back[(size_t)i * K + p];
      path[i - 1] = p;
    }
  }
}

struct SpecArgs {
  TrackCfg c;
  const float* energy;      // [B][F]
  const float* cand_pitch;  // [B][F][4]
  const float* cand_merit;
  const int32_t* n_frames;  // [B]
  const int32_t* n_tda;     // [B]
  int F;
  double* en_norm;          // [B][F]
  uint8_t* vuv;             // [B][F]
  double* spec;             // [B][F]
  double* spec_std;         // [B]
  int32_t* lag_min;         // [B][F]
  int32_t* lag_max;
  double* wsd;              // per utterance: 12 F doubles
  int32_t* wsi;             // per utterance: 3 F ints
  uint8_t* wsb;             // per utterance: 8 F bytes (back-pointers when LDS is too small)
  int back_in_lds;
};

__global__ void __launch_bounds__(DP_NT) yaapt_spec_track_kernel(const SpecArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_back[];
  __shared__ double red[DP_NT];
  __shared__ int sc[DP_NT];
  __shared__ double sh_d[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int F = a.F, f = a.n_frames[b], ntd = a.n_tda[b] < f ? a.n_tda[b] : f;
  const TrackCfg& c = a.c;
  const float* en_in = a.energy + (size_t)b * F;
  const float* cp = a.cand_pitch + (size_t)b * F * 4;
  const float* cm = a.cand_merit + (size_t)b * F * 4;
  double* en = a.en_norm + (size_t)b * F;
  uint8_t* vuv = a.vuv + (size_t)b * F;
  double* spec = a.spec + (size_t)b * F;
  int32_t* lmin = a.lag_min + (size_t)b * F;
  int32_t* lmax = a.lag_max + (size_t)b * F;
  double* W = a.wsd + (size_t)b * 12 * F;
  auto vp = [&](int k) { return W + (size_t)k * F; };        // candidate pitch rows over the compacted voiced frames
  auto vm = [&](int k) { return W + (size_t)(4 + k) * F; };  // their merits
  double* picked = W + 8 * (size_t)F;   // picked / smoothed candidate, later the Viterbi track
  double* track = W + 9 * (size_t)F;
  double* raw = W + 10 * (size_t)F;     // spec before the 3-tap mean
  double* dk = W + 11 * (size_t)F;      // pchip derivatives per knot
  int32_t* vidx = a.wsi + (size_t)b * 3 * F;  // voiced frame of compacted column c
  int32_t* knot = vidx + F;                   // frame of knot k
  int32_t* kcnt = knot + F;                   // knots at frames <= i
  uint8_t* back = a.back_in_lds ? lds_back : a.wsb + (size_t)b * 8 * F;

  for (int i = tid; i < F; i += DP_NT) {
    lmin[i] = 1;
    lmax[i] = 2;
    if (i >= f) { en[i] = 0.0; vuv[i] = 0; spec[i] = 0.0; }
  }
  if (f <= 0) {
    if (tid == 0) a.spec_std[b] = 1.0;
    return;
  }
  // ---- NLFER normalisation and the voicing decision
  double s = 0.0;
  for (int i = tid; i < f; i += DP_NT) s += (double)en_in[i];
  const double mean = block_sum(s, red) / (double)f;
  // contiguous chunk per thread for the order-preserving compactions
  const int per = (f + DP_NT - 1) / DP_NT;
  const int i0 = tid * per, i1 = (i0 + per < f) ? i0 + per : f;
  int cnt = 0;
  for (int i = i0; i < i1; ++i) {
    const double e = mean > 0.0 ? (double)en_in[i] / mean : (double)en_in[i];
    en[i] = e;
    const bool v = e > c.nlfer_thresh1;
    vuv[i] = v;
    cnt += (v && cp[(size_t)i * 4] > 0.f) ? 1 : 0;
  }
  int nv;
  int off = block_scan(cnt, sc, &nv);
  double t_avg = 150.0, t_std = 150.0 * c.min_std;
  if (nv > 0) {
    for (int i = i0; i < i1; ++i)
      if (vuv[i] && cp[(size_t)i * 4] > 0.f) {
        vidx[off] = i;
        for (int k = 0; k < 4; ++k) {
          vp(k)[off] = (double)cp[(size_t)i * 4 + k];
          vm(k)[off] = (double)cm[(size_t)i * 4 + k];
        }
        ++off;
      }
    __syncthreads();
    s = 0.0;
    for (int q = tid; q < nv; q += DP_NT) s += vp(0)[q];
    const double avg = block_sum(s, red) / (double)nv;
    s = 0.0;
    for (int q = tid; q < nv; q += DP_NT) s += (vp(0)[q] - avg) * (vp(0)[q] - avg);
    const double std = sqrt(block_sum(s, red) / (double)nv);
    // the candidate closest to 0.8 avg (weighted by merit) is median-smoothed in place
    for (int q = tid; q < nv; q += DP_NT) {
      int arg = 0;
      double best = 0.0;
      for (int k = 0; k < 4; ++k) {
        const double d = fabs(vp(k)[q] - 0.8 * avg) * (3.0 - vm(k)[q]);
        if (k == 0 || d < best) { best = d; arg = k; }
      }
      picked[q] = vp(arg)[q];
      vidx[q] |= arg << 28;  // remember which row (frames < 2^28)
    }
    __syncthreads();
    int k5 = c.median_value - 2;
    k5 = k5 < 1 ? 1 : (k5 > MAXMED ? MAXMED : k5);
    for (int q = tid; q < nv; q += DP_NT) track[q] = median_at(picked, nv, q, k5);
    __syncthreads();
    for (int q = tid; q < nv; q += DP_NT) {
      const int arg = (vidx[q] >> 28) & 3;
      vidx[q] &= (1 << 28) - 1;
      vp(arg)[q] = track[q];
    }
    __syncthreads();
    if (nv > 2) {
      if (tid < 64) {
        const double kk = c.dp5_k1 * std / avg;
        const double f0_min = c.f0_min;
        viterbi_wave<4>(
            nv, [&](int i, int k) { return vp(k)[i]; }, [&](int i, int k) { return 1.0 - vm(k)[i]; },
            [](int) { return 0.0; }, [&](double cur, double prv, double) { return kk * fabs(cur - prv) / f0_min; }, back,
            kcnt /* path, reused below */);
      }
      __syncthreads();
      for (int q = tid; q < nv; q += DP_NT) picked[q] = vp(kcnt[q])[q];
      __syncthreads();
      for (int q = tid; q < nv; q += DP_NT) track[q] = median_at(picked, nv, q, k5);
    } else {
      for (int q = tid; q < nv; q += DP_NT) track[q] = 150.0;
    }
    __syncthreads();
    s = 0.0;
    for (int q = tid; q < nv; q += DP_NT) s += track[q];
    t_avg = block_sum(s, red) / (double)nv;
    s = 0.0;
    for (int q = tid; q < nv; q += DP_NT) s += (track[q] - t_avg) * (track[q] - t_avg);
    t_std = sqrt(block_sum(s, red) / (double)nv);
    if (t_std < t_avg * c.min_std) t_std = t_avg * c.min_std;
    // scatter, end points, knots = non-zero frames
    for (int i = tid; i < f; i += DP_NT) raw[i] = 0.0;
    __syncthreads();
    for (int q = tid; q < nv; q += DP_NT) raw[vidx[q]] = track[q];
    __syncthreads();
    if (tid == 0) {
      if (raw[0] < t_avg / 2) raw[0] = t_avg;
      if (raw[f - 1] < t_avg / 2) raw[f - 1] = t_avg;
    }
    __syncthreads();
    cnt = 0;
    for (int i = i0; i < i1; ++i) cnt += raw[i] != 0.0;
    int nk;
    off = block_scan(cnt, sc, &nk);
    for (int i = i0; i < i1; ++i) {
      if (raw[i] != 0.0) knot[off++] = i;
      kcnt[i] = off;
    }
    __syncthreads();
    if (nk > 1) {
      // scipy PchipInterpolator derivatives (harmonic mean of the secants, monotone end conditions)
      auto hk = [&](int k) { return (double)(knot[k + 1] - knot[k]); };
      auto mk = [&](int k) { return (raw[knot[k + 1]] - raw[knot[k]]) / hk(k); };
      auto sgn = [](double v) { return (v > 0.0) - (v < 0.0); };
      auto edge = [&](double h0, double h1, double m0, double m1) {
        double d = ((2.0 * h0 + h1) * m0 - h0 * m1) / (h0 + h1);
        if (sgn(d) != sgn(m0)) d = 0.0;
        else if (sgn(m0) != sgn(m1) && fabs(d) > 3.0 * fabs(m0)) d = 3.0 * m0;
        return d;
      };
      for (int k = tid; k < nk; k += DP_NT) {
        double d;
        if (nk == 2) {
          d = mk(0);
        } else if (k == 0) {
          d = edge(hk(0), hk(1), mk(0), mk(1));
        } else if (k == nk - 1) {
          d = edge(hk(nk - 2), hk(nk - 3), mk(nk - 2), mk(nk - 3));
        } else {
          const double m0 = mk(k - 1), m1 = mk(k), h0 = hk(k - 1), h1 = hk(k);
          if (sgn(m0) != sgn(m1) || m0 == 0.0 || m1 == 0.0) {
            d = 0.0;
          } else {
            const double w1 = 2.0 * h1 + h0, w2 = h1 + 2.0 * h0;
            d = 1.0 / ((w1 / m0 + w2 / m1) / (w1 + w2));
          }
        }
        dk[k] = d;
      }
      __syncthreads();
      for (int i = tid; i < f; i += DP_NT) {
        if (raw[i] != 0.0) { track[i] = raw[i]; continue; }   // track[] now holds the interpolated contour
        const int k = kcnt[i] - 1;  // knot[k] < i < knot[k+1] (frames 0 and f-1 are knots)
        const double h = hk(k), t = (double)(i - knot[k]) / h;
        const double y0 = raw[knot[k]], y1 = raw[knot[k + 1]];
        const double t2 = t * t, t3 = t2 * t;
        track[i] = (2 * t3 - 3 * t2 + 1) * y0 + (t3 - 2 * t2 + t) * h * dk[k] + (-2 * t3 + 3 * t2) * y1 +
                   (t3 - t2) * h * dk[k + 1];
      }
    } else {
      for (int i = tid; i < f; i += DP_NT) track[i] = t_avg;
    }
  } else {
    for (int i = tid; i < f; i += DP_NT) track[i] = 150.0;
  }
  __syncthreads();
  if (nv > 0) {
    // lfilter(ones(3)/3, 1, .) in scipy's transposed direct form: y[n] = b x[n] + (b x[n-1] + b x[n-2])
    const double b3 = 1.0 / 3.0;
    for (int i = tid; i < f; i += DP_NT) {
      const double x1 = i >= 1 ? track[i - 1] : 0.0, x2 = i >= 2 ? track[i - 2] : 0.0;
      spec[i] = b3 * track[i] + (b3 * x1 + b3 * x2);
    }
    __syncthreads();
    if (tid == 0 && f > 3) {
      spec[0] = spec[2];
      spec[1] = spec[3];
    }
  } else {
    for (int i = tid; i < f; i += DP_NT) spec[i] = 150.0;
  }
  __syncthreads();
  if (tid == 0) a.spec_std[b] = t_std;
  // NCCF search ranges
  const int half = c.nccf_pwidth / 2;
  for (int i = tid; i < ntd; i += DP_NT) {
    double lo = spec[i] - 2.0 * t_std, hi = spec[i] + 2.0 * t_std;
    lo = lo > c.f0_min ? lo : c.f0_min;
    hi = hi < c.f0_max ? hi : c.f0_max;
    lmin[i] = (int32_t)trunc((double)c.fs / hi) - half;
    lmax[i] = (int32_t)trunc((double)c.fs / lo) + half;
  }
  (void)sh_d;
}

struct FinalArgs {
  TrackCfg c;
  const float *tp1, *tm1, *tp2, *tm2;  // [B][F][3]
  const double* en_norm;               // [B][F]
  const uint8_t* vuv;
  const double* spec;
  const double* spec_std;              // [B]
  const int32_t* n_tda;
  int F;
  float* f0;                           // [B][F]
  double* wsd;                         // per utterance: 17 F doubles
  int32_t* wsi;                        // per utterance: F ints
  uint8_t* wsb;                        // per utterance: 8 F bytes
  int back_in_lds;
};

__global__ void __launch_bounds__(DP_NT) yaapt_final_track_kernel(const FinalArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds_back[];
  __shared__ double red[DP_NT];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int F = a.F;
  const TrackCfg& c = a.c;
  int t = a.n_tda[b];
  t = t < 0 ? 0 : (t > F ? F : t);
  float* f0 = a.f0 + (size_t)b * F;
  for (int i = tid; i < F; i += DP_NT) f0[i] = 0.f;
  if (t <= 0) return;
  const double* en = a.en_norm + (size_t)b * F;
  const uint8_t* vuv = a.vuv + (size_t)b * F;
  const double* sp = a.spec + (size_t)b * F;
  const double std = a.spec_std[b];
  double* W = a.wsd + (size_t)b * 17 * F;
  auto P = [&](int k) { return W + (size_t)k * F; };        // candidate table: pitch rows
  auto M = [&](int k) { return W + (size_t)(8 + k) * F; };  //                  merit rows
  double* top = W + 16 * (size_t)F;  // highest-merit candidate before the table is built (median input)
  int32_t* path = a.wsi + (size_t)b * F;
  uint8_t* back = a.back_in_lds ? lds_back : a.wsb + (size_t)b * 8 * F;
  const float* src_p[2] = {a.tp1 + (size_t)b * F * 3, a.tp2 + (size_t)b * F * 3};
  const float* src_m[2] = {a.tm1 + (size_t)b * F * 3, a.tm2 + (size_t)b * F * 3};
  const double thresh = 5.0 * std;
  const float boost = (float)(1.0 + c.merit_boost);  // a Python float times a float32 array stays float32

  // ---- re-weight by the distance to the spectral track, stable sort by merit (descending)
  for (int i = tid; i < t; i += DP_NT) {
    double p[6], m[6];
    for (int k = 0; k < 6; ++k) {
      const float pf = src_p[k / 3][(size_t)i * 3 + k % 3], mf = src_m[k / 3][(size_t)i * 3 + k % 3];
      const double diff = fabs((double)pf - sp[i]);
      const double w = diff < thresh ? 1.0 - diff / thresh : 0.0;
      p[k] = (double)pf;
      m[k] = (double)(boost * mf) * w;
    }
    for (int k = 1; k < 6; ++k) {  // insertion sort, stable
      const double pk = p[k], mkk = m[k];
      int j = k;
      while (j > 0 && m[j - 1] < mkk) { p[j] = p[j - 1]; m[j] = m[j - 1]; --j; }
      p[j] = pk;
      m[j] = mkk;
    }
    for (int k = 0; k < 5; ++k) { P(k)[i] = p[k]; M(k)[i] = m[k]; }
    P(7)[i] = p[5];
    M(7)[i] = m[5];
    top[i] = p[0];
  }
  __syncthreads();
  int ks = c.median_value;
  ks = ks < 1 ? 1 : (ks > MAXMED ? MAXMED : ks);
  for (int i = tid; i < t; i += DP_NT) {
    const bool vu = vuv[i] != 0;
    const double best = vu ? median_at(top, t, i, ks) : 0.0;
    const double e = en[i];
    const bool dead = e <= c.nlfer_thresh2;
    const bool has = P(0)[i] > 0.0 && !dead;
    const bool none = !has && !dead;
    const double m0 = M(0)[i];
    if (has) {
      for (int k = 1; k < 5; ++k)
        if (P(k)[i] == 0.0) M(k)[i] = 0.0;
      P(7)[i] = 0.0;
      M(7)[i] = 1.0 - m0;
    }
    if (none) {
      const double fall = fmin(1.0, e / 2.0);
      P(0)[i] = sp[i];
      M(0)[i] = fall;
      for (int k = 1; k < 5; ++k) { P(k)[i] = 0.0; M(k)[i] = 1.0 - fall; }
      P(7)[i] = 0.0;
      M(7)[i] = 1.0 - fall;
    }
    if (dead) {
      for (int k = 0; k < 5; ++k) { P(k)[i] = 0.0; M(k)[i] = c.merit_pivot; }
      P(7)[i] = 0.0;
      M(7)[i] = c.merit_pivot;
    }
    P(5)[i] = best;
    M(5)[i] = best > 0.0 ? c.merit_extra : 0.0;
    P(6)[i] = vu ? sp[i] : 0.0;
    M(6)[i] = vu ? c.merit_extra : 0.0;
  }
  __syncthreads();
  // ---- mean of the voiced top candidates
  double s = 0.0, n = 0.0;
  for (int i = tid; i < t; i += DP_NT)
    if (P(0)[i] > 0.0) { s += P(0)[i]; n += 1.0; }
  s = block_sum(s, red);
  n = block_sum(n, red);
  const double mean_pitch = n > 0.0 ? s / n : 150.0;
  if (tid < 64) {
    const double w1 = c.dp_w1, w2 = c.dp_w2, w3 = c.dp_w3, w4 = c.dp_w4;
    const double inv_mean = 1.0 / mean_pitch;  // eight fp64 divisions per frame on the serial chain otherwise
    viterbi_wave<8>(
        t, [&](int i, int k) { return P(k)[i]; }, [&](int i, int k) { return w4 * (1.0 - M(k)[i]); },
        [&](int i) { return w2 * (1.0 - fmin(1.0, fabs(en[i - 1] - en[i]))); },
        [&](double cur, double prv, double mixed) {
          if (cur > 0.0 && prv > 0.0) return w1 * fabs(cur - prv) * inv_mean;  // (numpy divides: <= 1 ulp apart)
          if (cur == 0.0 && prv == 0.0) return w3;
          return mixed;
        },
        back, path);
  }
  __syncthreads();
  for (int i = tid; i < t; i += DP_NT) f0[i] = (float)P(path[i])[i];
}

size_t track_ws_bytes(int B, int F) {
  const size_t f = (size_t)(F > 0 ? F : 1);
  return (size_t)B * (17 * f * sizeof(double) + 3 * f * sizeof(int32_t) + 8 * f) + 256;
}

TrackCfg to_cfg(const DisscYaaptTrackConfig* c) {
  TrackCfg t;
  t.nlfer_thresh1 = c->nlfer_thresh1; t.nlfer_thresh2 = c->nlfer_thresh2; t.dp5_k1 = c->dp5_k1;
  t.merit_boost = c->merit_boost; t.merit_pivot = c->merit_pivot; t.merit_extra = c->merit_extra;
  t.dp_w1 = c->dp_w1; t.dp_w2 = c->dp_w2; t.dp_w3 = c->dp_w3; t.dp_w4 = c->dp_w4;
  t.min_std = c->spec_pitch_min_std; t.f0_min = c->f0_min; t.f0_max = c->f0_max;
  t.median_value = c->median_value; t.nccf_pwidth = c->nccf_pwidth; t.fs = c->fs;
  return t;
}

bool bad_cfg(const DisscYaaptTrackConfig* c) {
  return !c || c->median_value < 3 || c->median_value > MAXMED || !(c->median_value & 1) || c->fs <= 0 ||
         !(c->f0_min > 0) || !(c->f0_max > c->f0_min);
}

constexpr size_t kLdsBack = 150 * 1024;

}  // namespace

}  // namespace dissc

using namespace dissc;

extern "C" {

size_t dissc_yaapt_track_workspace_bytes(int B, int F) { return (B > 0 && F > 0) ? track_ws_bytes(B, F) : 0; }

int dissc_yaapt_spec_track(const DisscYaaptTrackConfig* cfg, const float* energy, const float* cand_pitch,
                           const float* cand_merit, const int32_t* n_frames, const int32_t* n_tda, int B, int F,
                           double* en_norm, uint8_t* vuv, double* spec, double* spec_std, int32_t* lag_min,
                           int32_t* lag_max, void* workspace, size_t workspace_bytes, void* stream) {
  if (bad_cfg(cfg) || !energy || !cand_pitch || !cand_merit || !n_frames || !n_tda || B <= 0 || F <= 0 || !en_norm ||
      !vuv || !spec || !spec_std || !lag_min || !lag_max || !workspace || F >= (1 << 28)) {
    set_error("dissc_yaapt_spec_track: bad argument");
    return DISSC_EINVAL;
  }
  if (workspace_bytes < track_ws_bytes(B, F)) {
    set_error("dissc_yaapt_spec_track: workspace of %zu bytes, %zu needed", workspace_bytes, track_ws_bytes(B, F));
    return DISSC_EINVAL;
  }
  SpecArgs a;
  a.c = to_cfg(cfg);
  a.energy = energy; a.cand_pitch = cand_pitch; a.cand_merit = cand_merit; a.n_frames = n_frames; a.n_tda = n_tda;
  a.F = F; a.en_norm = en_norm; a.vuv = vuv; a.spec = spec; a.spec_std = spec_std; a.lag_min = lag_min;
  a.lag_max = lag_max;
  const uintptr_t base = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
  a.wsd = reinterpret_cast<double*>(base);
  a.wsi = reinterpret_cast<int32_t*>(a.wsd + (size_t)B * 17 * F);
  a.wsb = reinterpret_cast<uint8_t*>(a.wsi + (size_t)B * 3 * F);
  const size_t lds = (size_t)4 * F;
  a.back_in_lds = lds <= kLdsBack;
  static DeviceOnce attr_once;  // per device (common.h)
  DISSC_HIP_CHECK(attr_once.max_lds(reinterpret_cast<const void*>(&yaapt_spec_track_kernel), (int)kLdsBack));
  hipLaunchKernelGGL(yaapt_spec_track_kernel, dim3(B), dim3(DP_NT), a.back_in_lds ? lds : 0, (hipStream_t)stream, a);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

int dissc_yaapt_final_track(const DisscYaaptTrackConfig* cfg, const float* tp1, const float* tm1, const float* tp2,
                            const float* tm2, const double* en_norm, const uint8_t* vuv, const double* spec,
                            const double* spec_std, const int32_t* n_tda, int B, int F, float* f0, void* workspace,
                            size_t workspace_bytes, void* stream) {
  if (bad_cfg(cfg) || !tp1 || !tm1 || !tp2 || !tm2 || !en_norm || !vuv || !spec || !spec_std || !n_tda || B <= 0 ||
      F <= 0 || !f0 || !workspace) {
    set_error("dissc_yaapt_final_track: bad argument");
    return DISSC_EINVAL;
  }
  if (workspace_bytes < track_ws_bytes(B, F)) {
    set_error("dissc_yaapt_final_track: workspace of %zu bytes, %zu needed", workspace_bytes, track_ws_bytes(B, F));
    return DISSC_EINVAL;
  }
  FinalArgs a;
  a.c = to_cfg(cfg);
  a.tp1 = tp1; a.tm1 = tm1; a.tp2 = tp2; a.tm2 = tm2; a.en_norm = en_norm; a.vuv = vuv; a.spec = spec;
  a.spec_std = spec_std; a.n_tda = n_tda; a.F = F; a.f0 = f0;
  const uintptr_t base = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
  a.wsd = reinterpret_cast<double*>(base);
  a.wsi = reinterpret_cast<int32_t*>(a.wsd + (size_t)B * 17 * F);
  a.wsb = reinterpret_cast<uint8_t*>(a.wsi + (size_t)B * 3 * F);
  const size_t lds = (size_t)8 * F;
  a.back_in_lds = lds <= kLdsBack;
  static DeviceOnce attr_once;  // per device (common.h)
  DISSC_HIP_CHECK(attr_once.max_lds(reinterpret_cast<const void*>(&yaapt_final_track_kernel), (int)kLdsBack));
  hipLaunchKernelGGL(yaapt_final_track_kernel, dim3(B), dim3(DP_NT), a.back_in_lds ? lds : 0, (hipStream_t)stream, a);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

}  // extern "C"
