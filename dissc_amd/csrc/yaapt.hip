// YAAPT F0 tracker, device front end: the numerically heavy stages of the tracker the
// reference reaches through amfm_decompy.pYAAPT (reference sr/dataset.py:27-43, eval.py:26-33; inside
// textless' SpeechEncoder for data/encode.py:32) -- algorithm restated in oracle/yaapt_ref.py, PARITY UNPINNED
// (amfm_decompy is not available offline).
//
//   fir            band-pass FIR of the signal and of its square (one pass, both outputs)
//   frame spectra  the zero-padded 8192-point DFT of every frame, but only the bins the algorithm reads:
//                  a [bins x frame_len] x [frame_len x frames] GEMM on the fp32 matrix cores (the windowed
//                  frames are laid out channels-first by an im2col kernel and go through the 1x1 instance of
//                  conv_mfma32_kernel) -- no FFT library, no 8192-point transform of mostly zeros
//   nlfer          sum of the magnitudes of the low-frequency band per frame
//   shc + peaks    spectral harmonics correlation from the magnitudes (mean removal of the frame folded in
//                  analytically), then the candidate peak logic, one thread per frame
//   nccf + peaks   normalised cross-correlation of every frame inside its own lag range (set by the spectral
//                  track), then the candidate logic (cmp_rate), one workgroup per frame
// The short sequential stages (median smoothing, the two dynamic-programming passes, interpolation) stay on
// the host (dissc_amd/f0.py).
#include <math.h>

#include "common.h"

namespace dissc {

constexpr int YA_MAXPEAKS = 4;   // shc_maxpeaks
constexpr int YA_MAXCANDS = 3;   // nccf_maxcands
constexpr int YA_MAXLIST = 64;   // candidate list bound inside the peak pickers

// y[n] = sum_k b[k] x[n-k] (causal, zero initial state) for x and x^2
__global__ void __launch_bounds__(256) yaapt_fir_kernel(const float* __restrict__ wav, const int32_t* __restrict__ ns,
                                                        const float* __restrict__ fir, int ntaps, int N,
                                                        float* __restrict__ filt, float* __restrict__ nlfilt) {
  extern __shared__ float sm[];  // [256 + ntaps - 1] samples | [ntaps] coefficients
  float* xs = sm;
  float* bs = sm + 256 + ntaps - 1;
  const int b = blockIdx.y;
  const int n0 = blockIdx.x * 256;
  const int len = ns ? ns[b] : N;
  const float* xb = wav + (size_t)b * N;
  for (int i = threadIdx.x; i < 256 + ntaps - 1; i += 256) {
    const int n = n0 - (ntaps - 1) + i;
    xs[i] = (n >= 0 && n < len) ? xb[n] : 0.f;
  }
  for (int i = threadIdx.x; i < ntaps; i += 256) bs[i] = fir[i];
  __syncthreads();
  const int n = n0 + threadIdx.x;
  if (n >= N) return;
  float a = 0.f, q = 0.f;
  for (int k = 0; k < ntaps; ++k) {
    const float v = xs[threadIdx.x + (ntaps - 1) - k];
    a = fmaf(bs[k], v, a);
    q = fmaf(bs[k], v * v, q);
  }
  const bool ok = n < len;
  filt[(size_t)b * N + n] = ok ? a : 0.f;
  nlfilt[(size_t)b * N + n] = ok ? q : 0.f;
}

// frames per utterance: mode 0 = len(arange(flen/2, n - flen/2, hop)) (spectral frames),
// mode 1 = int((n - (tda - hop)) / hop) capped at fcap (time-domain frames)
__global__ void yaapt_frames_kernel(const int32_t* __restrict__ ns, int B, int Nmax, int flen, int hop, int tda,
                                    int mode, int fcap, int32_t* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int n = ns ? ns[b] : Nmax;
  n = n < 0 ? 0 : (n > Nmax ? Nmax : n);
  int f;
  if (mode == 0) {
    const int a = flen / 2, e = n - flen / 2;
    f = e > a ? (e - a + hop - 1) / hop : 0;
  } else {
    const int v = n - (tda - hop);
    f = v > 0 ? v / hop : 0;
    f = f < fcap ? f : fcap;
  }
  out[b] = f;
}

// frames, channels-first: out[b][n][f] = sig[b][f*hop + n] (0 beyond the utterance / beyond its frames); the
// analysis window is folded into the DFT rows
__global__ void __launch_bounds__(256) yaapt_im2col_kernel(const float* __restrict__ sig, const int32_t* __restrict__ ns,
                                                           const int32_t* __restrict__ nfr, int N, int flen, int hop,
                                                           int F, int ldf, float* __restrict__ out) {
  const int b = blockIdx.z, n = blockIdx.y;
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= ldf) return;
  const int len = ns ? ns[b] : N;
  float v = 0.f;
  if (f < nfr[b]) {
    const int t = f * hop + n;
    if (t < len) v = sig[(size_t)b * N + t];
  }
  out[((size_t)b * flen + n) * ldf + f] = v;
}

// energy[b][f] = sum_k |X_k|, X rows: [cos bins | sin bins] x frames
__global__ void __launch_bounds__(256) yaapt_band_energy_kernel(const float* __restrict__ spec, const int32_t* __restrict__ nfr,
                                                                int nb, int rows_ld, int F, int ldf,
                                                                float* __restrict__ energy) {
  const int b = blockIdx.y;
  const int f = blockIdx.x * 256 + threadIdx.x;
  if (f >= F) return;
  float e = 0.f;
  if (f < nfr[b]) {
    const float* sb = spec + (size_t)b * rows_ld * ldf + f;
    for (int k = 0; k < nb; ++k) {
      const float re = sb[(size_t)k * ldf], im = sb[(size_t)(nb + k) * ldf];
      e += sqrtf(re * re + im * im);
    }
  }
  energy[(size_t)b * F + f] = e;
}

struct ShcArgs {
  const float* spec;   // [B][2*nb (padded rows_ld)][ldf]: rows 0..nb-1 Re-sum (cos), nb..2nb-1 Im-sum (sin)
  const float* dre;    // [nb] DFT of the rectangular frame (mean removal): Re, Im (as sums with +sin)
  const float* dim;
  const int32_t* nfr;
  int nb, rows_ld, F, ldf;
  int flen2;           // SHC frame length (2 x frame_len)
  int nharm, wl, half, min_shc, max_shc;
  float delta;         // Hz per bin
  // peak picking
  int width, center, min_lag, max_lag;
  float thresh1, thresh2, f0_double, f0_half;
  float* cand_pitch;   // [B][F][YA_MAXPEAKS]
  float* cand_merit;
  float* shc_out;      // optional [B][F][max_shc] (tests), may be null
};

constexpr int SHC_FR = 8;  // frames per workgroup

__global__ void __launch_bounds__(256) yaapt_shc_kernel(const ShcArgs a) {
  extern __shared__ float sm[];  // mag [SHC_FR][half + nb] | shc [SHC_FR][max_shc]
  const int magw = a.half + a.nb;
  float* mag = sm;
  float* shc = sm + SHC_FR * magw;
  const int b = blockIdx.y;
  const int f0 = blockIdx.x * SHC_FR;
  const float* sb = a.spec + (size_t)b * a.rows_ld * a.ldf;
  // magnitudes with the frame mean removed: Y_k = X_k - (X_0 / flen2) * D_k
  for (int e = threadIdx.x; e < SHC_FR * magw; e += 256) {
    const int fr = e % SHC_FR, k = e / SHC_FR - a.half;
    const int f = f0 + fr;
    float v = 0.f;
    if (k >= 0 && f < a.F) {
      const float m = sb[f] / (float)a.flen2;  // row 0 = cos bin 0 = sum of the windowed frame
      const float re = sb[(size_t)k * a.ldf + f] - m * a.dre[k];
      const float im = sb[(size_t)(a.nb + k) * a.ldf + f] - m * a.dim[k];
      v = sqrtf(re * re + im * im);
    }
    mag[fr * magw + (e / SHC_FR)] = v;
  }
  __syncthreads();
  const int nrow = a.max_shc - a.min_shc + 1;
  for (int e = threadIdx.x; e < SHC_FR * a.max_shc; e += 256) {
    const int fr = e / a.max_shc, idx = e - fr * a.max_shc;
    const int i = idx - (a.min_shc - 1);
    float s = 0.f;
    if (i >= 0 && i < nrow) {
      const float* mg = mag + fr * magw;
      for (int j = 0; j < a.wl; ++j) {
        float p = 1.f;
        for (int r = 1; r <= a.nharm + 1; ++r) p *= mg[(a.min_shc + i) * r + j];
        s += p;
      }
    }
    shc[fr * a.max_shc + idx] = s;
  }
  __syncthreads();
  if (a.shc_out)
    for (int e = threadIdx.x; e < SHC_FR * a.max_shc; e += 256) {
      const int fr = e / a.max_shc;
      if (f0 + fr < a.F) a.shc_out[((size_t)b * a.F + f0 + fr) * a.max_shc + (e - fr * a.max_shc)] = shc[e];
    }
  // ---- peak picking, one thread per frame (oracle/yaapt_ref.py peaks()) ----
  if (threadIdx.x >= SHC_FR) return;
  const int f = f0 + threadIdx.x;
  if (f >= a.F) return;
  float* d = shc + threadIdx.x * a.max_shc;
  float pitch[YA_MAXPEAKS], merit[YA_MAXPEAKS];
  for (int i = 0; i < YA_MAXPEAKS; ++i) { pitch[i] = 0.f; merit[i] = 1.f; }
  float* op = a.cand_pitch + ((size_t)b * a.F + f) * YA_MAXPEAKS;
  float* om = a.cand_merit + ((size_t)b * a.F + f) * YA_MAXPEAKS;
  auto emit = [&]() {
    for (int i = 0; i < YA_MAXPEAKS; ++i) { op[i] = pitch[i]; om[i] = merit[i]; }
  };
  if (f >= a.nfr[b]) { emit(); return; }
  const int lo = a.min_lag, hi = a.max_lag;  // inclusive range
  float mx = d[lo];
  for (int n = lo + 1; n <= hi; ++n) mx = fmaxf(mx, d[n]);
  const float scale = mx > 1e-14f ? 1.f / mx : 1.f;
  float avg = 0.f;
  for (int n = lo; n <= hi; ++n) avg += d[n] * scale;
  avg /= (float)(hi - lo + 1);
  if (avg > 1.f / a.thresh1) { emit(); return; }
  float lp[YA_MAXLIST], lm[YA_MAXLIST];
  int np_ = 0;
  const int c = a.center;
  for (int n = lo + c + 1; n < hi - c + 1; ++n) {
    const float v = d[n] * scale;
    if (!(v > d[n - 1] * scale && v > d[n + 1] * scale && v > a.thresh2 * avg)) continue;
    bool is_max = true;  // first maximum of the window [n-c, n+c]
    for (int k = n - c; k <= n + c && is_max; ++k) {
      if (k < n && d[k] >= d[n]) is_max = false;
      if (k > n && d[k] > d[n]) is_max = false;
    }
    if (is_max && np_ < YA_MAXLIST) { lp[np_] = (float)n * a.delta; lm[np_] = v; ++np_; }
  }
  float best = 0.f;
  for (int i = 0; i < np_; ++i) best = fmaxf(best, lm[i]);
  if (np_ == 0 || best / avg < a.thresh1) { emit(); return; }
  // stable selection of the YA_MAXPEAKS largest merits
  int n_sel = np_ < YA_MAXPEAKS ? np_ : YA_MAXPEAKS;
  bool used[YA_MAXLIST];
  for (int i = 0; i < np_; ++i) used[i] = false;
  for (int s = 0; s < YA_MAXPEAKS; ++s) { pitch[s] = 0.f; merit[s] = 0.f; }
  for (int s = 0; s < n_sel; ++s) {
    int arg = -1;
    for (int i = 0; i < np_; ++i)
      if (!used[i] && (arg < 0 || lm[i] > lm[arg])) arg = i;
    used[arg] = true;
    pitch[s] = lp[arg];
    merit[s] = lm[arg];
  }
  int n = n_sel;
  if (pitch[0] > a.f0_double) {
    n = n + 1 < YA_MAXPEAKS ? n + 1 : YA_MAXPEAKS;
    pitch[n - 1] = pitch[0] / 2.f;
    merit[n - 1] = merit[1];
  }
  if (pitch[0] < a.f0_half) {
    n = n + 1 < YA_MAXPEAKS ? n + 1 : YA_MAXPEAKS;
    pitch[n - 1] = pitch[0] * 2.f;
    merit[n - 1] = merit[1];
  }
  for (int i = n; i < YA_MAXPEAKS; ++i) { pitch[i] = pitch[0]; merit[i] = merit[0]; }
  emit();
}

// NCCF of frame f of one signal within [lag_min, lag_max) + candidate logic (oracle crs_corr / cmp_rate)
struct NccfArgs {
  const float* sig;        // [B][N]
  const int32_t* ns;       // [B]
  const int32_t* nfr;      // [B] time-domain frames per utterance
  const int32_t* lag_min;  // [B][F]
  const int32_t* lag_max;
  int N, F, tda, hop;
  float fs, thresh1, thresh2;
  int center;
  float* pitch;            // [B][F][YA_MAXCANDS]
  float* merit;
  float* phi_out;          // optional [B][F][tda] (tests), may be null
};

__global__ void __launch_bounds__(64) yaapt_nccf_kernel(const NccfArgs a) {
  extern __shared__ float sm[];  // x [tda] | phi [tda]
  float* x = sm;
  float* phi = sm + a.tda;
  const int b = blockIdx.y, f = blockIdx.x;
  float* op = a.pitch + ((size_t)b * a.F + f) * YA_MAXCANDS;
  float* om = a.merit + ((size_t)b * a.F + f) * YA_MAXCANDS;
  const int lane = threadIdx.x;
  if (f >= a.nfr[b]) {
    if (lane < YA_MAXCANDS) { op[lane] = 0.f; om[lane] = 0.001f; }
    return;
  }
  const int len = a.ns ? a.ns[b] : a.N;
  const float* sb = a.sig + (size_t)b * a.N + (size_t)f * a.hop;
  float s = 0.f;
  for (int i = lane; i < a.tda; i += 64) {
    const float v = (f * a.hop + i) < len ? sb[i] : 0.f;
    x[i] = v;
    s += v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  const float mean = s / (float)a.tda;
  __syncthreads();
  for (int i = lane; i < a.tda; i += 64) {
    x[i] -= mean;
    phi[i] = 0.f;
  }
  __syncthreads();
  const int lmin = a.lag_min[(size_t)b * a.F + f], lmax = a.lag_max[(size_t)b * a.F + f];
  const int n = a.tda - lmax;
  if (n > 0 && lmin >= 1 && lmax > lmin) {
    float e0 = 0.f;
    for (int i = lane; i < n; i += 64) e0 = fmaf(x[i], x[i], e0);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) e0 += __shfl_xor(e0, off);
    for (int lag = lmin + lane; lag < lmax; lag += 64) {
      float num = 0.f, den = 0.f;
      for (int i = 0; i < n; ++i) {
        const float v = x[i + lag];
        num = fmaf(x[i], v, num);
        den = fmaf(v, v, den);
      }
      const float dd = den * e0;
      phi[lag] = dd > 0.f ? num / sqrtf(dd) : 0.f;
    }
  }
  __syncthreads();
  if (a.phi_out)
    for (int i = lane; i < a.tda; i += 64) a.phi_out[((size_t)b * a.F + f) * a.tda + i] = phi[i];
  if (lane != 0) return;
  // ---- cmp_rate, sequential ----
  float pitch[YA_MAXCANDS], merit[YA_MAXCANDS];
  for (int i = 0; i < YA_MAXCANDS; ++i) { pitch[i] = 0.f; merit[i] = 0.001f; }
  const int c = a.center;
  float pmax = 0.f;  // np.amax over the whole (zero-filled) array
  for (int i = 0; i < a.tda; ++i) pmax = fmaxf(pmax, phi[i]);
  int lp[YA_MAXLIST];
  int np_ = 0;
  for (int k = lmin + c; k < lmax - c + 1 && k + 1 < a.tda; ++k)
    if (k >= 1 && phi[k] > phi[k - 1] && phi[k] > phi[k + 1] && phi[k] > a.thresh1 && np_ < YA_MAXLIST) lp[np_++] = k;
  float cp[YA_MAXLIST], cm[YA_MAXLIST];
  int nc = 0;
  if (np_ > 0 && pmax > a.thresh2) {
    int arg = 0;
    for (int i = 1; i < np_; ++i)
      if (phi[lp[i]] > phi[lp[arg]]) arg = i;
    cp[0] = a.fs / (float)lp[arg];
    cm[0] = phi[lp[arg]];
    nc = 1;
  } else {
    for (int i = 0; i < np_; ++i) {
      const int k = lp[i];
      bool is_max = true;
      for (int j = k - c; j <= k + c && is_max; ++j) {
        if (j < 0 || j >= a.tda) continue;
        if (j < k && phi[j] >= phi[k]) is_max = false;
        if (j > k && phi[j] > phi[k]) is_max = false;
      }
      if (is_max) { cp[nc] = a.fs / (float)k; cm[nc] = phi[k]; ++nc; }
    }
  }
  if (nc > 0) {
    bool used[YA_MAXLIST];
    for (int i = 0; i < nc; ++i) used[i] = false;
    const int nsel = nc < YA_MAXCANDS ? nc : YA_MAXCANDS;
    for (int s2 = 0; s2 < nsel; ++s2) {
      int arg = -1;
      for (int i = 0; i < nc; ++i)
        if (!used[i] && (arg < 0 || cm[i] > cm[arg])) arg = i;
      used[arg] = true;
      pitch[s2] = cp[arg];
      merit[s2] = cm[arg];
    }
    for (int i = nsel; i < YA_MAXCANDS; ++i) { pitch[i] = pitch[0]; merit[i] = merit[0]; }
  }
  for (int i = 0; i < YA_MAXCANDS; ++i) { op[i] = pitch[i]; om[i] = merit[i]; }
}

}  // namespace dissc

using namespace dissc;

struct dissc_yaapt {
  int fs = 16000, flen = 320, hop = 80, tda = 400, nfft = 8192, ntaps = 151;
  float f0_min = 60.f, f0_max = 400.f;
  int nharm = 3;
  float shc_window = 40.f, shc_pwidth = 50.f, shc_thresh1 = 5.f, shc_thresh2 = 1.25f, f0_double = 150.f, f0_half = 150.f;
  float nccf_thresh1 = 0.25f, nccf_thresh2 = 0.9f;
  int nccf_pwidth = 5;
  // derived
  int nl_lo = 0, nl_nb = 0;                 // NLFER band: bins nl_lo .. nl_lo + nl_nb - 1
  int wl = 0, half = 0, min_shc = 0, max_shc = 0, shc_nb = 0;
  float delta = 0.f;
  float *fir = nullptr, *dre = nullptr, *dim = nullptr;
  DevConv dft_nlfer, dft_shc;
  ~dissc_yaapt() {
    for (float* p : {fir, dre, dim})
      if (p) (void)hipFree(p);
    free_conv(dft_nlfer);
    free_conv(dft_shc);
  }
};

static inline size_t ya_rup(size_t x, size_t m) { return (x + m - 1) / m * m; }

static int ya_frames(const dissc_yaapt* y, int n) {  // len(arange(flen/2, n - flen/2, hop))
  const int a = y->flen / 2, b = n - y->flen / 2;
  return b > a ? (b - a + y->hop - 1) / y->hop : 0;
}
static int ya_tda_frames(const dissc_yaapt* y, int n) {  // int((n - (tda - hop)) / hop), >= 0
  const int v = n - (y->tda - y->hop);
  return v > 0 ? v / y->hop : 0;
}

// modified Bessel function I0 (kaiser window), series
static double bessel_i0(double x) {
  double s = 1.0, t = 1.0;
  for (int k = 1; k < 50; ++k) {
    t *= (x / (2.0 * k)) * (x / (2.0 * k));
    s += t;
    if (t < 1e-17 * s) break;
  }
  return s;
}

// DFT basis rows [cos bins | sin bins] over `flen` samples with `win` folded in: X_k = sum_n s[n] win[n] e^{-2 pi i k n / nfft}
// (the sin rows hold +sin: the imaginary part's sign does not matter for magnitudes, dim uses the same sign)
static int make_dft(dissc_yaapt* y, int bin0, int nb, int flen, const std::vector<double>& win, DevConv& dc) {
  std::vector<float> w((size_t)2 * nb * flen);
  const double tw = 2.0 * M_PI / y->nfft;
  for (int k = 0; k < nb; ++k)
    for (int n = 0; n < flen; ++n) {
      const long long ph = ((long long)(bin0 + k) * n) % y->nfft;  // exact phase reduction
      w[(size_t)k * flen + n] = (float)(win[n] * cos(tw * (double)ph));
      w[(size_t)(nb + k) * flen + n] = (float)(win[n] * sin(tw * (double)ph));
    }
  return make_conv(w.data(), nullptr, 2 * nb, flen, 1, 1, dc);
}

extern "C" {

int dissc_yaapt_create(const float* fir, int n_taps, const DisscYaaptConfig* cfg, dissc_yaapt_t* out) {
  if (!fir || n_taps < 1 || n_taps > 1024 || !cfg || !out || cfg->frame_len < 16 || cfg->frame_hop < 1 ||
      cfg->tda_len < 16 || cfg->nfft < 2 * cfg->frame_len || cfg->fs <= 0 || cfg->f0_min <= 0 ||
      cfg->f0_max <= cfg->f0_min || cfg->shc_numharms < 1 || cfg->shc_numharms > 7) {
    set_error("dissc_yaapt_create: bad argument");
    return DISSC_EINVAL;
  }
  dissc_yaapt* y = new dissc_yaapt();
  y->fs = cfg->fs; y->flen = cfg->frame_len; y->hop = cfg->frame_hop; y->tda = cfg->tda_len; y->nfft = cfg->nfft;
  y->ntaps = n_taps; y->f0_min = cfg->f0_min; y->f0_max = cfg->f0_max; y->nharm = cfg->shc_numharms;
  y->shc_window = cfg->shc_window_hz; y->shc_pwidth = cfg->shc_pwidth_hz; y->shc_thresh1 = cfg->shc_thresh1;
  y->shc_thresh2 = cfg->shc_thresh2; y->f0_double = cfg->f0_double; y->f0_half = cfg->f0_half;
  y->nccf_thresh1 = cfg->nccf_thresh1; y->nccf_thresh2 = cfg->nccf_thresh2; y->nccf_pwidth = cfg->nccf_pwidth;
  auto fail = [&](int rc) { delete y; return rc; };
  // NLFER band (oracle nlfer()): bins round(2 f0_min / fs * nfft) - 1 .. round(f0_max / fs * nfft) - 1
  const int n_lo = (int)nearbyint((double)y->f0_min * 2.0 / y->fs * y->nfft);
  const int n_hi = (int)nearbyint((double)y->f0_max / y->fs * y->nfft);
  y->nl_lo = n_lo - 1;
  y->nl_nb = n_hi - y->nl_lo;
  // SHC geometry (oracle shc_geometry())
  y->delta = (float)y->fs / (float)y->nfft;
  const double delta = (double)y->fs / y->nfft;
  int wl = (int)((double)y->shc_window / delta);
  y->half = wl / 2;
  if (wl % 2 == 0) wl += 1;
  y->wl = wl;
  y->max_shc = (int)(((double)y->f0_max + 2.0 * y->shc_pwidth) / delta);
  y->min_shc = (int)ceil((double)y->f0_min / delta);
  // highest magnitude index read: (min_shc + nrow - 1) * (nharm + 1) + wl - 1 (in the half-shifted array)
  y->shc_nb = y->max_shc * (y->nharm + 1) + wl - y->half;
  if (y->nl_nb < 1 || y->shc_nb > y->nfft / 2 + 1 || y->min_shc < 1 || y->max_shc <= y->min_shc) {
    set_error("dissc_yaapt_create: inconsistent frequency range");
    return fail(DISSC_EINVAL);
  }
  int rc;
  if ((rc = upload(std::vector<float>(fir, fir + n_taps), &y->fir))) return fail(rc);
  // hann(flen + 2)[1:-1] and kaiser(2 flen, 0.5) (scipy.signal.windows, symmetric)
  std::vector<double> hann(y->flen), kais(2 * y->flen);
  for (int n = 0; n < y->flen; ++n) hann[n] = 0.5 - 0.5 * cos(2.0 * M_PI * (n + 1) / (y->flen + 1));
  const int m2 = 2 * y->flen;
  for (int n = 0; n < m2; ++n) {
    const double r = (n - (m2 - 1) / 2.0) / ((m2 - 1) / 2.0);
    kais[n] = bessel_i0(0.5 * sqrt(1.0 - r * r)) / bessel_i0(0.5);
  }
  if ((rc = make_dft(y, y->nl_lo, y->nl_nb, y->flen, hann, y->dft_nlfer))) return fail(rc);
  if ((rc = make_dft(y, 0, y->shc_nb, m2, kais, y->dft_shc))) return fail(rc);
  // D_k = sum_{n < 2 flen} e^{...}: what subtracting the frame mean removes from bin k (same sign convention)
  std::vector<float> dre(y->shc_nb), dim(y->shc_nb);
  const double tw = 2.0 * M_PI / y->nfft;
  for (int k = 0; k < y->shc_nb; ++k) {
    double sr = 0, si = 0;
    for (int n = 0; n < m2; ++n) {
      const long long ph = ((long long)k * n) % y->nfft;
      sr += cos(tw * (double)ph);
      si += sin(tw * (double)ph);
    }
    dre[k] = (float)sr;
    dim[k] = (float)si;
  }
  if ((rc = upload(dre, &y->dre))) return fail(rc);
  if ((rc = upload(dim, &y->dim))) return fail(rc);
  *out = y;
  return DISSC_OK;
}

void dissc_yaapt_destroy(dissc_yaapt_t y) { delete y; }

int dissc_yaapt_frames(dissc_yaapt_t y, int n_samples) { return y ? ya_frames(y, n_samples) : 0; }
int dissc_yaapt_tda_frames(dissc_yaapt_t y, int n_samples) { return y ? ya_tda_frames(y, n_samples) : 0; }
int dissc_yaapt_shc_bins(dissc_yaapt_t y) { return y ? y->max_shc : 0; }

static size_t ya_ws_floats(const dissc_yaapt* y, int B, int F) {
  const size_t ldf = ya_rup(F > 0 ? F : 1, 4);
  const size_t col = (size_t)B * (2 * y->flen) * ldf;                                    // im2col frames (the larger)
  const size_t spec = (size_t)B * ya_rup(2 * (size_t)y->shc_nb, 256) * ldf;               // spectra rows x frames
  return ya_rup(col, 64) + ya_rup(spec, 64) + 2 * ya_rup((size_t)B, 64) + 256;
}

size_t dissc_yaapt_workspace_bytes(dissc_yaapt_t y, int B, int Nmax) {
  if (!y || B <= 0 || Nmax <= 0) return 0;
  return ya_ws_floats(y, B, ya_frames(y, Nmax)) * sizeof(float) + 512;
}

int dissc_yaapt_spectral(dissc_yaapt_t y, const float* wav, const int32_t* n_samples, int B, int Nmax, float* filt,
                         float* nlfilt, float* energy, float* cand_pitch, float* cand_merit, float* shc_out,
                         void* workspace, size_t ws_bytes, void* stream_) {
  if (!y || !wav || !filt || !nlfilt || !energy || !cand_pitch || !cand_merit || !workspace || B <= 0 || Nmax <= 0) {
    set_error("dissc_yaapt_spectral: bad argument");
    return DISSC_EINVAL;
  }
  const int F = ya_frames(y, Nmax);
  if (F <= 0) {
    set_error("dissc_yaapt_spectral: %d samples give no frame", Nmax);
    return DISSC_EINVAL;
  }
  if (ws_bytes < dissc_yaapt_workspace_bytes(y, B, Nmax)) {
    set_error("dissc_yaapt_spectral: workspace %zu < %zu bytes", ws_bytes, dissc_yaapt_workspace_bytes(y, B, Nmax));
    return DISSC_ENOMEM;
  }
  hipStream_t st = (hipStream_t)stream_;
  const int ldf = (int)ya_rup(F, 4);
  float* base = (float*)ya_rup((size_t)workspace, 256);
  float* col = base;
  float* spec = col + ya_rup((size_t)B * (2 * y->flen) * ldf, 64);
  int32_t* nfr = (int32_t*)(spec + ya_rup((size_t)B * ya_rup(2 * (size_t)y->shc_nb, 256) * ldf, 64));
  hipLaunchKernelGGL(yaapt_frames_kernel, dim3((B + 63) / 64), dim3(64), 0, st, n_samples, B, Nmax, y->flen, y->hop,
                     y->tda, 0, F, nfr);
  // 1. band-pass of x and x^2
  {
    dim3 grid((Nmax + 255) / 256, B);
    const size_t lds = (size_t)(256 + 2 * y->ntaps) * sizeof(float);
    hipLaunchKernelGGL(yaapt_fir_kernel, grid, dim3(256), lds, st, wav, n_samples, y->fir, y->ntaps, Nmax, filt, nlfilt);
  }
  int rc;
  ConvIO io;
  io.lengths_in = nfr;   // "time" axis of the GEMM = frames
  io.len_default = F;
  // 2. NLFER: frames of the filtered signal -> band DFT (hann folded into the rows) -> sum of magnitudes
  {
    dim3 grid((ldf + 255) / 256, y->flen, B);
    hipLaunchKernelGGL(yaapt_im2col_kernel, grid, dim3(256), 0, st, filt, n_samples, nfr, Nmax, y->flen, y->hop, F, ldf, col);
    if ((rc = run_conv_ex(y->dft_nlfer, col, spec, nullptr, io, B, y->flen, ldf, ldf, F, 1.0f, EPI_STORE, st))) return rc;
    dim3 g2((F + 255) / 256, B);
    hipLaunchKernelGGL(yaapt_band_energy_kernel, g2, dim3(256), 0, st, spec, nfr, y->nl_nb, 2 * y->nl_nb, F, ldf, energy);
  }
  // 3. SHC of the squared signal: 2 x frame_len frames (zero-extended past the end), kaiser folded into the rows
  {
    const int fl2 = 2 * y->flen;
    dim3 grid((ldf + 255) / 256, fl2, B);
    hipLaunchKernelGGL(yaapt_im2col_kernel, grid, dim3(256), 0, st, nlfilt, n_samples, nfr, Nmax, fl2, y->hop, F, ldf, col);
    if ((rc = run_conv_ex(y->dft_shc, col, spec, nullptr, io, B, fl2, ldf, ldf, F, 1.0f, EPI_STORE, st))) return rc;
    ShcArgs a;
    a.spec = spec; a.dre = y->dre; a.dim = y->dim; a.nfr = nfr; a.nb = y->shc_nb; a.rows_ld = 2 * y->shc_nb;
    a.F = F; a.ldf = ldf; a.flen2 = fl2; a.nharm = y->nharm; a.wl = y->wl; a.half = y->half; a.min_shc = y->min_shc;
    a.max_shc = y->max_shc; a.delta = y->delta;
    int width = (int)((double)y->shc_pwidth / ((double)y->fs / y->nfft));
    if (width % 2 == 0) width += 1;
    a.width = width;
    a.center = (int)ceil(width / 2.0);
    int min_lag = (int)((double)y->f0_min / ((double)y->fs / y->nfft) - a.center);
    int max_lag = (int)((double)y->f0_max / ((double)y->fs / y->nfft) + a.center);
    if (min_lag < 1) min_lag = 1;
    if (max_lag > y->max_shc - width) max_lag = y->max_shc - width;
    a.min_lag = min_lag; a.max_lag = max_lag;
    a.thresh1 = y->shc_thresh1; a.thresh2 = y->shc_thresh2; a.f0_double = y->f0_double; a.f0_half = y->f0_half;
    a.cand_pitch = cand_pitch; a.cand_merit = cand_merit; a.shc_out = shc_out;
    const size_t lds = ((size_t)SHC_FR * (y->half + y->shc_nb) + (size_t)SHC_FR * y->max_shc) * sizeof(float);
    static DeviceOnce attr_once;  // per device (common.h)
    DISSC_HIP_CHECK(attr_once.max_lds(reinterpret_cast<const void*>(&yaapt_shc_kernel), 160 * 1024));
    if (lds > 160 * 1024) {
      set_error("dissc_yaapt_spectral: %zu bytes of LDS needed", lds);
      return DISSC_EINVAL;
    }
    dim3 g3((F + SHC_FR - 1) / SHC_FR, B);
    hipLaunchKernelGGL(yaapt_shc_kernel, g3, dim3(256), lds, st, a);
  }
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

int dissc_yaapt_nccf(dissc_yaapt_t y, const float* sig, const int32_t* n_samples, const int32_t* lag_min,
                     const int32_t* lag_max, int B, int Nmax, int F, float* pitch, float* merit, float* phi_out,
                     void* workspace, size_t ws_bytes, void* stream_) {
  if (!y || !sig || !lag_min || !lag_max || !pitch || !merit || !workspace || B <= 0 || Nmax <= 0 || F <= 0 ||
      ws_bytes < (size_t)B * sizeof(int32_t) + 256) {
    set_error("dissc_yaapt_nccf: bad argument");
    return DISSC_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream_;
  int32_t* nfr = (int32_t*)ya_rup((size_t)workspace, 256);
  hipLaunchKernelGGL(yaapt_frames_kernel, dim3((B + 63) / 64), dim3(64), 0, st, n_samples, B, Nmax, y->flen, y->hop,
                     y->tda, 1, F, nfr);
  NccfArgs a;
  a.sig = sig; a.ns = n_samples; a.nfr = nfr; a.lag_min = lag_min; a.lag_max = lag_max; a.N = Nmax; a.F = F;
  a.tda = y->tda; a.hop = y->hop; a.fs = (float)y->fs; a.thresh1 = y->nccf_thresh1; a.thresh2 = y->nccf_thresh2;
  a.center = y->nccf_pwidth / 2;
  a.pitch = pitch; a.merit = merit; a.phi_out = phi_out;
  dim3 grid(F, B);
  hipLaunchKernelGGL(yaapt_nccf_kernel, grid, dim3(64), (size_t)2 * y->tda * sizeof(float), st, a);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

}  // extern "C"
