// Device-side glue between the stages of the conversion pipeline and the waveform exchange
// (HBM-bound byte movers; nothing here is arithmetic).
#include "common.h"

namespace dissc {

// Ragged exchange buffer (include/dissc_hip.h): rows lie back to back in the data region, each padded with
// zeros to a multiple of 4 floats, at the offsets the host's prefix sum assigned.  One workgroup per
// (utterance, 4096-float span), 16 B per lane.
constexpr int PACK_SPAN = 4096;
__global__ void __launch_bounds__(256) pack_rows_kernel(const float* __restrict__ wav, long long ld_wav,
                                                        const int32_t* __restrict__ n_samples,
                                                        const long long* __restrict__ offsets,
                                                        float* __restrict__ data) {
  const int b = blockIdx.y;
  int n = n_samples[b];
  if (n < 0) n = 0;
  const long long cap = ((long long)n + 3) & ~3LL;  // the row's slot: n rounded up to 4 floats
  const long long t0 = (long long)blockIdx.x * PACK_SPAN;
  if (t0 >= cap) return;
  const float* src = wav + (size_t)b * ld_wav;
  float* dst = data + offsets[b];
  const bool vec = ((ld_wav & 3) == 0) && ((reinterpret_cast<uintptr_t>(wav) & 15) == 0);
  for (long long t = t0 + threadIdx.x * 4; t < t0 + PACK_SPAN && t < cap; t += 256 * 4) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t + 3 < n && vec) {
      v = *reinterpret_cast<const float4*>(src + t);
    } else {
      if (t < n) v.x = src[t];
      if (t + 1 < n) v.y = src[t + 1];
      if (t + 2 < n) v.z = src[t + 2];
      if (t + 3 < n) v.w = src[t + 3];
    }
    *reinterpret_cast<float4*>(dst + t) = v;  // offsets and cap are multiples of 4 floats
  }
}

}  // namespace dissc

using namespace dissc;

extern "C" int dissc_pack_rows(const float* wav, long long ld_wav, const int32_t* n_samples,
                               const long long* offsets, int B, int n_max, float* data, void* stream) {
  if (!data || B < 0 || n_max < 0 || (B > 0 && (!wav || !n_samples || !offsets))) {
    set_error("dissc_pack_rows: bad argument");
    return DISSC_EINVAL;
  }
  if ((reinterpret_cast<uintptr_t>(data) & 15) != 0) {
    set_error("dissc_pack_rows: data must be 16-byte aligned");
    return DISSC_EINVAL;
  }
  if (B == 0 || n_max == 0) return DISSC_OK;
  dim3 grid((unsigned)((n_max + PACK_SPAN - 1) / PACK_SPAN), B);
  hipLaunchKernelGGL(pack_rows_kernel, grid, dim3(256), 0, (hipStream_t)stream, wav, ld_wav, n_samples, offsets,
                     data);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

// ---- band-limited sinc resampling (the step before the path: reference data/preprocess.py:19-23) --------
// y[t] = sum_i w(frac + i) x[n - i] + sum_k w(1 - frac + k) x[n + 1 + k], n = floor(t / ratio): resampy's
// resample_f [3P-unverified, restated in oracle/preprocess_ref.py] with its Kaiser-windowed sinc table
// (interp_win, linearly interpolated with interp_delta).  fp64 like the reference; one thread per output sample.
namespace dissc {
__global__ void __launch_bounds__(256) resample_kernel(const double* __restrict__ x, int n_orig, double* __restrict__ y,
                                                       int n_out, double ratio, const double* __restrict__ win,
                                                       const double* __restrict__ delta, int nwin, int num_table) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= n_out) return;
  const double scale = ratio < 1.0 ? ratio : 1.0;
  const int index_step = (int)(scale * num_table);
  const double time_register = (double)t * (1.0 / ratio);
  const int n = (int)time_register;
  double frac = scale * (time_register - n);
  double index_frac = frac * num_table;
  int offset = (int)index_frac;
  double eta = index_frac - offset;
  double acc = 0.0;
  int i_max = (nwin - offset) / index_step;
  i_max = i_max < n + 1 ? i_max : n + 1;
  for (int i = 0; i < i_max; ++i) {
    const int j = offset + i * index_step;
    acc += (win[j] + eta * delta[j]) * x[n - i];
  }
  frac = scale - frac;
  index_frac = frac * num_table;
  offset = (int)index_frac;
  eta = index_frac - offset;
  int k_max = (nwin - offset) / index_step;
  k_max = k_max < n_orig - n - 1 ? k_max : n_orig - n - 1;
  for (int k = 0; k < k_max; ++k) {
    const int j = offset + k * index_step;
    acc += (win[j] + eta * delta[j]) * x[n + k + 1];
  }
  y[t] = acc;
}
}  // namespace dissc

extern "C" int dissc_resample(const double* x, int n_orig, double* y, int n_out, double ratio, const double* win,
                              const double* delta, int nwin, int num_table, void* stream) {
  if (!x || !y || !win || !delta || n_orig <= 0 || n_out < 0 || ratio <= 0.0 || nwin < 2 || num_table < 1 ||
      (int)((ratio < 1.0 ? ratio : 1.0) * num_table) < 1) {
    set_error("dissc_resample: bad argument");
    return DISSC_EINVAL;
  }
  if (n_out == 0) return DISSC_OK;
  hipLaunchKernelGGL(resample_kernel, dim3((n_out + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, n_orig, y, n_out,
                     ratio, win, delta, nwin, num_table);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}
