// Device-side glue between the stages of the conversion pipeline and the waveform exchange
// (HBM-bound byte movers; nothing here is arithmetic).
#include "common.h"

namespace dissc {

// One workgroup row-slice per (utterance, 4096-float span): copies the utterance's valid samples
// behind a 4-float header [job id bits | sample count bits | 0 | 0] into its row of the packed
// exchange buffer and zero-fills the rest of the row, 16 B per lane.
constexpr int PACK_SPAN = 4096;
__global__ void __launch_bounds__(256) pack_waves_kernel(const float* __restrict__ wav, long long ld_wav,
                                                         const int32_t* __restrict__ n_samples,
                                                         const int32_t* __restrict__ job_ids,
                                                         float* __restrict__ buf, long long ld_buf,
                                                         int row0) {
  const int b = blockIdx.y;
  int n = n_samples[b];
  const long long cap = ld_buf - 4;
  if (n < 0) n = 0;
  if (n > cap) n = (int)cap;
  const float* src = wav + (size_t)b * ld_wav;
  float* row = buf + (size_t)(row0 + b) * ld_buf;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    int32_t* h = reinterpret_cast<int32_t*>(row);
    h[0] = job_ids[b];
    h[1] = n;
    h[2] = 0;
    h[3] = 0;
  }
  float* dst = row + 4;
  const long long t0 = (long long)blockIdx.x * PACK_SPAN;
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
    if (t + 3 < cap) {
      *reinterpret_cast<float4*>(dst + t) = v;
    } else {
      dst[t] = v.x;
      if (t + 1 < cap) dst[t + 1] = v.y;
      if (t + 2 < cap) dst[t + 2] = v.z;
    }
  }
}

// rows without a job (a rank with fewer jobs than n_max): header job id = -1, n = 0
__global__ void pack_empty_rows_kernel(float* __restrict__ buf, long long ld_buf, int row0, int rows) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  int32_t* h = reinterpret_cast<int32_t*>(buf + (size_t)(row0 + r) * ld_buf);
  h[0] = -1;
  h[1] = 0;
  h[2] = 0;
  h[3] = 0;
}

}  // namespace dissc

using namespace dissc;

extern "C" int dissc_pack_waves(const float* wav, long long ld_wav, const int32_t* n_samples,
                                const int32_t* job_ids, int B, float* buf, long long ld_buf, int row0,
                                void* stream) {
  if (!buf || ld_buf < 4 || (ld_buf & 3) || row0 < 0 || B < 0 || (B > 0 && (!wav || !n_samples || !job_ids))) {
    set_error("dissc_pack_waves: bad argument (ld_buf must be a multiple of 4 floats, >= 4)");
    return DISSC_EINVAL;
  }
  if ((reinterpret_cast<uintptr_t>(buf) & 15) != 0) {
    set_error("dissc_pack_waves: buf must be 16-byte aligned");
    return DISSC_EINVAL;
  }
  if (B == 0) return DISSC_OK;
  const long long cap = ld_buf - 4;
  dim3 grid((unsigned)((cap + PACK_SPAN - 1) / PACK_SPAN > 0 ? (cap + PACK_SPAN - 1) / PACK_SPAN : 1), B);
  hipLaunchKernelGGL(pack_waves_kernel, grid, dim3(256), 0, (hipStream_t)stream, wav, ld_wav, n_samples,
                     job_ids, buf, ld_buf, row0);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}

extern "C" int dissc_pack_empty_rows(float* buf, long long ld_buf, int row0, int rows, void* stream) {
  if (!buf || ld_buf < 4 || row0 < 0 || rows < 0) {
    set_error("dissc_pack_empty_rows: bad argument");
    return DISSC_EINVAL;
  }
  if (rows == 0) return DISSC_OK;
  hipLaunchKernelGGL(pack_empty_rows_kernel, dim3((rows + 63) / 64), dim3(64), 0, (hipStream_t)stream, buf,
                     ld_buf, row0, rows);
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
