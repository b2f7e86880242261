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
