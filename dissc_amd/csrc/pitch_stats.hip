// Per-speaker F0 statistics (the step between data/encode.py and infer.py): mean and population
// standard deviation of the VOICED (non-zero) frames of each speaker, in fp64 like the reference's
// numpy code (reference data/data_utils.py:33-46).  The host groups the frames by speaker (it has
// to parse the JSONL anyway) and hands over one contiguous segment per speaker; one workgroup
// reduces one segment with a fixed-shape tree, so the result does not depend on scheduling.
#include "common.h"

namespace dissc {

constexpr int PS_NT = 256;

__device__ __forceinline__ double block_sum(double v, double* red) {
  // 64-lane butterfly, then the 4 wave sums through LDS: same order on every run
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[wave] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(PS_NT) pitch_stats_kernel(const double* __restrict__ f0,
                                                            const int64_t* __restrict__ offsets,
                                                            double* __restrict__ mean_out,
                                                            double* __restrict__ std_out,
                                                            int64_t* __restrict__ count_out) {
  __shared__ double red[4];
  const int s = blockIdx.x;
  const int64_t lo = offsets[s], hi = offsets[s + 1];
  double sum = 0.0, cnt = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += PS_NT) {
    const double v = f0[i];
    if (v != 0.0) {
      sum += v;
      cnt += 1.0;
    }
  }
  sum = block_sum(sum, red);
  cnt = block_sum(cnt, red);
  const double mean = sum / cnt;  // NaN for a speaker without voiced frames, like numpy
  double sq = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += PS_NT) {
    const double v = f0[i];
    if (v != 0.0) sq += (v - mean) * (v - mean);
  }
  sq = block_sum(sq, red);
  if (threadIdx.x == 0) {
    mean_out[s] = mean;
    std_out[s] = sqrt(sq / cnt);
    count_out[s] = (int64_t)cnt;
  }
}

}  // namespace dissc

extern "C" int dissc_pitch_stats(const double* f0, const int64_t* offsets, int n_speakers, double* mean_out,
                                 double* std_out, int64_t* count_out, void* stream) {
  using namespace dissc;
  if (!f0 || !offsets || !mean_out || !std_out || !count_out || n_speakers < 0) {
    set_error("dissc_pitch_stats: bad argument");
    return DISSC_EINVAL;
  }
  if (n_speakers == 0) return DISSC_OK;
  hipLaunchKernelGGL(pitch_stats_kernel, dim3(n_speakers), dim3(PS_NT), 0, (hipStream_t)stream, f0, offsets,
                     mean_out, std_out, count_out);
  DISSC_HIP_CHECK(hipGetLastError());
  return DISSC_OK;
}
