// Shared by respair_f23.hip (C = 32) and respair16_f23.hip (C = 16): the register-only Toom-Cook F(2,3) residual pairs.
#pragma once
#include "common.h"

namespace dissc {

struct PairFArgs {
  const float* x;
  float* out;
  float* acc;
  const float* w1;  // [chunk 2][sub-filter 4][point 4][half 2][64 lanes][4 k-steps]
  const float* w2;
  const float* b1;
  const float* b2;
  const int32_t* lengths;
  int len_default, len_mul;
  int ld;
  long long bstride;
  float slope, mrf_div;
  int epi;
  int dbg;  // diagnostics ("kernel_dbg" option): knock-outs -- bit 0 the tap loops, 1 the T epilogue, 2 the output epilogue
};

constexpr int f23_round32_16(int n) { return (n - 16 + 31) / 32 * 32 + 16; }  // smallest v >= n with v % 32 == 16

// (b, first output column) of workgroup `lin` when only the tiles that EXIST are enumerated (respair.hip's pair_tile: the
// workgroups beyond the last real tile all sit at the end of the dispatch order and return at once)
template <int WOUT>
__device__ __forceinline__ bool f23_tile(const PairFArgs& a, int B, int& b, int& len, int& o0) {
  const int lin = blockIdx.y * gridDim.x + blockIdx.x;
  if (a.lengths == nullptr) {
    b = blockIdx.y;
    len = a.len_default;
    o0 = blockIdx.x * WOUT;
    return o0 < len;
  }
  const int lane = threadIdx.x & 63;
  int base = 0;
  for (int b0 = 0; b0 < B; b0 += 64) {
    const int l = b0 + lane < B ? a.lengths[b0 + lane] * a.len_mul : 0;
    const int nt = (l + WOUT - 1) / WOUT;
    int incl = nt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    const int total = __shfl(incl, 63, 64);
    if (lin < base + total) {
      const unsigned long long m = __ballot(base + incl > lin);
      const int lb = __ffsll((long long)m) - 1;
      b = __builtin_amdgcn_readfirstlane(b0 + lb);
      len = __builtin_amdgcn_readfirstlane(__shfl(l, lb, 64));
      o0 = __builtin_amdgcn_readfirstlane((lin - base - __shfl(incl - nt, lb, 64)) * WOUT);
      return true;
    }
    base += total;
  }
  return false;
}

// respair16_f23.hip
int pack_pair16_f23(const float* w, float** dev, int KS);
int launch_pair16_f23(const PairFArgs& a, int KS, int dil, int B, int Lmax, hipStream_t stream);

}  // namespace dissc
