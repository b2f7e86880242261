// Shared pieces of the Toom-Cook F(4,3) kernels (conv_wino.hip: one conv per launch; respair_wino.hip: a residual pair
// per launch): the point set, the input transform B^T written out per point, the weight transform G (host, double).
// Evaluation points 0, 1, -1, 2, -1/2, inf; the rows of B^T are scaled to exactly representable entries and the scaling
// is folded into G.  A^T = {a == 0, 1, (-1)^a, 2^a, (-1/2)^a, a == 3} is written out in the epilogues.
#pragma once
#include "common.h"

namespace dissc {

// B^T rows (point p, input m) of F(4,3) at the points 0, 1, -1, 2, -1/2, inf (documentation: the kernel evaluates them
// through wino_bt<P> below); A^T is written out in the epilogue
[[maybe_unused]] static const float kWinoBT[6][6] = {{0.5f, 0.75f, -1.0f, -0.75f, 0.5f, 0.0f}, {0.0f, 1.0f, 2.5f, 0.5f, -1.0f, 0.0f},
                                    {0.0f, 1.0f, 0.5f, -2.5f, 1.0f, 0.0f},    {0.0f, -0.5f, -1.0f, 0.5f, 1.0f, 0.0f},
                                    {0.0f, -1.0f, 0.5f, 1.0f, -0.5f, 0.0f},   {0.0f, 0.5f, 0.75f, -1.0f, -0.75f, 0.5f}};
// G rows matching the scaling of kWinoBT (host, double)
static const double kWinoG[6][3] = {{2.0, 0.0, 0.0},
                                    {1.0 / 3.0, 1.0 / 3.0, 1.0 / 3.0},
                                    {1.0 / 3.0, -1.0 / 3.0, 1.0 / 3.0},
                                    {1.0 / 15.0, 2.0 / 15.0, 4.0 / 15.0},
                                    {32.0 / 15.0, -16.0 / 15.0, 8.0 / 15.0},
                                    {0.0, 0.0, 2.0}};

// row P of B^T applied to six neighbouring samples, written out per point (3-5 operations instead of 6: the rows have
// 4-5 non-zero entries, half of them +-1)
template <int P>
__device__ __forceinline__ float wino_bt(float r0, float r1, float r2, float r3, float r4, float r5) {
  if constexpr (P == 0) return fmaf(0.75f, r1 - r3, 0.5f * (r0 + r4)) - r2;
  if constexpr (P == 1) return fmaf(0.5f, r3, fmaf(2.5f, r2, r1 - r4));
  if constexpr (P == 2) return fmaf(-2.5f, r3, fmaf(0.5f, r2, r1 + r4));
  if constexpr (P == 3) return fmaf(0.5f, r3 - r1, r4 - r2);
  if constexpr (P == 4) return fmaf(0.5f, r2 - r4, r3 - r1);
  if constexpr (P == 5) return fmaf(0.75f, r2 - r4, 0.5f * (r1 + r5)) - r3;
  return 0.f;
}

}  // namespace dissc
