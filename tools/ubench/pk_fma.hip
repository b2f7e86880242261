// Micro-benchmark (round 5): what does vector-ALU work cost the fp32 matrix pipe, and is packed fp32 math (v_pk_fma_f32: two
// IEEE fmas per lane and instruction) really twice as cheap as v_fma_f32 there?  SQ_VALU_MFMA_COEXEC_CYCLES is 0 on every fp32
// kernel of this repo: VALU work is paid in matrix-pipe time.
//   mode 0: VALU only, K v_fma_f32 per iteration (8 independent chains)         mode 1: K / 2 v_pk_fma_f32 (same FLOPs)
//   mode 2: 16 MFMA 32x32x2 per iteration                                        mode 3: 2 + K v_fma_f32     mode 4: 2 + K / 2 v_pk_fma_f32
//   mode 5: 2 + K v_mov_b32
// 8 waves per CU (2 per SIMD), one workgroup per CU.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/pk_fma tools/ubench/pk_fma.hip && /tmp/pk_fma
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int K>
__global__ void __launch_bounds__(512, 1) k(float* out, int iters, float seed) {
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  float s[8];
  f32x2 p[8];
  for (int i = 0; i < 8; ++i) {
    s[i] = seed * (i + 1 + threadIdx.x);
    p[i] = f32x2{seed * i, seed * (i + threadIdx.x)};
  }
  const float ca = 1.0001f, cb = 0.5f;
  const f32x2 pa = {1.0001f, 0.9999f}, pb = {0.5f, 0.25f};
  float av = seed, bv = seed * 2;
  for (int it = 0; it < iters; ++it) {
    if (MODE >= 2) {
#pragma unroll
      for (int m = 0; m < 16; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m & 3], 0, 0, 0);
    }
    if (MODE == 0 || MODE == 3) {
#pragma unroll
      for (int j = 0; j < K; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[j & 7]) : "v"(ca), "v"(cb));
    }
    if (MODE == 1 || MODE == 4) {
#pragma unroll
      for (int j = 0; j < K / 2; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j & 7]) : "v"(pa), "v"(pb));
    }
    if (MODE == 5) {
#pragma unroll
      for (int j = 0; j < K; ++j) asm volatile("v_mov_b32 %0, %1" : "=v"(s[j & 7]) : "v"(ca));
    }
  }
  float sum = 0.f;
  for (int a = 0; a < 4; ++a)
    for (int e = 0; e < 16; ++e) sum += acc[a][e];
  for (int i = 0; i < 8; ++i) sum += s[i] + p[i][0] + p[i][1];
  if (sum == 12345.678f) out[0] = sum;
}

template <int MODE, int K>
static void run(float* out, const char* what) {
  const int iters = 2048, blocks = 256 * 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, K>), dim3(blocks), dim3(512), 0, 0, out, 16, 1e-3f);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE, K>), dim3(blocks), dim3(512), 0, 0, out, iters, 1e-3f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double mfma_flops = MODE >= 2 ? 2.0 * 32 * 32 * 2 * 16.0 * iters * 8 * blocks : 0.0;
  const double valu_flops = (MODE == 2 || MODE == 5) ? 0.0 : 2.0 * 64 * K * (double)iters * 8 * blocks;
  printf("mode %d K %3d  %-44s %8.3f ms  MFMA %6.1f TFLOP/s  VALU %6.1f TFLOP/s\n", MODE, K, what, best,
         mfma_flops / (best * 1e-3) / 1e12, valu_flops / (best * 1e-3) / 1e12);
}

int main() {
  float* out = nullptr;
  hipMalloc((void**)&out, 16);
  run<0, 64>(out, "VALU only: 64 v_fma_f32 / iter");
  run<1, 64>(out, "VALU only: 32 v_pk_fma_f32 / iter");
  run<2, 0>(out, "16 MFMA / iter");
  run<3, 16>(out, "16 MFMA + 16 v_fma_f32");
  run<4, 16>(out, "16 MFMA +  8 v_pk_fma_f32");
  run<3, 64>(out, "16 MFMA + 64 v_fma_f32");
  run<4, 64>(out, "16 MFMA + 32 v_pk_fma_f32");
  run<3, 128>(out, "16 MFMA + 128 v_fma_f32");
  run<4, 128>(out, "16 MFMA + 64 v_pk_fma_f32");
  run<5, 64>(out, "16 MFMA + 64 v_mov_b32");
  return 0;
}
