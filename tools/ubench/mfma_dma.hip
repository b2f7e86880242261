// Micro-benchmark (round 5): can the A fragments of an fp32 MFMA tap loop be STREAMED without costing the matrix pipe?
// Round 3 (mfma_mix.hip) found: A by global_load_dwordx4 into VGPRs 131-135 TFLOP/s whatever the source (MALL / L2 / L1), A
// resident in LDS (ds_read_b128) 144-146, A through a 2-slot LDS-DMA ring with vmcnt(0) per tap 125-131.  conv_s2tc.hip
// (round 5) measured the same +13 % for its A loads and found it insensitive to latency, bytes and wait placement.  What was
// never tried: a DEEP DMA ring -- global_load_lds_dwordx4 issued R - 1 taps ahead in inline asm (the compiler serialises every
// LDS read behind a builtin DMA with vmcnt(0)), waited for with s_waitcnt vmcnt(2 (R - 2)), read back with ds_read_b128.
//   mode 0: MFMA only        1: + B fragments (8 ds_read_b32 per tap)
//   mode 2: 1 + A by global_load_dwordx4 -> VGPR, one tap ahead (today's kernels)
//   mode 3: 1 + A resident in LDS (ds_read_b128)                     (the ceiling)
//   mode 4: 1 + A through the asm DMA ring, R slots (template)
// WAVES = 12 (3 per SIMD, conv_wino / conv_mfma32 occupancy) or 8 (2 per SIMD); one workgroup per CU (LDS-padded).
// Build + run: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/mfma_dma tools/ubench/mfma_dma.hip && /tmp/mfma_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void dma16(const void* gptr, unsigned lds_byte_addr) {
  // LDS address of the wave's first lane in M0; the hardware adds lane * 16
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gptr), "s"(lds_byte_addr) : "memory");
}

template <int MODE, int WAVES, int R>
__global__ void __launch_bounds__(64 * WAVES, 1) k(const float* __restrict__ wslab, float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 36 * 1024; i += 64 * WAVES) lds[i] = 1e-3f * (i & 255);
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  // every wave of a workgroup streams its own 64 KB slab, the same for every workgroup (L2 hits, like a kernel's weights)
  const f32x4* wp = reinterpret_cast<const f32x4*>(wslab) + (size_t)wave * 64 * 64 + lane;
  const float* vt = lds + wave * 8 * 112 + (lane & 31) + (lane >> 5) * 112;
  const f32x4* at4 = reinterpret_cast<const f32x4*>(lds + 12 * 1024 + (wave & 3) * 512) + lane;
  float* const ring = lds + 16 * 1024 + wave * (R * 512);  // R slots of 2 x 1 KB per wave
  const unsigned ring_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)ring);
  f32x4 av[2], avn[2];
  av[0] = f32x4{1.f, 2.f, 3.f, 4.f};
  av[1] = f32x4{0.5f, 0.25f, 2.f, 1.f};
  avn[0] = av[0];
  avn[1] = av[1];
  float b0[2] = {1.f, 2.f}, bk[3][2] = {{1.f, 2.f}, {3.f, 4.f}, {5.f, 6.f}}, b0n[2] = {1.f, 2.f};
  if (MODE == 4) {  // prologue: taps 0 .. R - 2 in flight
    for (int t = 0; t < R - 1; ++t) {
      dma16(wp + (size_t)(t & 31) * 64, ring_addr + (t % R) * 2048);
      dma16(wp + (size_t)((t & 31) + 32) * 64, ring_addr + (t % R) * 2048 + 1024);
    }
  }
  int slot = 0;  // it % R
  for (int it = 0; it < iters; ++it) {
    const int j = it & 3;
    if (MODE == 4) {
      // tap it + R - 1 goes into the slot of tap it - 1 (read into registers two taps ago)
      const int sl = slot == 0 ? R - 1 : slot - 1;
      const int t = it + R - 1;
      dma16(wp + (size_t)(t & 31) * 64, ring_addr + sl * 2048);
      dma16(wp + (size_t)((t & 31) + 32) * 64, ring_addr + sl * 2048 + 1024);
      // tap it + 1 must have landed: R - 2 younger pairs (taps it + 2 .. it + R - 1) may stay in flight
      if (R == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (R == 3) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      if (R == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      if (R == 6) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      const int s1 = slot + 1 == R ? 0 : slot + 1;
      avn[0] = *reinterpret_cast<const f32x4*>(ring + s1 * 512 + lane * 4);
      avn[1] = *reinterpret_cast<const f32x4*>(ring + s1 * 512 + 256 + lane * 4);
    } else if (MODE == 3) {
      avn[0] = at4[(it & 7) * 64];
      avn[1] = at4[(it & 7) * 64 + 8 * 64];
    } else if (MODE == 2) {
      avn[0] = wp[(size_t)(it & 31) * 64];
      avn[1] = wp[(size_t)((it & 31) + 32) * 64];
    }
    if (MODE >= 1) {
#pragma unroll
      for (int s = 1; s < 4; ++s) {
        bk[s - 1][0] = vt[s * 2 * 112 + j];
        bk[s - 1][1] = vt[s * 2 * 112 + j + 32];
      }
      b0n[0] = vt[j + 1];
      b0n[1] = vt[j + 33];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][0], b0[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
    for (int s = 1; s < 4; ++s)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][s], bk[s - 1][ni], acc[mi][ni], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    b0[0] = b0n[0];
    b0[1] = b0n[1];
    av[0] = avn[0];
    av[1] = avn[1];
    slot = slot + 1 == R ? 0 : slot + 1;
  }
  if (MODE == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float sum = 0.f;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int e = 0; e < 16; ++e) sum += acc[a][b][e];
  if (sum == 12345.678f) out[0] = sum;
}

template <int MODE, int WAVES, int R>
static void run(const float* w, float* out, int iters, const char* what) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, WAVES, R>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int blocks = 256 * 4;
  const size_t lds = 150 * 1024;  // one workgroup per CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, WAVES, R>), dim3(blocks), dim3(64 * WAVES), lds, 0, w, out, 64);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE, WAVES, R>), dim3(blocks), dim3(64 * WAVES), lds, 0, w, out, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const hipError_t err = hipGetLastError();
  const double flops = 2.0 * 32 * 32 * 2 * 16.0 * iters * WAVES * blocks;
  printf("%2d waves  mode %d R %d  %-52s %8.3f ms  %6.1f TFLOP/s %s\n", WAVES, MODE, R, what, best, flops / (best * 1e-3) / 1e12,
         err == hipSuccess ? "" : hipGetErrorString(err));
}

int main() {
  float *w = nullptr, *out = nullptr;
  const size_t nw = (size_t)12 * 64 * 64 * 4;
  hipMalloc((void**)&w, nw * sizeof(float));
  hipMalloc((void**)&out, 16);
  hipMemset(w, 0, nw * sizeof(float));
  const int iters = 4096;
  run<0, 12, 2>(w, out, iters, "MFMA only");
  run<1, 12, 2>(w, out, iters, "+ B (8 ds_read_b32 per tap)");
  run<2, 12, 2>(w, out, iters, "+ B + A global -> VGPR, one tap ahead");
  run<3, 12, 2>(w, out, iters, "+ B + A resident in LDS (ceiling)");
  run<4, 12, 2>(w, out, iters, "+ B + A asm DMA ring, vmcnt(0) per tap");
  run<4, 12, 3>(w, out, iters, "+ B + A asm DMA ring of 3");
  run<4, 12, 4>(w, out, iters, "+ B + A asm DMA ring of 4");
  run<4, 12, 6>(w, out, iters, "+ B + A asm DMA ring of 6");
  run<0, 8, 2>(w, out, iters, "MFMA only");
  run<1, 8, 2>(w, out, iters, "+ B (8 ds_read_b32 per tap)");
  run<2, 8, 2>(w, out, iters, "+ B + A global -> VGPR, one tap ahead");
  run<3, 8, 2>(w, out, iters, "+ B + A resident in LDS (ceiling)");
  run<4, 8, 3>(w, out, iters, "+ B + A asm DMA ring of 3");
  run<4, 8, 4>(w, out, iters, "+ B + A asm DMA ring of 4");
  run<4, 8, 6>(w, out, iters, "+ B + A asm DMA ring of 6");
  return 0;
}
