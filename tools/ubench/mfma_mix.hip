// Micro-benchmark: what does the operand traffic of conv_wino_kernel's tap loop cost the fp32 matrix pipe?
// 12 waves per workgroup (3 per SIMD), one workgroup per CU (LDS-padded), every wave runs `iters` "taps" of 16
// v_mfma_f32_32x32x2_f32 on 4 accumulator tiles (64 x 64 wave tile) with, per tap and depending on MODE:
//   bit 0: the 8 B-fragment values read from LDS (ds_read_b32 / read2), one k-step ahead like the real loop
//   bit 1: the 2 x 16-byte A-fragment loads from global memory (an L2-resident slab per wave), one tap ahead
//   bit 2: B fragments as 2 x ds_read_b128 (4 k-steps of one column block per read) instead of 8 x b32
//   bit 3: A fragments from LDS (2 x ds_read_b128) instead of global
//   bit 5: A fragments from global TWO taps ahead (three register sets)
//   bit 4: A fragments global -> LDS by DMA (global_load_lds_dwordx4, a 2-slot ring per wave), read back with ds_read_b128
// Build + run: tools/ubench/run.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(768, 3) mix_kernel(const float* __restrict__ wslab, float* out, int iters, int shared) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 16 * 1024; i += 768) lds[i] = 1e-3f * (i & 255);
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  // shared = 1: every wave of the chip streams the SAME 64 KB (L2 / L1 hits); 2: the 12 waves of a workgroup stream 12
  // slabs, the same for every workgroup (768 KB: L2 hits, like the kernel's weights); 0: its own 64 KB (196 MB: MALL)
  const f32x4* wp = reinterpret_cast<const f32x4*>(wslab) + (shared == 1 ? 0 : shared == 2 ? (size_t)wave * 64 * 64 : (size_t)(blockIdx.x * 12 + wave) * 64 * 64) + lane;
  const float* vt = lds + wave * 8 * 112 + (lane & 31) + (lane >> 5) * 112;
  const f32x4* vt4 = reinterpret_cast<const f32x4*>(lds + wave * 1024) + lane;
  const f32x4* at4 = reinterpret_cast<const f32x4*>(lds + 12 * 1024 + (wave & 3) * 512) + lane;
  f32x4 av[2], avn[2], avnn[2];
  av[0] = f32x4{1.f, 2.f, 3.f, 4.f};
  av[1] = f32x4{0.5f, 0.25f, 2.f, 1.f};
  float b0[2] = {1.f, 2.f}, bk[3][2] = {{1.f, 2.f}, {3.f, 4.f}, {5.f, 6.f}}, b0n[2] = {1.f, 2.f};
  for (int it = 0; it < iters; ++it) {
    const int j = it & 3;
    if (MODE & 32) {
      if (it == 0) {
        avn[0] = wp[0];
        avn[1] = wp[32 * 64];
      }
      avnn[0] = wp[(size_t)((it + 2) & 31) * 64];
      avnn[1] = wp[(size_t)(((it + 2) & 31) + 32) * 64];
    } else if (MODE & 16) {
      // slot (it + 1) & 1 was filled by the DMA issued one tap ago; slot it & 1 (read into registers one tap ago) is refilled
      float* ring = lds + 16 * 1024 + wave * 2048;
      // the compiler does not order a ds_read behind the DMA that fills its slot: wait by hand (issued one tap ago)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      avn[0] = *reinterpret_cast<const f32x4*>(ring + ((it + 1) & 1) * 512 + lane * 4);
      avn[1] = *reinterpret_cast<const f32x4*>(ring + ((it + 1) & 1) * 512 + 256 + lane * 4);
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(wp + (size_t)(it & 31) * 64),
                                       (void __attribute__((address_space(3)))*)(ring + (it & 1) * 512), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(wp + (size_t)((it & 31) + 32) * 64),
                                       (void __attribute__((address_space(3)))*)(ring + (it & 1) * 512 + 256), 16, 0, 0);
    } else if (MODE & 8) {
      avn[0] = at4[(it & 7) * 64];
      avn[1] = at4[(it & 7) * 64 + 512 / 4 * 0 + 32 * 0 + 8 * 64];
    } else if (MODE & 2) {
      avn[0] = wp[(size_t)(it & 31) * 64];
      avn[1] = wp[(size_t)((it & 31) + 32) * 64];
    } else {
      avn[0] = av[0];
      avn[1] = av[1];
    }
    if (MODE & 4) {
      const f32x4 q0 = vt4[j * 64], q1 = vt4[(j + 4) * 64];
      bk[0][0] = q0[1]; bk[1][0] = q0[2]; bk[2][0] = q0[3]; b0n[0] = q0[0];
      bk[0][1] = q1[1]; bk[1][1] = q1[2]; bk[2][1] = q1[3]; b0n[1] = q1[0];
    } else if (MODE & 1) {
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
    b0[0] = b0n[0]; b0[1] = b0n[1];
    av[0] = avn[0]; av[1] = avn[1];
    if (MODE & 32) { avn[0] = avnn[0]; avn[1] = avnn[1]; }
  }
  float sum = 0.f;
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b)
      for (int e = 0; e < 16; ++e) sum += acc[a][b][e];
  if (sum == 12345.678f) out[0] = sum;
}

template <int MODE>
static void run(const float* w, float* out, int iters, const char* what, int shared = 0) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&mix_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int blocks = 256 * 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(mix_kernel<MODE>, dim3(blocks), dim3(768), 100 * 1024, 0, w, out, 64, shared);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(mix_kernel<MODE>, dim3(blocks), dim3(768), 100 * 1024, 0, w, out, iters, shared);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double flops = 2.0 * 32 * 32 * 2 * 16.0 * iters * 12 * blocks;
  printf("mode %2d  %-58s %8.3f ms  %6.1f TFLOP/s\n", MODE, what, ms, flops / (ms * 1e-3) / 1e12);
}

int main() {
  float *w = nullptr, *out = nullptr;
  const size_t nw = (size_t)256 * 4 * 12 * 64 * 64 * 4;
  hipMalloc((void**)&w, nw * sizeof(float));
  hipMalloc((void**)&out, 16);
  hipMemset(w, 0, nw * sizeof(float));
  const int iters = 4096;
  run<0>(w, out, iters, "MFMA only");
  run<1>(w, out, iters, "+ B: 8 ds_read_b32 per tap");
  run<2>(w, out, iters, "+ A: 2 global_load_dwordx4 per tap");
  run<3>(w, out, iters, "+ A global + B b32 (the kernel's tap loop)");
  run<4>(w, out, iters, "+ B: 2 ds_read_b128 per tap");
  run<6>(w, out, iters, "+ A global + B b128");
  run<8>(w, out, iters, "+ A: 2 ds_read_b128 from LDS");
  run<9>(w, out, iters, "+ A LDS + B b32");
  run<12>(w, out, iters, "+ A LDS + B b128");
  run<33>(w, out, iters, "+ A global TWO taps ahead + B b32");
  run<33>(w, out, iters, "+ A global TWO taps ahead (cache hits) + B b32", 1);
  run<3>(w, out, iters, "+ A global (12 slabs per workgroup, L2 hits) + B b32", 2);
  run<33>(w, out, iters, "+ A global TWO taps ahead (12 slabs, L2 hits) + B b32", 2);
  run<2>(w, out, iters, "+ A global, one 64 KB slab for the whole chip (cache hits)", 1);
  run<3>(w, out, iters, "+ A global (cache hits) + B b32", 1);
  run<17>(w, out, iters, "+ A global->LDS DMA ring (cache hits) + ds_read_b128, B b32", 1);
  run<17>(w, out, iters, "+ A global->LDS DMA ring + ds_read_b128, B b32");
  run<20>(w, out, iters, "+ A global->LDS DMA ring + ds_read_b128, B b128");
  return 0;
}
