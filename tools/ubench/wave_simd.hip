// Which SIMD does wave w of a 512-thread workgroup run on?  (HW_ID bits 5:4 = SIMD_ID, 3:0 = WAVE_ID.)
// Build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/wave_simd tools/ubench/wave_simd.hip && /tmp/wave_simd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(512) k(unsigned* rec) {
  if ((threadIdx.x & 63) == 0) rec[blockIdx.x * 8 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
}

int main() {
  const int N = 512;
  unsigned* d;
  hipMalloc(&d, N * 32);
  hipMemset(d, 0, N * 32);
  hipLaunchKernelGGL(k, dim3(N), dim3(512), 0, 0, d);
  hipDeviceSynchronize();
  std::vector<unsigned> h(N * 8);
  hipMemcpy(h.data(), d, N * 32, hipMemcpyDeviceToHost);
  int hist[8][4] = {};
  for (int i = 0; i < N; ++i)
    for (int w = 0; w < 8; ++w) ++hist[w][(h[i * 8 + w] >> 4) & 3];
  for (int w = 0; w < 8; ++w) printf("wave %d: simd 0/1/2/3 = %d %d %d %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  for (int i = 0; i < 4; ++i) {
    printf("wg %d:", i);
    for (int w = 0; w < 8; ++w) printf(" %u", (h[i * 8 + w] >> 4) & 3);
    printf("\n");
  }
  return 0;
}
