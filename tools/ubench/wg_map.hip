// Where does the dispatcher put workgroup i?  Records (XCC_ID, HW_ID, start time) of every workgroup of a 1-D grid of 256-thread
// workgroups that fit twice per CU (64 KB LDS each), and prints, per id range, how many distinct CUs were used and which ids share a CU.
// Build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/wg_map tools/ubench/wg_map.hip && /tmp/wg_map
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>

__global__ void __launch_bounds__(256, 2) k(unsigned* rec, int spin) {
  extern __shared__ float lds[];
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID, all 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID [3:0]
    const unsigned long long t = __builtin_readcyclecounter();
    rec[blockIdx.x * 4 + 0] = hw;
    rec[blockIdx.x * 4 + 1] = xcc;
    rec[blockIdx.x * 4 + 2] = (unsigned)(t >> 8);
  }
  lds[threadIdx.x] = 1.f;
  float v = threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + lds[(threadIdx.x + i) & 255];
  if (v == 1234.5f) rec[0] = 0;
}

int main() {
  const int N = 1536;
  unsigned* d;
  hipMalloc(&d, N * 16);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipMemset(d, 0, N * 16);
    hipLaunchKernelGGL(k, dim3(N), dim3(256), 64 * 1024, 0, d, 200000);
    hipDeviceSynchronize();
  }
  std::vector<unsigned> h(N * 4);
  hipMemcpy(h.data(), d, N * 16, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> by_cu;
  unsigned tmin = ~0u;
  for (int i = 0; i < N; ++i) tmin = h[i * 4 + 2] < tmin ? h[i * 4 + 2] : tmin;
  for (int i = 0; i < N; ++i) {
    const unsigned hw = h[i * 4], xcc = h[i * 4 + 1];
    const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(i);
    if (i < 24 || (i >= 256 && i < 264) || (i >= 512 && i < 520))
      printf("wg %4d: xcc %u se %u sh %u cu %2u simd %u  hw_id %08x  start +%u (x256 cycles)\n", i, xcc, se, sh, cu, (hw >> 4) & 3, hw,
             h[i * 4 + 2] - tmin);
  }
  printf("%zu distinct CUs\n", by_cu.size());
  int shown = 0;
  for (auto& kv : by_cu) {
    if (shown++ >= 12) break;
    printf("cu %05x:", kv.first);
    for (int id : kv.second) printf(" %d", id);
    printf("\n");
  }
  // histogram: for ids < 512 -- how many CUs hold two of them with both ids < 256 / one below and one above
  int both_low = 0, split = 0, both_high = 0;
  for (auto& kv : by_cu) {
    int lo = 0, hi = 0;
    for (int id : kv.second) {
      if (id < 256) ++lo;
      else if (id < 512) ++hi;
    }
    if (lo == 2) ++both_low;
    if (lo == 1 && hi == 1) ++split;
    if (hi == 2) ++both_high;
  }
  printf("first 512 ids: CUs with two ids < 256: %d, one < 256 and one in [256, 512): %d, two in [256, 512): %d\n", both_low, split,
         both_high);
  return 0;
}
