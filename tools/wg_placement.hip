// Diagnostic: which workgroups of a 2-per-CU persistent grid share a CU on gfx950?
//   hipcc --offload-arch=gfx950 -O2 tools/wg_placement.hip -o tools/wg_placement && tools/wg_placement
// Every workgroup (256 threads, 70 KB of LDS so that two fit a CU) records HW_ID (se / sh / cu) and XCC_ID and spins
// long enough for the whole grid to be resident.  Prints, per physical CU, the blockIdx values it received.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <tuple>

__global__ __launch_bounds__(256) void probe(unsigned* out, int spin) {
  extern __shared__ unsigned char smem[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  double x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 0.999 + 1e-3;
  smem[threadIdx.x] = (unsigned char)x;
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

int main() {
  const int grid = 512;
  unsigned* d; hipMalloc(&d, grid * 2 * sizeof(unsigned));
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 70 * 1024);
  hipLaunchKernelGGL(probe, dim3(grid), dim3(256), 70 * 1024, 0, d, 200000);
  std::vector<unsigned> h(grid * 2);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  std::map<std::tuple<unsigned, unsigned, unsigned, unsigned>, std::vector<int>> cu;
  for (int g = 0; g < grid; ++g) {
    const unsigned hw = h[2 * g], xcc = h[2 * g + 1] & 0xF;
    cu[{xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15}].push_back(g);
  }
  printf("%zu distinct (xcc, se, sh, cu)\n", cu.size());
  int shown = 0;
  std::map<int, int> diffs;
  for (auto& kv : cu) {
    if (shown++ < 24) {
      printf("xcc %u se %u sh %u cu %2u :", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first));
      for (int g : kv.second) printf(" %d", g);
      printf("\n");
    }
    if (kv.second.size() == 2) diffs[kv.second[1] - kv.second[0]]++;
  }
  printf("index distance of the two workgroups of a CU: ");
  for (auto& kv : diffs) printf("%d x%d  ", kv.first, kv.second);
  printf("\n");
  return 0;
}
