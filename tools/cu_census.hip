// Which blocks of a 2-per-CU persistent grid share a CU?  (placement is undefined by contract: speed experiments only)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned* out) {
  extern __shared__ char smem[];
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[blockIdx.x * 2] = hw;
    out[blockIdx.x * 2 + 1] = xcc;
    smem[0] = 1;
  }
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(64);  // keep every block resident while the others arrive
}
int main() {
  const int nb = 512;
  unsigned* d;
  hipMalloc(&d, nb * 8);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 79872);
  k<<<nb, 256, 79872>>>(d);
  std::vector<unsigned> h(nb * 2);
  hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost);
  std::map<unsigned long long, std::vector<int>> cu;
  for (int b = 0; b < nb; ++b) {
    const unsigned hw = h[b * 2], xcc = h[b * 2 + 1] & 0xF;
    const unsigned cu_id = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    cu[((unsigned long long)xcc << 32) | (se << 8) | (sh << 4) | cu_id].push_back(b);
  }
  printf("%zu distinct CUs\n", cu.size());
  int shown = 0, diff256 = 0, other = 0;
  for (auto& kv : cu) {
    if (kv.second.size() == 2 && kv.second[1] - kv.second[0] == 256) ++diff256; else ++other;
    if (shown++ < 12) { printf("xcc %llu key %llx:", kv.first >> 32, kv.first & 0xFFFFFFFF); for (int b : kv.second) printf(" %d", b); printf("\n"); }
  }
  printf("CUs whose two blocks are (b, b + 256): %d, other: %d\n", diff256, other);
  return 0;
}
