// Where do the 8 waves of a 512-thread block with 2 waves per SIMD (256 VGPRs) land?  Prints HW_ID fields per wave for a few blocks and
// the histogram of "waves of a block per SIMD".  Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/simd_probe tools/simd_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_probe(unsigned* out) {
  extern __shared__ unsigned char sm[];
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);  // HW_REG_HW_ID, all 32 bits
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = hw;
  // keep the registers / LDS honest
  if (hw == 0xFFFFFFFFu) sm[threadIdx.x] = 1;
}

int main() {
  const int nb = 512;
  unsigned* d;
  hipMalloc(&d, nb * 8 * 4);
  hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
  k_probe<<<nb, 512, 150000>>>(d);
  std::vector<unsigned> h(nb * 8);
  hipMemcpy(h.data(), d, nb * 8 * 4, hipMemcpyDeviceToHost);
  for (int b = 0; b < 4; ++b) {
    printf("block %d:", b);
    for (int w = 0; w < 8; ++w) {
      const unsigned v = h[b * 8 + w];
      printf("  w%d: wave %u simd %u cu %u sh %u se %u |", w, v & 15, (v >> 4) & 3, (v >> 8) & 15, (v >> 12) & 1, (v >> 13) & 7);
    }
    printf("\n");
  }
  int hist[9] = {}, pair_w4 = 0;
  for (int b = 0; b < nb; ++b) {
    int cnt[4] = {};
    for (int w = 0; w < 8; ++w) cnt[(h[b * 8 + w] >> 4) & 3]++;
    for (int s = 0; s < 4; ++s) hist[cnt[s]]++;
    bool ok = true;
    for (int w = 0; w < 4; ++w) ok = ok && (((h[b * 8 + w] >> 4) & 3) == ((h[b * 8 + w + 4] >> 4) & 3));
    pair_w4 += ok;
  }
  printf("waves of one block on one SIMD: ");
  for (int i = 0; i <= 8; ++i) printf("%d:%d ", i, hist[i]);
  printf("\nblocks in which waves w and w + 4 share a SIMD: %d of %d\n", pair_w4, nb);
  return 0;
}
