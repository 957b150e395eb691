// Micro-benchmark: sustained rate of v_mfma_f32_16x16x4_f32 (the instruction of csrc/gemm.hip) on this chip, with the
// accumulator count and wave count of the GEMM kernel (26 accumulators per wave, 2 waves per SIMD).  Gives the
// realistic ceiling to price k_gemm_nn / k_gemm_tn against (the 157 TF spec assumes 2.4 GHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x4 acc[NACC];
  for (int j = 0; j < NACC; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
    a += 1e-9f;
  }
  float s = 0.f;
  for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu) {
    const int grid = 256 * blocks_per_cu, iters = 20000;
    k<26><<<grid, 256>>>(out, 100, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<26><<<grid, 256>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * 26 * iters * 2.0 * 16 * 16 * 4;
    printf("mfma_f32_16x16x4: %d waves/SIMD, %.1f TFLOP/s (%.3f ms)\n", blocks_per_cu, flop / ms / 1e9, ms);
  }
  return 0;
}
