// Does an out-of-range buffer_load ... lds write zeros into LDS (or leave the slot alone)?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* a, float* out, int n) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* f = reinterpret_cast<float*>(smem);
  for (int i = threadIdx.x; i < 256; i += 64) f[i] = -7.f;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a), 0, n * 4, 0x00020000);
  const unsigned off = (threadIdx.x & 1) ? 0x80000000u : threadIdx.x * 16u;   // odd lanes: out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)smem, 16, off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = f[i];
}
int main() {
  float h[256], *d, *o;
  for (int i = 0; i < 256; ++i) h[i] = 100.f + i;
  hipMalloc(&d, 1024); hipMalloc(&o, 1024);
  hipMemcpy(d, h, 1024, hipMemcpyHostToDevice);
  k<<<1, 64, 4096>>>(d, o, 256);
  hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
  for (int i = 0; i < 24; ++i) printf("%g ", h[i]);
  printf("\n");
  return 0;
}
