#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdarg.h>
#include "../qagnn_amd/csrc/common.h"
namespace qagnn { void set_error(const char* fmt, ...) {} }
using namespace qagnn;
__global__ void k_split(const float* x, float s, float* rec, unsigned* hl) {
  const int i = threadIdx.x;
  uint32_t hi, lo;
  split2(x[2 * i], x[2 * i + 1], s, hi, lo);
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 h = __builtin_bit_cast(h2, hi), l = __builtin_bit_cast(h2, lo);
  rec[2 * i] = ((float)h[0] + (float)l[0]) / s;
  rec[2 * i + 1] = ((float)h[1] + (float)l[1]) / s;
  hl[2 * i] = hi; hl[2 * i + 1] = lo;
}
// C = A B^T with A[16][32], B[16][32] via one f16 MFMA in the kernel's fragment layout; checks the lane mapping
__global__ void k_mfma(const float* A, const float* B, float* C, int bf) {
  const int l = threadIdx.x, row = l & 15, c = l >> 4;
  u32x4s fa, fb;
  for (int e = 0; e < 8; e += 2) {
    uint32_t h, lo_;
    if (bf) {
      uint32_t u0 = __builtin_bit_cast(uint32_t, A[row * 32 + c * 8 + e]) & 0xFFFF0000u, u1 = __builtin_bit_cast(uint32_t, A[row * 32 + c * 8 + e + 1]) & 0xFFFF0000u;
      fa[e >> 1] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
      u0 = __builtin_bit_cast(uint32_t, B[row * 32 + c * 8 + e]) & 0xFFFF0000u; u1 = __builtin_bit_cast(uint32_t, B[row * 32 + c * 8 + e + 1]) & 0xFFFF0000u;
      fb[e >> 1] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    } else {
      split2(A[row * 32 + c * 8 + e], A[row * 32 + c * 8 + e + 1], 1.f, h, lo_); fa[e >> 1] = h;
      split2(B[row * 32 + c * 8 + e], B[row * 32 + c * 8 + e + 1], 1.f, h, lo_); fb[e >> 1] = h;
    }
  }
  f32x4s acc = {0, 0, 0, 0};
  if (bf) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(QAGNN_BF(fa), QAGNN_BF(fb), acc, 0, 0, 0);
  else acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(QAGNN_HF(fa), QAGNN_HF(fb), acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];  // C[m][n]: row = A row, col = B row
}
int main() {
  float hx[128], *dx, *dr; unsigned* dh;
  for (int i = 0; i < 128; ++i) hx[i] = (i % 2 ? -1.f : 1.f) * (0.001f + i * 0.0371f);
  hipMalloc(&dx, 512); hipMalloc(&dr, 512); hipMalloc(&dh, 512);
  hipMemcpy(dx, hx, 512, hipMemcpyHostToDevice);
  k_split<<<1, 64>>>(dx, 4096.f, dr, dh);
  float hr[128]; unsigned hh[128];
  hipMemcpy(hr, dr, 512, hipMemcpyDeviceToHost); hipMemcpy(hh, dh, 512, hipMemcpyDeviceToHost);
  double w = 0;
  for (int i = 0; i < 128; ++i) { double e = fabs(hr[i] - hx[i]) / fabs(hx[i]); if (e > w) w = e; }
  printf("split2 worst rel err %.3e   x0 %g rec %g hi %08x lo %08x\n", w, hx[0], hr[0], hh[0], hh[1]);
  float A[512], B[512], C[256], *dA, *dB, *dC;
  for (int i = 0; i < 512; ++i) { A[i] = (float)((i * 7) % 11 - 5); B[i] = (float)((i * 5) % 13 - 6); }
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 1024);
  hipMemcpy(dA, A, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B, 2048, hipMemcpyHostToDevice);
  for (int bf = 0; bf < 2; ++bf) {
    k_mfma<<<1, 64>>>(dA, dB, dC, bf);
    hipMemcpy(C, dC, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { float r = 0; for (int k = 0; k < 32; ++k) r += A[m * 32 + k] * B[n * 32 + k]; if (r != C[m * 16 + n]) { if (bad < 4) printf("  %s m %d n %d got %g want %g\n", bf ? "bf16" : "f16", m, n, C[m * 16 + n], r); ++bad; } }
    printf("%s mfma mismatches: %d\n", bf ? "bf16" : "f16", bad);
  }
  return 0;
}
