// Debug / sanity program for the three-MFMA NN form: product against a host float64 reference, ratio statistics per column tile.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../qagnn_amd/csrc/gemm_nn2.hip"

namespace qagnn {
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}
}  // namespace qagnn
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 9000, K1 = argc > 2 ? atoi(argv[2]) : 208, K2 = argc > 3 ? atoi(argv[3]) : 0, No = argc > 4 ? atoi(argv[4]) : 208;
  const int np = argc > 5 ? atoi(argv[5]) : 2;
  std::vector<float> A1((size_t)M * K1), A2((size_t)M * (K2 ? K2 : 1)), B1((size_t)No * K1), B2((size_t)No * (K2 ? K2 : 1));
  unsigned s = 12345u;
  auto rnd = [&] { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
  const int probe = argc > 6 ? atoi(argv[6]) : 0;
  for (auto& v : A1) v = rnd();
  if (probe) for (int m = 0; m < M; ++m) for (int k = 0; k < K1; ++k) A1[(size_t)m * K1 + k] = (k == m % K1) ? 1.f : 0.f;
  if (probe) for (int n = 0; n < No; ++n) for (int k = 0; k < K1; ++k) B1[(size_t)n * K1 + k] = 0.f;
  for (auto& v : A2) v = rnd();
  for (auto& v : B1) v = rnd() * 0.1f;
  for (auto& v : B2) v = rnd() * 0.1f;
  if (probe) for (int n = 0; n < No; ++n) for (int k = 0; k < K1; ++k) B1[(size_t)n * K1 + k] = (float)(n * 1000 + k);
  float am1 = 0, am2 = 0;
  for (auto v : A1) am1 = fmaxf(am1, fabsf(v));
  for (auto v : A2) am2 = fmaxf(am2, fabsf(v));
  float *dA1, *dA2, *dB1, *dB2, *dC;
  uint32_t* dam;
  CK(hipMalloc(&dA1, A1.size() * 4)); CK(hipMalloc(&dA2, A2.size() * 4)); CK(hipMalloc(&dB1, B1.size() * 4)); CK(hipMalloc(&dB2, B2.size() * 4));
  CK(hipMalloc(&dC, (size_t)M * No * 4)); CK(hipMalloc(&dam, 16));
  CK(hipMemcpy(dA1, A1.data(), A1.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dA2, A2.data(), A2.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB1, B1.data(), B1.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB2, B2.data(), B2.size() * 4, hipMemcpyHostToDevice));
  uint32_t amw[4] = {0, 0, 0, 0};
  memcpy(&amw[0], &am1, 4); memcpy(&amw[1], &am2, 4);
  CK(hipMemcpy(dam, amw, 16, hipMemcpyHostToDevice));
  qagnn_gemm_nn_args a = {};
  a.A1 = dA1; a.lda1 = K1; a.K1 = K1; a.A2 = K2 ? dA2 : nullptr; a.lda2 = K2; a.K2 = K2; a.C = dC; a.ldc = No; a.M = M; a.No = No;
  a.a_amax1 = dam; a.a_amax2 = K2 ? dam + 1 : nullptr;
  void* ws;
  CK(hipMalloc(&ws, qagnn::nn2_pack_bytes(No, K1, K2, np)));
  const int nt = No >= 208 ? 13 : 7;
  int rc = qagnn::launch_nn2_packed(nt, a, dB1, K1, K2 ? dB2 : nullptr, K2, ws, 0, np);
  CK(hipDeviceSynchronize());
  printf("rc %d  M %d K1 %d K2 %d No %d np %d  amax %g %g\n", rc, M, K1, K2, No, np, am1, am2);
  std::vector<float> C((size_t)M * No);
  CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
  if (np == 2) {
    const int NJ = (No + 15) / 16, nkt = qagnn::nn2::walk_tiles(K1, K2);
    std::vector<uint32_t> f(NJ);
    CK(hipMemcpy(f.data(), (char*)ws + ((int64_t)nkt * NJ + 13) * 2048, NJ * 4, hipMemcpyDeviceToHost));
    printf("B fields:");
    for (int j = 0; j < NJ; ++j) printf(" %u", f[j]);
    printf("\n");
  }
  if (probe) {  // C[m][n] should be B1[n][m % K1] = n * 1000 + m % K1
    for (int m = 0; m < 40; ++m) { printf("m %2d:", m); for (int n = 0; n < 6; ++n) printf(" %9.1f", C[(size_t)m * No + n]); printf("\n"); }
  }
  double worst = 0;
  int bad = 0;
  for (int m = 0; m < M; m += 37)
    for (int n = 0; n < No; ++n) {
      double r = 0, ab = 0;
      for (int k = 0; k < K1; ++k) { r += (double)A1[(size_t)m * K1 + k] * B1[(size_t)n * K1 + k]; ab += fabs((double)A1[(size_t)m * K1 + k] * B1[(size_t)n * K1 + k]); }
      for (int k = 0; k < K2; ++k) { r += (double)A2[(size_t)m * K2 + k] * B2[(size_t)n * K2 + k]; ab += fabs((double)A2[(size_t)m * K2 + k] * B2[(size_t)n * K2 + k]); }
      const double e = fabs(C[(size_t)m * No + n] - r) / (ab + 1e-30);
      if (e > worst) worst = e;
      if (e > 2e-6 && bad < 12) { printf("  m %d n %d got %g ref %g ratio %g\n", m, n, C[(size_t)m * No + n], r, C[(size_t)m * No + n] / r); ++bad; }
    }
  printf("worst err / sum|ab| = %.3e (eps32 = 1.2e-7)\n", worst);
  return 0;
}
