// Timings of k_gemm_nn2 (csrc/gemm_nn2.hip) alone, HIP events, on the NN shapes of the 320-subgraph batch: the six-MFMA form as the
// library launches it and the three-MFMA form (4-wave blocks).  With -DQAGNN_NN2_ABL=<bits> (see the kernel source)
// the 4-wave kernels drop parts of their k-loop: numerically wrong, timing only.  Build: tools/build_micro.sh, run: tools/bin/nn2_ablate_<bits>.
#include <hip/hip_runtime.h>
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

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } \
  } while (0)

static float* dev_rand(size_t n, unsigned seed) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    h[i] = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f;
  }
  float* d;
  CK(hipMalloc(&d, n * 4));
  CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  return d;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 64000;
  struct Shape { const char* name; int K1, K2, No; } shapes[] = {{"mlp 208->208", 208, 0, 208}, {"proj [208|112]->624", 208, 112, 624},
                                                                 {"dX 624->208", 624, 0, 208}, {"dS 624->112", 624, 0, 112}};
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("QAGNN_NN2_ABL=%d M=%d\n", QAGNN_NN2_ABL, M);
  for (auto& sh : shapes) {
    float* A1 = dev_rand((size_t)M * sh.K1, 1);
    float* A2 = sh.K2 ? dev_rand((size_t)M * sh.K2, 2) : nullptr;
    float* B1n = dev_rand((size_t)sh.No * sh.K1, 3);
    float* B2n = sh.K2 ? dev_rand((size_t)sh.No * sh.K2, 4) : nullptr;
    float* C;
    CK(hipMalloc(&C, (size_t)M * sh.No * 4));
    qagnn_gemm_nn_args a = {};
    a.A1 = A1; a.lda1 = sh.K1; a.K1 = sh.K1;
    a.A2 = A2; a.lda2 = sh.K2; a.K2 = sh.K2;
    a.C = C; a.ldc = sh.No; a.M = M; a.No = sh.No;
    const int nt = sh.No >= 208 ? 13 : 7;
    // the operand maxima of the three-MFMA form (rand in [-1, 1))
    uint32_t* am;
    CK(hipMalloc(&am, 16));
    const float one = 1.0f;
    uint32_t w4[4];
    for (int i = 0; i < 4; ++i) memcpy(&w4[i], &one, 4);
    CK(hipMemcpy(am, w4, 16, hipMemcpyHostToDevice));
    const int NJ = (sh.No + 15) / 16;
    struct Form { const char* name; int np; int wv; } forms[] = {{"six MFMAs, library's choice", 3, 0}, {"three MFMAs, 4-wave blocks", 2, 4},
                                                          {"one MFMA (reduced precision)", 1, 4}};
    for (auto& f : forms) {
      void* ws;
      CK(hipMalloc(&ws, qagnn::nn2_pack_bytes(sh.No, sh.K1, sh.K2, f.np)));
      qagnn_gemm_nn_args b = a;
      if (f.np <= 2) { b.a_amax1 = am; b.a_amax2 = sh.K2 ? am + 1 : nullptr; }
      qagnn::launch_nn2_packed(nt, b, B1n, sh.K1, B2n, sh.K2, ws, st, f.np);  // (packs the image)
      auto run = [&] {
        if (f.wv == 0) return qagnn::launch_nn2_prepacked(nt, b, ws, st, 3);
        if (f.np == 1)
          return nt == 13 ? qagnn::nn2::launch_nt<13, 1, true>(b, (const float*)ws, NJ, nullptr, 0, st) : qagnn::nn2::launch_nt<7, 1, true>(b, (const float*)ws, NJ, nullptr, 0, st);
        return nt == 13 ? qagnn::nn2::launch_nt<13, 2, true>(b, (const float*)ws, NJ, nullptr, 0, st) : qagnn::nn2::launch_nt<7, 2, true>(b, (const float*)ws, NJ, nullptr, 0, st);
      };
      for (int i = 0; i < 3; ++i) run();
      CK(hipStreamSynchronize(st));
      const int reps = 30;
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) run();
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / reps;
      printf("  %-22s %-30s %8.1f us  %7.1f TFLOP/s fp32-eq\n", sh.name, f.name, us, 2.0 * M * (sh.K1 + sh.K2) * sh.No / us / 1e6);
      CK(hipFree(ws));
    }
    CK(hipFree(am));
    CK(hipFree(A1)); CK(hipFree(B1n)); CK(hipFree(C));
    if (A2) CK(hipFree(A2));
    if (B2n) CK(hipFree(B2n));
  }
  return 0;
}
