// Timing ablations of k_gemm_nn2 (csrc/gemm_nn2.hip): the kernel is compiled into this program with -DQAGNN_NN2_ABL=<bits> (see the
// kernel source) and timed alone on the GPU with HIP events on the projection and mlp shapes of the 320-subgraph batch.
// Numerically wrong for ABL != 0; timing only.  Build: tools/build_micro.sh, run: tools/bin/nn2_ablate_<bits>.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
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
    {  // B packed once (k_pack_b), then the DMA-fed kernel: pack + product, and the product alone
      void* ws;
      const int64_t wsb = qagnn::nn2_pack_bytes(sh.No, sh.K1, sh.K2);
      CK(hipMalloc(&ws, wsb));
      const int NJ = (sh.No + 15) / 16;
      for (int alone = 0; alone < 3; ++alone) {  // 2: the staggered 8-wave block
        for (int i = 0; i < 3; ++i) qagnn::launch_nn2_packed(nt, a, B1n, sh.K1, B2n, sh.K2, ws, st);
        CK(hipStreamSynchronize(st));
        const int reps = 30;
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < reps; ++i) {
          if (!alone) qagnn::launch_nn2_packed(nt, a, B1n, sh.K1, B2n, sh.K2, ws, st);
          else if (alone == 2 && nt == 13) qagnn::nn2::launch_nt<13, 0, true, 8>(a, (const float*)ws, NJ, nullptr, 0, st);
          else if (alone == 2) continue;  // (the staggered block is built for 13 and 8 column tiles)
          else if (nt == 13) qagnn::nn2::launch_nt<13, 0, true>(a, (const float*)ws, NJ, nullptr, 0, st);
          else qagnn::nn2::launch_nt<7, 0, true>(a, (const float*)ws, NJ, nullptr, 0, st);
        }
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        printf("  %-22s %s  %8.1f us  %7.1f TFLOP/s fp32-eq\n", sh.name, alone == 2 ? "staggered, product only" : alone ? "packed, product only" : "packed, pack+product", us,
               2.0 * M * (sh.K1 + sh.K2) * sh.No / us / 1e6);
      }
      CK(hipFree(ws));
    }
    for (int order = 0; order < 2; ++order) {
      for (int i = 0; i < 3; ++i) {
        int rc = order ? qagnn::nn2::launch_nt<13, 1>(a, B1n, sh.K1, B2n, sh.K2, st) : qagnn::nn2::launch_nt<13, 0>(a, B1n, sh.K1, B2n, sh.K2, st);
        if (nt == 7) rc = order ? qagnn::nn2::launch_nt<7, 1>(a, B1n, sh.K1, B2n, sh.K2, st) : qagnn::nn2::launch_nt<7, 0>(a, B1n, sh.K1, B2n, sh.K2, st);
        if (rc) return 1;
      }
      CK(hipStreamSynchronize(st));
      const int reps = 30;
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < reps; ++i) {
        if (nt == 13) order ? qagnn::nn2::launch_nt<13, 1>(a, B1n, sh.K1, B2n, sh.K2, st) : qagnn::nn2::launch_nt<13, 0>(a, B1n, sh.K1, B2n, sh.K2, st);
        else order ? qagnn::nn2::launch_nt<7, 1>(a, B1n, sh.K1, B2n, sh.K2, st) : qagnn::nn2::launch_nt<7, 0>(a, B1n, sh.K1, B2n, sh.K2, st);
      }
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / reps;
      printf("  %-22s order %d  %8.1f us  %7.1f TFLOP/s fp32-eq\n", sh.name, order, us, 2.0 * M * (sh.K1 + sh.K2) * sh.No / us / 1e6);
    }
    CK(hipFree(A1)); CK(hipFree(B1n)); CK(hipFree(C));
    if (A2) CK(hipFree(A2));
    if (B2n) CK(hipFree(B2n));
  }
  return 0;
}
