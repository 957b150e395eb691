// Timing ablations of k_gemm_tn_ws / k_gemm_tn_split (csrc/gemm_split.hip compiled into this program with -DQAGNN_TNW_ABL=<bits>): the
// weight-gradient products of the 320-subgraph batch, kernel alone (no chunk sum), HIP events.  Numerically wrong for ABL != 0.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "../qagnn_amd/csrc/gemm_split.hip"

namespace qagnn {
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}
// (the NN half of the translation unit refers to gemm_nn2.hip)
bool nn2_ok(const qagnn_gemm_nn_args&, int, int) { return false; }
bool nn2_packed_ok(const qagnn_gemm_nn_args&, int64_t) { return false; }
int64_t nn2_pack_bytes(int, int, int) { return 0; }
int launch_nn2(int, const qagnn_gemm_nn_args&, const float*, int, const float*, int, hipStream_t) { return 0; }
int launch_nn2_packed(int, const qagnn_gemm_nn_args&, const float*, int, const float*, int, void*, hipStream_t) { return 0; }
const void* nn2_prepack_lookup(const float*, int, int, const float*, int, int, int) { return nullptr; }
int launch_nn2_prepacked(int, const qagnn_gemm_nn_args&, const void*, hipStream_t) { return 0; }
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

int main() {
  const int R = 64000;
  struct Shape { const char* name; int Ka1, Ka2, No; } shapes[] = {{"[X|S]^T dKMQ  208+112 x 624", 208, 112, 624}, {"208 x 624", 208, 0, 624},
                                                                  {"208 x 208", 208, 0, 208}, {"112 x 624", 112, 0, 624}};
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("QAGNN_TNW_ABL=%d R=%d\n", QAGNN_TNW_ABL, R);
  for (auto& sh : shapes) {
    float* A1 = dev_rand((size_t)R * sh.Ka1, 1);
    float* A2 = sh.Ka2 ? dev_rand((size_t)R * sh.Ka2, 2) : nullptr;
    float* B = dev_rand((size_t)R * sh.No, 3);
    const int chunk = sh.Ka2 ? qagnn::tn_split2_chunk_rows(R, sh.Ka1, sh.Ka2, sh.No, 0) : qagnn::tn_split_chunk_rows(R, sh.Ka1, sh.No, 0);
    const int nchunk = (R + chunk - 1) / chunk;
    float* P;
    CK(hipMalloc(&P, (size_t)nchunk * (sh.Ka1 + sh.Ka2) * sh.No * 4));
    for (int ws = 0; ws < 2; ++ws) {
      auto run = [&] {
        if (sh.Ka2) {
          if (ws) { dim3 grid((sh.No + 207) / 208, (sh.Ka1 + 111) / 112 + (sh.Ka2 + 111) / 112, nchunk);
            qagnn::launch_tn_ws_i<7, 13, false>(grid, st, A1, sh.Ka1, B, sh.No, P, R, sh.Ka1, sh.No, nullptr, nullptr, chunk, A2, sh.Ka2, sh.Ka2);
          } else { dim3 grid((sh.No + 207) / 208, (sh.Ka1 + 111) / 112 + (sh.Ka2 + 111) / 112, nchunk);
            qagnn::launch_tn_split_i<7, 13, false>(grid, st, A1, sh.Ka1, B, sh.No, P, R, sh.Ka1, sh.No, nullptr, nullptr, chunk, nullptr, A2, sh.Ka2, sh.Ka2); }
        } else if (sh.Ka1 <= 112) {
          dim3 grid((sh.No + 207) / 208, (sh.Ka1 + 111) / 112, nchunk);
          if (ws) qagnn::launch_tn_ws_i<7, 13, false>(grid, st, A1, sh.Ka1, B, sh.No, P, R, sh.Ka1, sh.No, nullptr, nullptr, chunk);
          else qagnn::launch_tn_split_i<7, 13, false>(grid, st, A1, sh.Ka1, B, sh.No, P, R, sh.Ka1, sh.No, nullptr, nullptr, chunk);
        } else {
          dim3 grid((sh.No + 111) / 112, (sh.Ka1 + 207) / 208, nchunk);
          if (ws) qagnn::launch_tn_ws_i<13, 7, false>(grid, st, A1, sh.Ka1, B, sh.No, P, R, sh.Ka1, sh.No, nullptr, nullptr, chunk);
          else qagnn::launch_tn_split_i<13, 7, false>(grid, st, A1, sh.Ka1, B, sh.No, P, R, sh.Ka1, sh.No, nullptr, nullptr, chunk);
        }
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
      printf("  %-30s %s  chunks %3d x %4d rows  %8.1f us  %7.1f TFLOP/s fp32-eq\n", sh.name, ws ? "k_gemm_tn_ws   " : "k_gemm_tn_split", nchunk, chunk, us,
             2.0 * R * (sh.Ka1 + sh.Ka2) * sh.No / us / 1e6);
    }
    CK(hipFree(A1)); CK(hipFree(B)); CK(hipFree(P));
    if (A2) CK(hipFree(A2));
  }
  return 0;
}
