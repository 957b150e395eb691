// Ablation / variant harness for k_gemm_nn and k_gemm_tn (no torch): times the (M = 64000, K, No) shapes of the stack with
// hipEvents.  Build with -DQAGNN_ABLATE_NOGLOAD / _NOMMA / _NOEPI to see which phase bounds a kernel; launch shapes are
// chosen with the library's own switches (QAGNN_NN_PERSIST, QAGNN_NN_BLOCKS_PER_CU, QAGNN_TN_CHUNK).
#include "../qagnn_amd/csrc/gemm.hip"
#include <vector>
namespace qagnn { void set_error(const char*, ...) {} }
int main() {
  const int M = 64000;
  struct Shape { int K1, K2, No; const char* name; } shapes[] = {{208, 0, 208, "nn mlp 208x208"}, {624, 0, 208, "nn dX 624->208"},
                                                                 {208, 112, 624, "nn node_proj 320->624"}};
  struct TShape { int Ka, No; const char* name; } tshapes[] = {{208, 208, "tn 208x208"}, {208, 624, "tn 208x624"}, {112, 624, "tn 112x624"},
                                                               {624, 208, "tn 624x208"}, {1024, 208, "tn 1024x208"}};
  float *A1, *A2, *B1, *B2, *Cc, *W;
  hipMalloc(&A1, (size_t)M * 1024 * 4); hipMalloc(&A2, (size_t)M * 112 * 4);
  hipMalloc(&B1, (size_t)M * 624 * 4); hipMalloc(&B2, 112 * 624 * 4); hipMalloc(&Cc, (size_t)M * 624 * 4);
  hipMalloc(&W, (size_t)qagnn_gemm_tn_workspace_elems(M, 1024, 624) * 4);
  hipMemset(A1, 0x3c, (size_t)M * 1024 * 4); hipMemset(A2, 0x3c, (size_t)M * 112 * 4);
  hipMemset(B1, 0x3c, (size_t)M * 624 * 4); hipMemset(B2, 0x3c, 112 * 624 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 20;
  for (auto& s : shapes) {
    qagnn_gemm_nn_args a = {};
    a.A1 = A1; a.lda1 = s.K1; a.K1 = s.K1; a.B1 = B1; a.ldb1 = s.No;
    if (s.K2) { a.A2 = A2; a.lda2 = s.K2; a.K2 = s.K2; a.B2 = B2; a.ldb2 = s.No; }
    a.C = Cc; a.ldc = s.No; a.M = M; a.No = s.No;
    for (int i = 0; i < 3; ++i) qagnn_gemm_nn_f32(&a, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) qagnn_gemm_nn_f32(&a, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double gf = 2.0 * M * (s.K1 + s.K2) * s.No / 1e9;
    printf("%-24s %8.1f us  %6.1f TFLOP/s\n", s.name, ms / reps * 1e3, gf / (ms / reps));
  }
  for (auto& s : tshapes) {
    for (int i = 0; i < 3; ++i) qagnn_gemm_tn_f32(A1, s.Ka, B1, s.No, Cc, s.No, M, s.Ka, s.No, nullptr, nullptr, nullptr, 0, W, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) qagnn_gemm_tn_f32(A1, s.Ka, B1, s.No, Cc, s.No, M, s.Ka, s.No, nullptr, nullptr, nullptr, 0, W, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double gf = 2.0 * M * s.Ka * s.No / 1e9;
    printf("%-24s %8.1f us  %6.1f TFLOP/s  (incl. k_sum_chunks)\n", s.name, ms / reps * 1e3, gf / (ms / reps));
  }
  return 0;
}
