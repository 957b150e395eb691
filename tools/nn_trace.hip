// Per-phase cycle stamps of one block of k_gemm_nn_split (build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DQAGNN_NN_TRACE=<block>)
#include "../qagnn_amd/csrc/gemm_split.hip"
#include <vector>
namespace qagnn { void set_error(const char*, ...) {} }
int main() {
  const int M = 64000;
  struct Shape { int K1, No; const char* name; } shapes[] = {{208, 208, "mlp 208->208"}, {624, 208, "dX 624->208"}};
  float *A1, *B1, *Cc;
  hipMalloc(&A1, (size_t)M * 624 * 4); hipMalloc(&B1, 624 * 624 * 4); hipMalloc(&Cc, (size_t)M * 624 * 4);
  hipMemset(A1, 0x3c, (size_t)M * 624 * 4); hipMemset(B1, 0x3c, 624 * 624 * 4);
  for (auto& s : shapes) {
    qagnn_gemm_nn_args a = {};
    a.A1 = A1; a.lda1 = s.K1; a.K1 = s.K1; a.C = Cc; a.ldc = s.No; a.M = M; a.No = s.No;
    for (int i = 0; i < 5; ++i) qagnn_gemm_nn_split_f32(&a, B1, s.K1, nullptr, 0, 0);
    hipDeviceSynchronize();
    static unsigned long long tr[4][40][7];
    hipMemcpyFromSymbol(tr, HIP_SYMBOL(qagnn::g_nn_trace), sizeof(tr));
    const int nkt = (s.K1 + 31) / 32;
    printf("%s: block %d, cycles per k-tile (mean over tiles 1..%d): wave | wait barrier A | wait loads | split + LDS store | wait barrier B | issue loads | fragment reads + MFMAs | total\n",
           s.name, QAGNN_NN_TRACE, nkt - 1);
    for (int w = 0; w < 4; ++w) {
      double d[6] = {0, 0, 0, 0, 0, 0}, tot = 0;
      const int order[7] = {0, 1, 6, 2, 3, 4, 5};
      for (int kt = 1; kt < nkt; ++kt) {
        for (int p = 0; p < 6; ++p) d[p] += (double)(tr[w][kt][order[p + 1]] - tr[w][kt][order[p]]);
        tot += (double)(tr[w][kt][5] - tr[w][kt - 1][5]);
      }
      printf("  wave %d  %8.0f %8.0f %8.0f %8.0f %8.0f %8.0f   %8.0f\n", w, d[0] / (nkt - 1), d[1] / (nkt - 1), d[2] / (nkt - 1), d[3] / (nkt - 1), d[4] / (nkt - 1), d[5] / (nkt - 1), tot / (nkt - 1));
    }
  }
  return 0;
}
