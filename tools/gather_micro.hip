// Micro-benchmark: how fast can one MI355X gather 832-byte feature rows (52 lanes x float4) out of L2 / MALL / HBM?
//
// The edge kernels of csrc/edge_attn.hip spend their time in exactly this access: per edge, one wave reads the head-padded
// row of a neighbour node (K[tgt], M[src], Q[src], G[tgt]) and the row of the edge's class table (Ek[c], Em[c]).  The A/B
// of run 56 showed every edge kernel at ~33 us per row gather per edge (E' = 460 800, i.e. ~11 TB/s), whatever the wave
// shape.  This program measures that rate in isolation -- balanced work (one wave per 64 edges), no softmax, no segment
// logic -- against the knobs a kernel has: rows in flight per wave (U), node rows only vs node + class-table rows, all 64
// lanes vs 52 active, subgraph-local vs batch-wide indices.  It prices the edge kernels against what the memory system
// delivers for this pattern rather than against the HBM number on the data sheet.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

__device__ __forceinline__ int xcd_remap(int b, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7;
  const int xcd = b & 7, slot = b >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

// wave w sums the rows idx[64w .. 64w+63] (and, with CLS, the class rows cidx[..]) and writes one row
template <int U, bool CLS, int LANES>
__global__ __launch_bounds__(256) void k_gather(const int* __restrict__ idx, const int* __restrict__ cidx, const float* __restrict__ A,
                                                int lda, int off, const float* __restrict__ T, int ldt, float* __restrict__ out, int E) {
  const int w = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6));
  const int c0 = w * 64;
  if (c0 >= E) return;
  const int lane = threadIdx.x & 63;
  const bool act = lane < LANES;
  const int cnt = min(64, E - c0);
  const int iv = idx[c0 + min(lane, cnt - 1)];
  const int cv = CLS ? cidx[c0 + min(lane, cnt - 1)] : 0;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = 0; i < cnt; i += U) {
    float4 r[U], c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = min(i + u, cnt - 1);
      const int s = __builtin_amdgcn_readlane(iv, k);
      r[u] = act ? *reinterpret_cast<const float4*>(A + (int64_t)s * lda + off + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (CLS) {
        const int cc = __builtin_amdgcn_readlane(cv, k);
        c[u] = act ? *reinterpret_cast<const float4*>(T + (int64_t)cc * ldt + lane * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc.x += r[u].x; acc.y += r[u].y; acc.z += r[u].z; acc.w += r[u].w;
      if (CLS) { acc.x += c[u].x; acc.y += c[u].y; acc.z += c[u].z; acc.w += c[u].w; }
    }
  }
  if (act) *reinterpret_cast<float4*>(out + (int64_t)w * 256 + lane * 4) = acc;
}

struct Case {
  const char* name;
  void (*launch)(const int*, const int*, const float*, int, int, const float*, int, float*, int, hipStream_t);
  int rows_per_edge, lanes;
};
template <int U, bool CLS, int LANES>
static void go(const int* idx, const int* cidx, const float* A, int lda, int off, const float* T, int ldt, float* out, int E, hipStream_t s) {
  const int waves = (E + 63) / 64;
  k_gather<U, CLS, LANES><<<(waves + 3) / 4, 256, 0, s>>>(idx, cidx, A, lda, off, T, ldt, out, E);
}

int main() {
  const int N = 64000, n = 200, E = 460800, C = 612;
  const int lda = 624, ldt = 416;  // K|M|Q rows and Ek|Em rows of the stack at d = 200
  std::vector<int> h_local(E), h_wide(E), h_cls(E);
  srand(7);
  for (int e = 0; e < E; ++e) {
    const int blk = (int)((int64_t)e * (N / n) / E);   // edges in subgraph order, like a CSR of the batched graph
    h_local[e] = blk * n + rand() % n;                 // neighbour inside the same subgraph (what the edge kernels see)
    h_wide[e] = (int)(((int64_t)rand() * 32768 + rand()) % N);  // anywhere in the batch (what the class pass sees)
    const int r = rand() % 100;
    h_cls[e] = r < 14 ? C - 1 - rand() % 4 : (r < 60 ? rand() % 40 : rand() % C);  // skewed like real class counts
  }
  int *d_local, *d_wide, *d_cls;
  float *A, *T, *out;
  hipMalloc(&d_local, E * 4); hipMalloc(&d_wide, E * 4); hipMalloc(&d_cls, E * 4);
  hipMalloc(&A, (size_t)N * 1024 * 4);  // also holds the 1024-float-pitch variant
  hipMalloc(&T, (size_t)C * ldt * 4);
  hipMalloc(&out, (size_t)(E / 64 + 8) * 256 * 4);
  hipMemcpy(d_local, h_local.data(), E * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_wide, h_wide.data(), E * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_cls, h_cls.data(), E * 4, hipMemcpyHostToDevice);
  hipMemset(A, 0, (size_t)N * 1024 * 4);
  hipMemset(T, 0, (size_t)C * ldt * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const Case cases[] = {
      {"U=2  node row", go<2, false, 52>, 1, 52},        {"U=4  node row", go<4, false, 52>, 1, 52},
      {"U=8  node row", go<8, false, 52>, 1, 52},        {"U=16 node row", go<16, false, 52>, 1, 52},
      {"U=4  node + class row", go<4, true, 52>, 2, 52}, {"U=8  node + class row", go<8, true, 52>, 2, 52},
      {"U=4  node row, 64 lanes (1 KB)", go<4, false, 64>, 1, 64}, {"U=8  node row, 64 lanes (1 KB)", go<8, false, 64>, 1, 64},
  };
  printf("%-34s %-10s %9s %9s %11s\n", "variant", "indices", "us", "TB/s", "us/row-pass");
  for (int wide = 0; wide < 2; ++wide)
    for (const Case& c : cases) {
      const int* idx = wide ? d_wide : d_local;
      const int pitch = c.lanes == 64 ? 1024 : lda;
      for (int it = 0; it < 3; ++it) c.launch(idx, d_cls, A, pitch, c.lanes == 64 ? 0 : 208, T, ldt, out, E, 0);
      hipDeviceSynchronize();
      float best = 1e9f, sum = 0.f;
      const int reps = 10;
      for (int it = 0; it < reps; ++it) {
        hipEventRecord(e0);
        c.launch(idx, d_cls, A, pitch, c.lanes == 64 ? 0 : 208, T, ldt, out, E, 0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
        sum += ms;
      }
      const double bytes = (double)E * c.rows_per_edge * c.lanes * 16;
      printf("%-34s %-10s %9.1f %9.2f %11.1f   (mean %.1f us)\n", c.name, wide ? "batch-wide" : "subgraph", best * 1e3, bytes / best / 1e9,
             best * 1e3 / c.rows_per_edge, sum / reps * 1e3);
    }
  return 0;
}
