// The edge kernels of GATConvE: relation-aware multi-head attention over the batched QA subgraphs.
//
// Reference semantics (modeling/modeling_qagnn.py:442, 455-484 + torch-geometric 1.7.0 propagate/softmax +
// torch-scatter 2.0.7 scatter), rewritten as project-then-gather:
//   key_e = K[tgt] + Ek[cls],  msg_e = M[src] + Em[cls],  score_eh = qscale * <Q[src], key_e>_h
//   a_eh  = exp(score_eh - max_out(src)) / (sum_out(src) exp(.) + 1e-16)        (softmax grouped by SOURCE)
//   alpha_eh = deg(src) * a_eh ;   aggr[tgt] += alpha_eh * msg_e                 (sum grouped by TARGET)
//
// Data layout: node rows are head-padded (H = 4 heads x HP floats, HP % 4 == 0, pads zero), so one wave reads a
// row with ONE float4 per lane: lane = 16*head + j owns floats [4j, 4j+4) of its head (lanes with 4j >= HP idle),
// and the per-head dot product is a 4-step butterfly inside a 16-lane DPP row.  One wave walks one node's
// segment; segment entries are staged 64 at a time in registers and broadcast with v_readlane, and the row loads
// of 4 edges are issued back to back so their L2/HBM latencies overlap.
//
// Everything here is HBM/L2-bound gather work: no MFMA, no atomics; every reduction has a fixed order.
#include <stdlib.h>

#include "common.h"

namespace qagnn {

#define EDGE_UNROLL 4
#ifndef SRC1W_UNROLL
#define SRC1W_UNROLL 3  // k_edge_bwd_src1_w gathers three rows per edge: 3 edges in flight keep it at 8 waves per SIMD
#endif

struct Lane {
  int g, j, off;
  bool act;
};
__device__ __forceinline__ Lane lane_info(int HP) {
  const int lane = threadIdx.x & 63;
  Lane L;
  L.g = lane >> 4;
  L.j = lane & 15;
  L.act = L.j * 4 < HP;
  L.off = L.g * HP + L.j * 4;
  return L;
}
// node handled by this wave: 4 waves per block, blocks remapped so an XCD owns a contiguous node range
__device__ __forceinline__ int wave_node() {
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  return __builtin_amdgcn_readfirstlane(lb * 4 + (threadIdx.x >> 6));
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// Per-edge head scalars (a, alpha, ga, gs, score: [E', 4] arrays) travel as LANE-HELD float4: lane i of the wave holds the 4
// head values of edge i of the current 64-edge chunk, read or written with ONE coalesced 16-byte-per-lane instruction per
// chunk.  Measured (profiles/r1_run59_gather_micro.txt + run 56): these kernels are bound by the number of vector-memory
// wave-instructions they issue per edge (~21 us per instruction per edge at E' = 460 800, whether it moves 4 or 832 useful
// bytes -- the texture addresser spends its 16 cycles per wave-instruction either way), so a 4-byte `alpha[e*4+g]` load per
// edge costs as much as a whole feature row.  Moving a value between "lane i, component g" and "the 16 lanes of head group
// g" is register traffic only: v_readlane + v_cndmask.
__device__ __forceinline__ float lane_f(float v, int i) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), i)); }
// value of edge i (wave-uniform) for this lane's head group g
__device__ __forceinline__ float head_get(const float4 v, int i, int g) {
  const float x = lane_f(v.x, i), y = lane_f(v.y, i), z = lane_f(v.z, i), w = lane_f(v.w, i);
  return g == 0 ? x : (g == 1 ? y : (g == 2 ? z : w));
}
// lane i's float4 <- (p of head group 0, 1, 2, 3); p is uniform inside each 16-lane group
__device__ __forceinline__ void head_put(float4& v, int i, float p, int lane) {
  const float x = lane_f(p, 0), y = lane_f(p, 16), z = lane_f(p, 32), w = lane_f(p, 48);
  if (lane == i) v = make_float4(x, y, z, w);
}

// ---------------------------------------------------------------------------------------------------------------
// forward 1/3: raw scores, one wave per SOURCE node (Q row in registers; gathers K[tgt] and Ek[cls])
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_edge_scores(const int* __restrict__ rowptr_s, const int* __restrict__ tgt_s,
                                                     const int* __restrict__ cls_s, const float* __restrict__ KMQ, int ldk,
                                                     const float* __restrict__ EkEm, int lde, int HP, float qscale,
                                                     float* __restrict__ score, int N, const int* __restrict__ gate) {
  if (gate && *gate == 0) return;  // the LDS-resident kernel took this graph (device-side decision)
  const int s = wave_node();
  if (s >= N) return;
  const Lane L = lane_info(HP);
  const int lane = threadIdx.x & 63, DP = 4 * HP;
  const float4 q = L.act ? ld4(KMQ + (int64_t)s * ldk + 2 * DP + L.off) : zero4();
  const int beg = __builtin_amdgcn_readfirstlane(rowptr_s[s]), end = __builtin_amdgcn_readfirstlane(rowptr_s[s + 1]);
  for (int e0 = beg; e0 < end; e0 += 64) {
    const int cnt = min(64, end - e0);
    const int tv = lane < cnt ? tgt_s[e0 + lane] : 0;
    const int cv = lane < cnt ? cls_s[e0 + lane] : 0;
    float4 sc = zero4();  // lane i: the 4 head scores of edge e0 + i
    for (int i = 0; i < cnt; i += EDGE_UNROLL) {
      float4 k[EDGE_UNROLL], ek[EDGE_UNROLL];
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        const int idx = min(i + u, cnt - 1);
        const int t = __builtin_amdgcn_readlane(tv, idx), c = __builtin_amdgcn_readlane(cv, idx);
        k[u] = L.act ? ld4(KMQ + (int64_t)t * ldk + L.off) : zero4();
        ek[u] = L.act ? ld4(EkEm + (int64_t)c * lde + L.off) : zero4();
      }
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        const float p = row16_sum(dot4(q, add4(k[u], ek[u]))) * qscale;
        if (i + u < cnt) head_put(sc, i + u, p, lane);
      }
    }
    if (lane < cnt) st4(score + (int64_t)(e0 + lane) * 4, sc);
  }
}

// forward 2/3: softmax over each source segment (PyG softmax: max, exp, sum, / (sum + 1e-16)), then * out-degree
__global__ __launch_bounds__(256) void k_edge_softmax(const int* __restrict__ rowptr_s, const float* __restrict__ score,
                                                      float* __restrict__ a, float* __restrict__ alpha, int N,
                                                      const int* __restrict__ gate) {
  if (gate && *gate == 0) return;
  const int s = wave_node();
  if (s >= N) return;
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  const int beg = __builtin_amdgcn_readfirstlane(rowptr_s[s]), end = __builtin_amdgcn_readfirstlane(rowptr_s[s + 1]);
  float m = -INFINITY;
  for (int e = beg + j; e < end; e += 16) m = fmaxf(m, score[(int64_t)e * 4 + g]);
  m = row16_max(m);
  float sum = 0.f;
  for (int e = beg + j; e < end; e += 16) sum += expf(score[(int64_t)e * 4 + g] - m);
  sum = row16_sum(sum);
  const float deg = (float)(end - beg);
  for (int e = beg + j; e < end; e += 16) {
    const float av = expf(score[(int64_t)e * 4 + g] - m) / (sum + 1e-16f);
    a[(int64_t)e * 4 + g] = av;
    alpha[(int64_t)e * 4 + g] = av * deg;
  }
}

// forward 3/3: weighted sum of messages, one wave per TARGET node (gathers M[src] and Em[cls]; no atomics)
__global__ __launch_bounds__(256) void k_edge_aggregate(const int* __restrict__ rowptr_t, const int* __restrict__ src_t,
                                                        const int* __restrict__ cls_t, const int* __restrict__ pos_t,
                                                        const float* __restrict__ KMQ, int ldk, const float* __restrict__ EkEm,
                                                        int lde, int HP, const float* __restrict__ alpha,
                                                        float* __restrict__ aggr, int lda, int N, const int* __restrict__ gate) {
  if (gate && *gate == 0) return;
  const int t = wave_node();
  if (t >= N) return;
  const Lane L = lane_info(HP);
  const int lane = threadIdx.x & 63, DP = 4 * HP;
  const int beg = __builtin_amdgcn_readfirstlane(rowptr_t[t]), end = __builtin_amdgcn_readfirstlane(rowptr_t[t + 1]);
  float4 acc = zero4();
  for (int e0 = beg; e0 < end; e0 += 64) {
    const int cnt = min(64, end - e0);
    const int sv = lane < cnt ? src_t[e0 + lane] : 0;
    const int cv = lane < cnt ? cls_t[e0 + lane] : 0;
    const int pv = lane < cnt ? pos_t[e0 + lane] : 0;
    const float4 al4 = lane < cnt ? ld4(alpha + (int64_t)pv * 4) : zero4();  // lane i: alpha of edge e0 + i (one gather per chunk)
    for (int i = 0; i < cnt; i += EDGE_UNROLL) {
      float4 m[EDGE_UNROLL], em[EDGE_UNROLL];
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        const int idx = min(i + u, cnt - 1);
        const int s = __builtin_amdgcn_readlane(sv, idx), c = __builtin_amdgcn_readlane(cv, idx);
        m[u] = L.act ? ld4(KMQ + (int64_t)s * ldk + DP + L.off) : zero4();
        em[u] = L.act ? ld4(EkEm + (int64_t)c * lde + DP + L.off) : zero4();
      }
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        const float wgt = i + u < cnt ? head_get(al4, min(i + u, cnt - 1), L.g) : 0.f;
        acc = fma4(wgt, add4(m[u], em[u]), acc);
      }
    }
  }
  if (L.act) st4(aggr + (int64_t)t * lda + L.off, acc);
}

// ---------------------------------------------------------------------------------------------------------------
// backward.  G = d aggr.
//   src pass 1: dM[s] = sum_out alpha_e G[t];  ga_e = deg * <M[s]+Em[c], G[t]>_h;  rs[s] = sum_out a_e ga_e
//   src pass 2: gs_e = qscale * a_e (ga_e - rs[s]);  dQ[s] = sum_out gs_e (K[t]+Ek[c]);  ga <- gs (in place)
//   tgt pass  : dK[t] = sum_in gs_e Q[s]
//   cls pass  : dEk[c] = sum_{e in c} gs_e Q[s],  dEm[c] = sum_{e in c} alpha_e G[t]   (chunk partials, then ordered sum)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_edge_bwd_src1(const int* __restrict__ rowptr_s, const int* __restrict__ tgt_s,
                                                       const int* __restrict__ cls_s, const float* __restrict__ KMQ, int ldk,
                                                       const float* __restrict__ EkEm, int lde, int HP,
                                                       const float* __restrict__ a, const float* __restrict__ alpha,
                                                       const float* __restrict__ G, int ldg, float* __restrict__ dKMQ,
                                                       float* __restrict__ ga, float* __restrict__ rs, int N) {
  const int s = wave_node();
  if (s >= N) return;
  const Lane L = lane_info(HP);
  const int lane = threadIdx.x & 63, DP = 4 * HP;
  const float4 mrow = L.act ? ld4(KMQ + (int64_t)s * ldk + DP + L.off) : zero4();
  const int beg = __builtin_amdgcn_readfirstlane(rowptr_s[s]), end = __builtin_amdgcn_readfirstlane(rowptr_s[s + 1]);
  const float deg = (float)(end - beg);
  float4 dM = zero4();
  float r = 0.f;
  for (int e0 = beg; e0 < end; e0 += 64) {
    const int cnt = min(64, end - e0);
    const int tv = lane < cnt ? tgt_s[e0 + lane] : 0;
    const int cv = lane < cnt ? cls_s[e0 + lane] : 0;
    // lane i: a, alpha of edge e0 + i (coalesced, once per chunk) and, on the way out, its ga
    const float4 a4 = lane < cnt ? ld4(a + (int64_t)(e0 + lane) * 4) : zero4();
    const float4 al4 = lane < cnt ? ld4(alpha + (int64_t)(e0 + lane) * 4) : zero4();
    float4 ga4 = zero4();
    for (int i = 0; i < cnt; i += EDGE_UNROLL) {
      float4 g4[EDGE_UNROLL], em[EDGE_UNROLL];
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        const int idx = min(i + u, cnt - 1);
        const int t = __builtin_amdgcn_readlane(tv, idx), c = __builtin_amdgcn_readlane(cv, idx);
        g4[u] = L.act ? ld4(G + (int64_t)t * ldg + L.off) : zero4();
        em[u] = L.act ? ld4(EkEm + (int64_t)c * lde + DP + L.off) : zero4();
      }
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        const bool ok = i + u < cnt;
        const int idx = min(i + u, cnt - 1);
        const float al = ok ? head_get(al4, idx, L.g) : 0.f, av = ok ? head_get(a4, idx, L.g) : 0.f;
        dM = fma4(al, g4[u], dM);
        const float gae = deg * row16_sum(dot4(add4(mrow, em[u]), g4[u]));
        r = fmaf(av, gae, r);
        if (ok) head_put(ga4, idx, gae, lane);
      }
    }
    if (lane < cnt) st4(ga + (int64_t)(e0 + lane) * 4, ga4);
  }
  if (L.act) st4(dKMQ + (int64_t)s * ldk + DP + L.off, dM);
  if (L.j == 0) rs[(int64_t)s * 4 + L.g] = r;
}

__global__ __launch_bounds__(256) void k_edge_bwd_src2(const int* __restrict__ rowptr_s, const int* __restrict__ tgt_s,
                                                       const int* __restrict__ cls_s, const float* __restrict__ KMQ, int ldk,
                                                       const float* __restrict__ EkEm, int lde, int HP, float qscale,
                                                       const float* __restrict__ a, float* __restrict__ dKMQ,
                                                       float* __restrict__ ga, const float* __restrict__ rs, int N) {
  const int s = wave_node();
  if (s >= N) return;
  const Lane L = lane_info(HP);
  const int lane = threadIdx.x & 63, DP = 4 * HP;
  const int beg = __builtin_amdgcn_readfirstlane(rowptr_s[s]), end = __builtin_amdgcn_readfirstlane(rowptr_s[s + 1]);
  const float4 r4 = ld4(rs + (int64_t)s * 4);  // the node's 4 head values, same address in every lane
  float4 dQ = zero4();
  for (int e0 = beg; e0 < end; e0 += 64) {
    const int cnt = min(64, end - e0);
    const int tv = lane < cnt ? tgt_s[e0 + lane] : 0;
    const int cv = lane < cnt ? cls_s[e0 + lane] : 0;
    // lane i: gs of edge e0 + i from its a and ga (coalesced), written back in place of ga (coalesced)
    float4 gs4 = zero4();
    if (lane < cnt) {
      const float4 a4 = ld4(a + (int64_t)(e0 + lane) * 4), g4 = ld4(ga + (int64_t)(e0 + lane) * 4);
      gs4 = make_float4(qscale * a4.x * (g4.x - r4.x), qscale * a4.y * (g4.y - r4.y), qscale * a4.z * (g4.z - r4.z),
                        qscale * a4.w * (g4.w - r4.w));
      st4(ga + (int64_t)(e0 + lane) * 4, gs4);
    }
    for (int i = 0; i < cnt; i += EDGE_UNROLL) {
      float4 k[EDGE_UNROLL], ek[EDGE_UNROLL];
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        const int idx = min(i + u, cnt - 1);
        const int t = __builtin_amdgcn_readlane(tv, idx), c = __builtin_amdgcn_readlane(cv, idx);
        k[u] = L.act ? ld4(KMQ + (int64_t)t * ldk + L.off) : zero4();
        ek[u] = L.act ? ld4(EkEm + (int64_t)c * lde + L.off) : zero4();
      }
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        const float gs = i + u < cnt ? head_get(gs4, min(i + u, cnt - 1), L.g) : 0.f;
        dQ = fma4(gs, add4(k[u], ek[u]), dQ);
      }
    }
  }
  if (L.act) st4(dKMQ + (int64_t)s * ldk + 2 * DP + L.off, dQ);
}

__global__ __launch_bounds__(256) void k_edge_bwd_tgt(const int* __restrict__ rowptr_t, const int* __restrict__ src_t,
                                                      const int* __restrict__ pos_t, const float* __restrict__ KMQ, int ldk,
                                                      int HP, const float* __restrict__ gsb, float* __restrict__ dKMQ, int N) {
  const int t = wave_node();
  if (t >= N) return;
  const Lane L = lane_info(HP);
  const int lane = threadIdx.x & 63, DP = 4 * HP;
  const int beg = __builtin_amdgcn_readfirstlane(rowptr_t[t]), end = __builtin_amdgcn_readfirstlane(rowptr_t[t + 1]);
  float4 dK = zero4();
  for (int e0 = beg; e0 < end; e0 += 64) {
    const int cnt = min(64, end - e0);
    const int sv = lane < cnt ? src_t[e0 + lane] : 0;
    const int pv = lane < cnt ? pos_t[e0 + lane] : 0;
    const float4 gs4 = lane < cnt ? ld4(gsb + (int64_t)pv * 4) : zero4();  // lane i: gs of edge e0 + i (one gather per chunk)
    for (int i = 0; i < cnt; i += EDGE_UNROLL) {
      float4 qv[EDGE_UNROLL];
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        const int idx = min(i + u, cnt - 1);
        const int s = __builtin_amdgcn_readlane(sv, idx);
        qv[u] = L.act ? ld4(KMQ + (int64_t)s * ldk + 2 * DP + L.off) : zero4();
      }
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        const float gs = i + u < cnt ? head_get(gs4, min(i + u, cnt - 1), L.g) : 0.f;
        dK = fma4(gs, qv[u], dK);
      }
    }
  }
  if (L.act) st4(dKMQ + (int64_t)t * ldk + L.off, dK);
}

// one wave per class chunk (<= QAGNN_CLS_CHUNK edges of one class)
__global__ __launch_bounds__(256) void k_edge_bwd_cls(const int* __restrict__ n_chunks, const int* __restrict__ chunk_beg,
                                                      const int* __restrict__ chunk_len, const int* __restrict__ src_c,
                                                      const int* __restrict__ tgt_c, const int* __restrict__ pos_c,
                                                      const float* __restrict__ KMQ, int ldk, int HP,
                                                      const float* __restrict__ alpha, const float* __restrict__ gsb,
                                                      const float* __restrict__ G, int ldg, float* __restrict__ cls_part) {
  const int k = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (k >= *n_chunks) return;
  const Lane L = lane_info(HP);
  const int lane = threadIdx.x & 63, DP = 4 * HP;
  const int beg = __builtin_amdgcn_readfirstlane(chunk_beg[k]), cnt = __builtin_amdgcn_readfirstlane(chunk_len[k]);
  const int sv = lane < cnt ? src_c[beg + lane] : 0;
  const int tv = lane < cnt ? tgt_c[beg + lane] : 0;
  const int pv = lane < cnt ? pos_c[beg + lane] : 0;
  // lane i: gs and alpha of the chunk's edge i (one gather each per chunk)
  const float4 gs4 = lane < cnt ? ld4(gsb + (int64_t)pv * 4) : zero4();
  const float4 al4 = lane < cnt ? ld4(alpha + (int64_t)pv * 4) : zero4();
  float4 dEk = zero4(), dEm = zero4();
  for (int i = 0; i < cnt; i += EDGE_UNROLL) {
    float4 qv[EDGE_UNROLL], g4[EDGE_UNROLL];
#pragma unroll
    for (int u = 0; u < EDGE_UNROLL; ++u) {
      const int idx = min(i + u, cnt - 1);
      const int s = __builtin_amdgcn_readlane(sv, idx), t = __builtin_amdgcn_readlane(tv, idx);
      qv[u] = L.act ? ld4(KMQ + (int64_t)s * ldk + 2 * DP + L.off) : zero4();
      g4[u] = L.act ? ld4(G + (int64_t)t * ldg + L.off) : zero4();
    }
#pragma unroll
    for (int u = 0; u < EDGE_UNROLL; ++u) {
      const bool ok = i + u < cnt;
      const int idx = min(i + u, cnt - 1);
      const float gs = ok ? head_get(gs4, idx, L.g) : 0.f, al = ok ? head_get(al4, idx, L.g) : 0.f;
      dEk = fma4(gs, qv[u], dEk);
      dEm = fma4(al, g4[u], dEm);
    }
  }
  if (L.act) {
    st4(cls_part + (int64_t)k * 2 * DP + L.off, dEk);
    st4(cls_part + (int64_t)k * 2 * DP + DP + L.off, dEm);
  }
}

// ordered sum of a class's chunk partials; block = one class, columns x partitions, fixed combine order
__global__ __launch_bounds__(1024) void k_cls_reduce(const int* __restrict__ chunkptr, const float* __restrict__ cls_part,
                                                     float* __restrict__ dEkEm, int lde, int DP2) {
  extern __shared__ float4 sm4[];
  const int c = blockIdx.x;
  const int ncol4 = DP2 >> 2;
  const int P = 1024 / ncol4;
  const int col4 = threadIdx.x % ncol4, part = threadIdx.x / ncol4;
  const int kb = chunkptr[c], ke = chunkptr[c + 1];
  float4 acc = zero4();
  if (part < P)
    for (int k = kb + part; k < ke; k += P) acc = add4(acc, ld4(cls_part + (int64_t)k * DP2 + col4 * 4));
  if (part < P) sm4[part * ncol4 + col4] = acc;
  __syncthreads();
  if (threadIdx.x < ncol4) {
    float4 s = sm4[threadIdx.x];
    for (int q = 1; q < P; ++q) s = add4(s, sm4[q * ncol4 + threadIdx.x]);
    st4(dEkEm + (int64_t)c * lde + threadIdx.x * 4, s);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// "Bucket walk" variants of the segment kernels (QAGNN_EDGE_WALK=1; default OFF, measured slower -- see below).
//
// The node-per-wave kernels above pay a dependent chain rowptr -> segment indices -> rows for every node, and a QA
// subgraph's segments are short: 7 edges on average, ONE (the self loop) for the ~40 % PAD rows.  The hypothesis was that
// at 64 000 waves of ~2 four-edge batches each the kernels are bound by that start-up chain and by the imbalance between
// PAD rows and hubs.  MEASURED (profiles/r1_run56_edge_walk_ab.txt, r1_run56_by_shape_edge_walk.txt): they are not -- the
// time of every edge kernel is proportional to the number of 832-byte row gathers it issues (~33 us per gather per edge
// at E' = 460 800, i.e. ~11 TB/s out of L1/L2), whatever the wave shape: aggregate 66.6 us vs 65, target pass 55 vs 48,
// source pass 2 92.7 vs 102 (the one win), and the two kernels that trade a register-held row for a per-edge gather pay
// for it exactly (source pass 1: 3 gathers instead of 2, 160 us vs 107).  Step: 24 290 vs 25 075 QA-subgraphs/s.
// Kept behind the switch (parity-tested both ways) as the balanced baseline for a gather-count-reducing design.
// Here a wave owns a BUCKET of 64 consecutive positions of a CSR order instead: it processes
// the segments that START inside its bucket (running over the bucket end to finish the last one), so every wave has
// 64..127 edges of work, the index arrays are read with one coalesced load per 64 edges and there is no rowptr lookup
// at all -- segment boundaries come from the per-position owner array (src_s / tgt_t) with one ballot.  A segment is
// still accumulated by ONE wave in edge order and flushed when its last edge is reached, so the results are
// bit-identical to the node-per-wave kernels (no atomics, fixed order).  Requires every node to own >= 1 position in
// both orders, which qagnn_graph_prep guarantees (a self loop is appended for all N rows, modeling_qagnn.py:436-438).
// The raw scores need no segment at all: k_edge_scores_e is purely edge-parallel.
// ---------------------------------------------------------------------------------------------------------------
struct WalkChunk {
  int ov;                     // lane: owner (node) of position c0 + lane; -1 past the end
  unsigned long long lastm;   // uniform: bit i set = position c0 + i is the last edge of its segment
  int i_lo, i_hi;             // uniform: this wave owns positions [c0 + i_lo, c0 + i_hi) of the chunk
};
// first = true: the wave owns the segments that START inside this chunk (from the first head on); false if none does.
// first = false: the wave is finishing a segment that ran over the previous chunk's end: it owns the positions up to the next head.
__device__ __forceinline__ bool walk_open(const int* __restrict__ own, int Ep, int c0, bool first, WalkChunk& w) {
  const int lane = threadIdx.x & 63, p = c0 + lane;
  const int o = p < Ep ? own[p] : -1;
  const int op = (p > 0 && p < Ep) ? own[p - 1] : -1;
  const int on = p + 1 < Ep ? own[p + 1] : -1;
  const unsigned long long hm = __ballot(p < Ep && o != op);
  w.ov = o;
  w.lastm = __ballot(p < Ep && o != on);
  w.i_lo = 0;
  w.i_hi = min(64, Ep - c0);
  if (first) {
    if (hm == 0) return false;
    w.i_lo = __builtin_ctzll(hm);
  } else if (hm != 0) {
    w.i_hi = __builtin_ctzll(hm);
  }
  return w.i_lo < w.i_hi;
}
__device__ __forceinline__ int wave_bucket() {
  const int lb = xcd_remap(blockIdx.x, gridDim.x);
  return __builtin_amdgcn_readfirstlane(lb * 4 + (threadIdx.x >> 6));
}
__device__ __forceinline__ bool bit_at(unsigned long long m, int i) { return (m >> i) & 1ull; }

// forward 1/3, edge-parallel: wave b scores positions [64 b, 64 b + 64) of the source order
__global__ __launch_bounds__(256) void k_edge_scores_e(const int* __restrict__ src_s, const int* __restrict__ tgt_s,
                                                       const int* __restrict__ cls_s, const float* __restrict__ KMQ, int ldk,
                                                       const float* __restrict__ EkEm, int lde, int HP, float qscale,
                                                       float* __restrict__ score, int Ep, const int* __restrict__ gate) {
  if (gate && *gate == 0) return;
  const int c0 = wave_bucket() * 64;
  if (c0 >= Ep) return;
  const Lane L = lane_info(HP);
  const int lane = threadIdx.x & 63, DP = 4 * HP;
  const int cnt = min(64, Ep - c0), pc = c0 + min(lane, cnt - 1);
  const int sv = src_s[pc], tv = tgt_s[pc], cv = cls_s[pc];
  for (int i = 0; i < cnt; i += EDGE_UNROLL) {
    float4 q[EDGE_UNROLL], k[EDGE_UNROLL], ek[EDGE_UNROLL];
#pragma unroll
    for (int u = 0; u < EDGE_UNROLL; ++u) {
      const int idx = min(i + u, cnt - 1);
      const int s = __builtin_amdgcn_readlane(sv, idx), t = __builtin_amdgcn_readlane(tv, idx);
      const int c = __builtin_amdgcn_readlane(cv, idx);
      q[u] = L.act ? ld4(KMQ + (int64_t)s * ldk + 2 * DP + L.off) : zero4();
      k[u] = L.act ? ld4(KMQ + (int64_t)t * ldk + L.off) : zero4();
      ek[u] = L.act ? ld4(EkEm + (int64_t)c * lde + L.off) : zero4();
    }
#pragma unroll
    for (int u = 0; u < EDGE_UNROLL; ++u) {
      const float p = row16_sum(dot4(q[u], add4(k[u], ek[u]))) * qscale;
      if (i + u < cnt && L.j == 0) score[(int64_t)(c0 + i + u) * 4 + L.g] = p;
    }
  }
}

// forward 3/3: weighted sum of messages by TARGET, bucket walk over the target order
__global__ __launch_bounds__(256) void k_edge_aggregate_w(const int* __restrict__ tgt_t, const int* __restrict__ src_t,
                                                          const int* __restrict__ cls_t, const int* __restrict__ pos_t,
                                                          const float* __restrict__ KMQ, int ldk, const float* __restrict__ EkEm,
                                                          int lde, int HP, const float* __restrict__ alpha,
                                                          float* __restrict__ aggr, int lda, int Ep, const int* __restrict__ gate) {
  if (gate && *gate == 0) return;
  const int b = wave_bucket();
  if (b * 64 >= Ep) return;
  const Lane L = lane_info(HP);
  const int lane = threadIdx.x & 63, DP = 4 * HP;
  float4 acc = zero4();
  bool first = true;
  for (int c0 = b * 64; c0 < Ep; c0 += 64) {
    const int pc = min(c0 + lane, Ep - 1);
    const int sv = src_t[pc], cv = cls_t[pc], pv = pos_t[pc];
    WalkChunk w;
    if (!walk_open(tgt_t, Ep, c0, first, w)) return;
    first = false;
    for (int i = w.i_lo; i < w.i_hi; i += EDGE_UNROLL) {
      float4 m[EDGE_UNROLL], em[EDGE_UNROLL];
      float wgt[EDGE_UNROLL];
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        const int idx = min(i + u, w.i_hi - 1);
        const int s = __builtin_amdgcn_readlane(sv, idx), c = __builtin_amdgcn_readlane(cv, idx);
        const int p = __builtin_amdgcn_readlane(pv, idx);
        m[u] = L.act ? ld4(KMQ + (int64_t)s * ldk + DP + L.off) : zero4();
        em[u] = L.act ? ld4(EkEm + (int64_t)c * lde + DP + L.off) : zero4();
        wgt[u] = i + u < w.i_hi ? alpha[(int64_t)p * 4 + L.g] : 0.f;
      }
      // every gathered row is consumed on every path (clamped duplicates enter with weight 0): a path that skipped them
      // would leave their loads pending at the loop back edge and cost a conservative vmcnt wait in front of the next batch
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        acc = fma4(wgt[u], add4(m[u], em[u]), acc);
        if (i + u < w.i_hi && bit_at(w.lastm, i + u)) {
          const int t = __builtin_amdgcn_readlane(w.ov, i + u);
          if (L.act) st4(aggr + (int64_t)t * lda + L.off, acc);
          acc = zero4();
        }
      }
    }
    if (bit_at(w.lastm, w.i_hi - 1)) return;
  }
}

// backward, source pass 1 (see k_edge_bwd_src1), bucket walk over the source order.  deg(s) comes from rowptr_s, read per
// lane for the chunk's owners (the loads land under the first row gathers).
__global__ __launch_bounds__(256) void k_edge_bwd_src1_w(const int* __restrict__ rowptr_s, const int* __restrict__ src_s,
                                                         const int* __restrict__ tgt_s, const int* __restrict__ cls_s,
                                                         const float* __restrict__ KMQ, int ldk, const float* __restrict__ EkEm,
                                                         int lde, int HP, const float* __restrict__ a,
                                                         const float* __restrict__ alpha, const float* __restrict__ G, int ldg,
                                                         float* __restrict__ dKMQ, float* __restrict__ ga, float* __restrict__ rs,
                                                         int Ep) {
  const int b = wave_bucket();
  if (b * 64 >= Ep) return;
  const Lane L = lane_info(HP);
  const int lane = threadIdx.x & 63, DP = 4 * HP;
  float4 dM = zero4();
  float r = 0.f;
  bool first = true;
  for (int c0 = b * 64; c0 < Ep; c0 += 64) {
    const int pc = min(c0 + lane, Ep - 1);
    const int tv = tgt_s[pc], cv = cls_s[pc];
    WalkChunk w;
    if (!walk_open(src_s, Ep, c0, first, w)) return;
    first = false;
    const int oc = max(w.ov, 0);
    const int dv = rowptr_s[oc + 1] - rowptr_s[oc];
    for (int i = w.i_lo; i < w.i_hi; i += SRC1W_UNROLL) {
      float4 g4[SRC1W_UNROLL], em[SRC1W_UNROLL], mr[SRC1W_UNROLL];
      float al[SRC1W_UNROLL], av[SRC1W_UNROLL];
#pragma unroll
      for (int u = 0; u < SRC1W_UNROLL; ++u) {
        const int idx = min(i + u, w.i_hi - 1);
        const int t = __builtin_amdgcn_readlane(tv, idx), c = __builtin_amdgcn_readlane(cv, idx);
        const int s = __builtin_amdgcn_readlane(w.ov, idx);
        g4[u] = L.act ? ld4(G + (int64_t)t * ldg + L.off) : zero4();
        em[u] = L.act ? ld4(EkEm + (int64_t)c * lde + DP + L.off) : zero4();
        mr[u] = L.act ? ld4(KMQ + (int64_t)s * ldk + DP + L.off) : zero4();
        const bool ok = i + u < w.i_hi;
        al[u] = ok ? alpha[(int64_t)(c0 + idx) * 4 + L.g] : 0.f;
        av[u] = ok ? a[(int64_t)(c0 + idx) * 4 + L.g] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < SRC1W_UNROLL; ++u) {  // clamped duplicates enter with weight 0 (see k_edge_aggregate_w)
        const int idx = min(i + u, w.i_hi - 1);
        const float deg = (float)__builtin_amdgcn_readlane(dv, idx);
        const float gae = deg * row16_sum(dot4(add4(mr[u], em[u]), g4[u]));  // all 64 lanes take part (DPP)
        dM = fma4(al[u], g4[u], dM);
        r = fmaf(av[u], gae, r);
        if (i + u < w.i_hi) {
          if (L.j == 0) ga[(int64_t)(c0 + i + u) * 4 + L.g] = gae;
          if (bit_at(w.lastm, i + u)) {
            const int s = __builtin_amdgcn_readlane(w.ov, i + u);
            if (L.act) st4(dKMQ + (int64_t)s * ldk + DP + L.off, dM);
            if (L.j == 0) rs[(int64_t)s * 4 + L.g] = r;
            dM = zero4();
            r = 0.f;
          }
        }
      }
    }
    if (bit_at(w.lastm, w.i_hi - 1)) return;
  }
}

// backward, source pass 2 (see k_edge_bwd_src2), bucket walk over the source order
__global__ __launch_bounds__(256) void k_edge_bwd_src2_w(const int* __restrict__ src_s, const int* __restrict__ tgt_s,
                                                         const int* __restrict__ cls_s, const float* __restrict__ KMQ, int ldk,
                                                         const float* __restrict__ EkEm, int lde, int HP, float qscale,
                                                         const float* __restrict__ a, float* __restrict__ dKMQ,
                                                         float* __restrict__ ga, const float* __restrict__ rs, int Ep) {
  const int b = wave_bucket();
  if (b * 64 >= Ep) return;
  const Lane L = lane_info(HP);
  const int lane = threadIdx.x & 63, DP = 4 * HP;
  float4 dQ = zero4();
  bool first = true;
  for (int c0 = b * 64; c0 < Ep; c0 += 64) {
    const int pc = min(c0 + lane, Ep - 1);
    const int tv = tgt_s[pc], cv = cls_s[pc];
    WalkChunk w;
    if (!walk_open(src_s, Ep, c0, first, w)) return;
    first = false;
    for (int i = w.i_lo; i < w.i_hi; i += EDGE_UNROLL) {
      float4 k[EDGE_UNROLL], ek[EDGE_UNROLL];
      float gs[EDGE_UNROLL];
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        const int idx = min(i + u, w.i_hi - 1);
        const int t = __builtin_amdgcn_readlane(tv, idx), c = __builtin_amdgcn_readlane(cv, idx);
        const int s = __builtin_amdgcn_readlane(w.ov, idx);
        k[u] = L.act ? ld4(KMQ + (int64_t)t * ldk + L.off) : zero4();
        ek[u] = L.act ? ld4(EkEm + (int64_t)c * lde + L.off) : zero4();
        const int64_t o = (int64_t)(c0 + idx) * 4 + L.g;
        gs[u] = i + u < w.i_hi ? qscale * a[o] * (ga[o] - rs[(int64_t)s * 4 + L.g]) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {  // clamped duplicates enter with weight 0 (see k_edge_aggregate_w)
        dQ = fma4(gs[u], add4(k[u], ek[u]), dQ);
        if (i + u < w.i_hi) {
          if (L.j == 0) ga[(int64_t)(c0 + i + u) * 4 + L.g] = gs[u];  // every lane of the group already read it
          if (bit_at(w.lastm, i + u)) {
            const int s = __builtin_amdgcn_readlane(w.ov, i + u);
            if (L.act) st4(dKMQ + (int64_t)s * ldk + 2 * DP + L.off, dQ);
            dQ = zero4();
          }
        }
      }
    }
    if (bit_at(w.lastm, w.i_hi - 1)) return;
  }
}

// backward, target pass (see k_edge_bwd_tgt), bucket walk over the target order
__global__ __launch_bounds__(256) void k_edge_bwd_tgt_w(const int* __restrict__ tgt_t, const int* __restrict__ src_t,
                                                        const int* __restrict__ pos_t, const float* __restrict__ KMQ, int ldk,
                                                        int HP, const float* __restrict__ gsb, float* __restrict__ dKMQ, int Ep) {
  const int b = wave_bucket();
  if (b * 64 >= Ep) return;
  const Lane L = lane_info(HP);
  const int lane = threadIdx.x & 63, DP = 4 * HP;
  float4 dK = zero4();
  bool first = true;
  for (int c0 = b * 64; c0 < Ep; c0 += 64) {
    const int pc = min(c0 + lane, Ep - 1);
    const int sv = src_t[pc], pv = pos_t[pc];
    WalkChunk w;
    if (!walk_open(tgt_t, Ep, c0, first, w)) return;
    first = false;
    for (int i = w.i_lo; i < w.i_hi; i += EDGE_UNROLL) {
      float4 qv[EDGE_UNROLL];
      float gs[EDGE_UNROLL];
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        const int idx = min(i + u, w.i_hi - 1);
        const int s = __builtin_amdgcn_readlane(sv, idx), p = __builtin_amdgcn_readlane(pv, idx);
        qv[u] = L.act ? ld4(KMQ + (int64_t)s * ldk + 2 * DP + L.off) : zero4();
        gs[u] = i + u < w.i_hi ? gsb[(int64_t)p * 4 + L.g] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {  // clamped duplicates enter with weight 0 (see k_edge_aggregate_w)
        dK = fma4(gs[u], qv[u], dK);
        if (i + u < w.i_hi && bit_at(w.lastm, i + u)) {
          const int t = __builtin_amdgcn_readlane(w.ov, i + u);
          if (L.act) st4(dKMQ + (int64_t)t * ldk + L.off, dK);
          dK = zero4();
        }
      }
    }
    if (bit_at(w.lastm, w.i_hi - 1)) return;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// forward, LDS-resident: one workgroup per (subgraph, head).  A QA subgraph is n consecutive node rows and its edges never
// leave it (LM_QAGNN.batch_graph), so the head's K, M and Q rows of the subgraph -- 3 x n x HP floats, 125 KB at n = 200,
// d = 200 -- fit the 160 KB LDS of a CU.  All per-edge gathers then hit LDS; HBM sees each K|M|Q row once.
// 16 lanes own one edge (13 of them carry the head's 52 floats as float4), so a wave works on 4 edges at a time.
// The softmax is three sweeps over the segment (max, sum, normalise), each recomputing the score from LDS: no per-edge state.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float grp4_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float grp4_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

__global__ __launch_bounds__(512) void k_edge_fwd_blocked(const int* __restrict__ cross_block, const int* __restrict__ rowptr_s,
                                                          const int* __restrict__ tgt_s, const int* __restrict__ cls_s,
                                                          const int* __restrict__ rowptr_t, const int* __restrict__ src_t,
                                                          const int* __restrict__ cls_t, const int* __restrict__ pos_t,
                                                          const float* __restrict__ KMQ, int ldk, const float* __restrict__ EkEm,
                                                          int lde, int HP, float qscale, int n, int alpha_cap,
                                                          float* __restrict__ a, float* __restrict__ alpha,
                                                          float* __restrict__ aggr, int lda) {
  if (*cross_block != 0) return;  // not block-structured: the generic kernels handle this graph
  extern __shared__ __attribute__((aligned(16))) float sm_blk[];
  const int tile = xcd_remap(blockIdx.x, gridDim.x);  // the 4 heads of a subgraph stay on one XCD (shared indices, tables)
  const int gph = tile >> 2, h = tile & 3, node0 = gph * n;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, grp = lane >> 4, j = lane & 15;
  const int DP = 4 * HP, f4 = HP >> 2;
  const bool act = j < f4;
  float* const Ks = sm_blk;
  float* const Ms = Ks + n * HP;
  float* const Qs = Ms + n * HP;
  float* const alphaL = Qs + n * HP;
  for (int idx = tid; idx < n * f4; idx += 512) {  // stage the head's K, M, Q slices: 208-byte runs, each read once
    const int row = idx / f4, c4 = idx - row * f4;
    const float* src = KMQ + (int64_t)(node0 + row) * ldk + h * HP + c4 * 4;
    st4(Ks + row * HP + c4 * 4, ld4(src));
    st4(Ms + row * HP + c4 * 4, ld4(src + DP));
    st4(Qs + row * HP + c4 * 4, ld4(src + 2 * DP));
  }
  const int ebase = rowptr_s[node0], Eg = rowptr_s[node0 + n] - ebase;
  const bool in_lds = Eg <= alpha_cap;
  __syncthreads();

  // phase 1: scores + softmax over the out-edges of every source node of the subgraph
  for (int sl = w; sl < n; sl += 8) {
    const int s = node0 + sl;
    const int beg = __builtin_amdgcn_readfirstlane(rowptr_s[s]), end = __builtin_amdgcn_readfirstlane(rowptr_s[s + 1]);
    const float4 q4 = act ? ld4(Qs + sl * HP + j * 4) : zero4();
    const float deg = (float)(end - beg);
    auto score_at = [&](int e) {  // all 64 lanes call this (shuffles inside); e is clamped by the caller
      const int t = tgt_s[e] - node0, c = cls_s[e];
      const float4 k4 = act ? ld4(Ks + t * HP + j * 4) : zero4();
      const float4 ek = act ? ld4(EkEm + (int64_t)c * lde + h * HP + j * 4) : zero4();
      return row16_sum(dot4(q4, add4(k4, ek))) * qscale;
    };
    float m = -INFINITY;
    for (int e0 = beg; e0 < end; e0 += 4) {
      const int e = e0 + grp;
      const float p = score_at(min(e, end - 1));
      if (e < end) m = fmaxf(m, p);
    }
    m = grp4_max(m);
    float sum = 0.f;
    for (int e0 = beg; e0 < end; e0 += 4) {
      const int e = e0 + grp;
      const float p = score_at(min(e, end - 1));
      if (e < end) sum += expf(p - m);
    }
    sum = grp4_sum(sum);
    for (int e0 = beg; e0 < end; e0 += 4) {
      const int e = e0 + grp;
      const float p = score_at(min(e, end - 1));
      if (e < end && j == 0) {
        const float av = expf(p - m) / (sum + 1e-16f), al = av * deg;
        a[(int64_t)e * 4 + h] = av;
        alpha[(int64_t)e * 4 + h] = al;
        if (in_lds) alphaL[e - ebase] = al;
      }
    }
  }
  if (!in_lds) __threadfence();  // oversized subgraph: phase 2 reads alpha back from global memory
  __syncthreads();

  // phase 2: weighted sum of messages into every target node of the subgraph
  for (int tl = w; tl < n; tl += 8) {
    const int t = node0 + tl;
    const int beg = __builtin_amdgcn_readfirstlane(rowptr_t[t]), end = __builtin_amdgcn_readfirstlane(rowptr_t[t + 1]);
    float4 acc = zero4();
    for (int e0 = beg; e0 < end; e0 += 4) {
      const int e = e0 + grp, ec = min(e, end - 1);
      const int sl = src_t[ec] - node0, c = cls_t[ec], p = pos_t[ec];
      const float al = e < end ? (in_lds ? alphaL[p - ebase] : alpha[(int64_t)p * 4 + h]) : 0.f;
      const float4 m4 = act ? ld4(Ms + sl * HP + j * 4) : zero4();
      const float4 em = act ? ld4(EkEm + (int64_t)c * lde + DP + h * HP + j * 4) : zero4();
      acc = fma4(al, add4(m4, em), acc);
    }
    acc.x = grp4_sum(acc.x);
    acc.y = grp4_sum(acc.y);
    acc.z = grp4_sum(acc.z);
    acc.w = grp4_sum(acc.w);
    if (grp == 0 && act) st4(aggr + (int64_t)t * lda + h * HP + j * 4, acc);
  }
}

// QAGNN_EDGE_WALK=1 selects the bucket-walk kernels (A/B switch; measured 3 % slower on the step, default off)
static bool edge_walk_enabled() {
  static const int v = getenv("QAGNN_EDGE_WALK") ? atoi(getenv("QAGNN_EDGE_WALK")) : 0;
  return v != 0;
}

static int check_common(const qagnn_graph* g, const float* KMQ, int ldk, const float* EkEm, int lde, int HP, const char* who) {
  QAGNN_REQUIRE(g && KMQ && EkEm, QAGNN_EINVAL, "%s: null pointer", who);
  QAGNN_REQUIRE(HP > 0 && HP % 4 == 0 && HP <= 64, QAGNN_EUNSUPPORTED, "%s: head pitch HP=%d must be a multiple of 4, <= 64", who, HP);
  QAGNN_REQUIRE(ldk >= 12 * HP && ldk % 4 == 0 && lde >= 8 * HP && lde % 4 == 0, QAGNN_EINVAL, "%s: ldk=%d lde=%d too small for HP=%d",
                who, ldk, lde, HP);
  QAGNN_REQUIRE(aligned16(KMQ) && aligned16(EkEm), QAGNN_EINVAL, "%s: operands must be 16-byte aligned", who);
  return QAGNN_OK;
}

}  // namespace qagnn

using namespace qagnn;

static int edge_attn_fwd_generic(const qagnn_graph* g, const float* KMQ, int32_t ldk, const float* EkEm, int32_t lde, int32_t HP,
                                 float qscale, float* score, float* a, float* alpha, float* aggr, int32_t lda, const int* gate,
                                 hipStream_t stream) {
  int rc = check_common(g, KMQ, ldk, EkEm, lde, HP, "edge_attn_fwd");
  if (rc != QAGNN_OK) return rc;
  QAGNN_REQUIRE(score && a && alpha && aggr && lda >= 4 * HP && lda % 4 == 0 && aligned16(aggr), QAGNN_EINVAL,
                "edge_attn_fwd: bad output arguments");
  const int nb = cdiv(g->N, 4);
  if (edge_walk_enabled()) {
    const int nbw = cdiv(cdiv(g->Ep, 64), 4);  // one wave per 64 positions of the source / target order
    k_edge_scores_e<<<nbw, 256, 0, stream>>>(g->src_s, g->tgt_s, g->cls_s, KMQ, ldk, EkEm, lde, HP, qscale, score, g->Ep, gate);
    QAGNN_LAUNCH_CHECK("k_edge_scores_e");
    k_edge_softmax<<<nb, 256, 0, stream>>>(g->rowptr_s, score, a, alpha, g->N, gate);
    QAGNN_LAUNCH_CHECK("k_edge_softmax");
    k_edge_aggregate_w<<<nbw, 256, 0, stream>>>(g->tgt_t, g->src_t, g->cls_t, g->pos_t, KMQ, ldk, EkEm, lde, HP, alpha, aggr, lda, g->Ep, gate);
    QAGNN_LAUNCH_CHECK("k_edge_aggregate_w");
    return QAGNN_OK;
  }
  k_edge_scores<<<nb, 256, 0, stream>>>(g->rowptr_s, g->tgt_s, g->cls_s, KMQ, ldk, EkEm, lde, HP, qscale, score, g->N, gate);
  QAGNN_LAUNCH_CHECK("k_edge_scores");
  k_edge_softmax<<<nb, 256, 0, stream>>>(g->rowptr_s, score, a, alpha, g->N, gate);
  QAGNN_LAUNCH_CHECK("k_edge_softmax");
  k_edge_aggregate<<<nb, 256, 0, stream>>>(g->rowptr_t, g->src_t, g->cls_t, g->pos_t, KMQ, ldk, EkEm, lde, HP, alpha, aggr, lda, g->N, gate);
  QAGNN_LAUNCH_CHECK("k_edge_aggregate");
  return QAGNN_OK;
}

extern "C" int qagnn_edge_attn_fwd_f32(const qagnn_graph* g, const float* KMQ, int32_t ldk, const float* EkEm, int32_t lde,
                                       int32_t HP, float qscale, float* score, float* a, float* alpha, float* aggr, int32_t lda,
                                       qagnn_stream_t stream_) {
  return edge_attn_fwd_generic(g, KMQ, ldk, EkEm, lde, HP, qscale, score, a, alpha, aggr, lda, nullptr, (hipStream_t)stream_);
}

extern "C" int qagnn_edge_attn_fwd_blocked_f32(const qagnn_graph* g, const float* KMQ, int32_t ldk, const float* EkEm, int32_t lde,
                                               int32_t HP, float qscale, float* score, float* a, float* alpha, float* aggr,
                                               int32_t lda, qagnn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_common(g, KMQ, ldk, EkEm, lde, HP, "edge_attn_fwd_blocked");
  if (rc != QAGNN_OK) return rc;
  QAGNN_REQUIRE(g->block_n > 0 && g->N % g->block_n == 0, QAGNN_EINVAL, "edge_attn_fwd_blocked: graph was not prepared with a block size");
  const int n = g->block_n;
  const size_t lds_max = 160 * 1024, slabs = (size_t)3 * n * HP * sizeof(float);
  QAGNN_REQUIRE(slabs + 4096 <= lds_max, QAGNN_EUNSUPPORTED, "edge_attn_fwd_blocked: 3 x %d x %d floats do not fit the LDS", n, HP);
  const int alpha_cap = (int)((lds_max - slabs) / sizeof(float)) & ~3;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_edge_fwd_blocked, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
    if (e != hipSuccess) { set_error("edge_attn_fwd_blocked: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e)); return QAGNN_EHIP; }
    attr_set = true;
  }
  const int* cross = g->err + 1;
  k_edge_fwd_blocked<<<(g->N / n) * 4, 512, slabs + (size_t)alpha_cap * sizeof(float), stream>>>(
      cross, g->rowptr_s, g->tgt_s, g->cls_s, g->rowptr_t, g->src_t, g->cls_t, g->pos_t, KMQ, ldk, EkEm, lde, HP, qscale, n,
      alpha_cap, a, alpha, aggr, lda);
  QAGNN_LAUNCH_CHECK("k_edge_fwd_blocked");
  // graphs that are not block-structured fall through to the generic kernels (gated on the same device flag)
  return edge_attn_fwd_generic(g, KMQ, ldk, EkEm, lde, HP, qscale, score, a, alpha, aggr, lda, cross, stream);
}

extern "C" int qagnn_edge_attn_bwd_f32(const qagnn_graph* g, const float* KMQ, int32_t ldk, const float* EkEm, int32_t lde,
                                       int32_t HP, float qscale, const float* a, const float* alpha, const float* G, int32_t ldg,
                                       float* dKMQ, float* dEkEm, float* ga, float* rs, float* cls_part, qagnn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_common(g, KMQ, ldk, EkEm, lde, HP, "edge_attn_bwd");
  if (rc != QAGNN_OK) return rc;
  QAGNN_REQUIRE(a && alpha && G && dKMQ && dEkEm && ga && rs && cls_part, QAGNN_EINVAL, "edge_attn_bwd: null pointer");
  QAGNN_REQUIRE(ldg >= 4 * HP && ldg % 4 == 0 && aligned16(G) && aligned16(dKMQ) && aligned16(dEkEm) && aligned16(cls_part),
                QAGNN_EINVAL, "edge_attn_bwd: bad pitch / alignment");
  const int DP2 = 8 * HP;
  QAGNN_REQUIRE(DP2 / 4 <= 1024, QAGNN_EUNSUPPORTED, "edge_attn_bwd: HP too large");
  const int nb = cdiv(g->N, 4);
  if (edge_walk_enabled()) {
    const int nbw = cdiv(cdiv(g->Ep, 64), 4);
    k_edge_bwd_src1_w<<<nbw, 256, 0, stream>>>(g->rowptr_s, g->src_s, g->tgt_s, g->cls_s, KMQ, ldk, EkEm, lde, HP, a, alpha, G, ldg, dKMQ, ga, rs, g->Ep);
    QAGNN_LAUNCH_CHECK("k_edge_bwd_src1_w");
    k_edge_bwd_src2_w<<<nbw, 256, 0, stream>>>(g->src_s, g->tgt_s, g->cls_s, KMQ, ldk, EkEm, lde, HP, qscale, a, dKMQ, ga, rs, g->Ep);
    QAGNN_LAUNCH_CHECK("k_edge_bwd_src2_w");
    k_edge_bwd_tgt_w<<<nbw, 256, 0, stream>>>(g->tgt_t, g->src_t, g->pos_t, KMQ, ldk, HP, ga, dKMQ, g->Ep);
    QAGNN_LAUNCH_CHECK("k_edge_bwd_tgt_w");
  } else {
    k_edge_bwd_src1<<<nb, 256, 0, stream>>>(g->rowptr_s, g->tgt_s, g->cls_s, KMQ, ldk, EkEm, lde, HP, a, alpha, G, ldg, dKMQ, ga, rs, g->N);
    QAGNN_LAUNCH_CHECK("k_edge_bwd_src1");
    k_edge_bwd_src2<<<nb, 256, 0, stream>>>(g->rowptr_s, g->tgt_s, g->cls_s, KMQ, ldk, EkEm, lde, HP, qscale, a, dKMQ, ga, rs, g->N);
    QAGNN_LAUNCH_CHECK("k_edge_bwd_src2");
    k_edge_bwd_tgt<<<nb, 256, 0, stream>>>(g->rowptr_t, g->src_t, g->pos_t, KMQ, ldk, HP, ga, dKMQ, g->N);
    QAGNN_LAUNCH_CHECK("k_edge_bwd_tgt");
  }
  k_edge_bwd_cls<<<cdiv(g->max_chunks, 4), 256, 0, stream>>>(g->n_chunks, g->chunk_beg, g->chunk_len, g->src_c, g->tgt_c, g->pos_c,
                                                             KMQ, ldk, HP, alpha, ga, G, ldg, cls_part);
  QAGNN_LAUNCH_CHECK("k_edge_bwd_cls");
  const int P = 1024 / (DP2 / 4);
  k_cls_reduce<<<g->C, 1024, (size_t)P * (DP2 / 4) * sizeof(float4), stream>>>(g->chunkptr, cls_part, dEkEm, lde, DP2);
  QAGNN_LAUNCH_CHECK("k_cls_reduce");
  return QAGNN_OK;
}
