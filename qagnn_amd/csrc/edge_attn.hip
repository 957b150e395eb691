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
#include "common.h"

namespace qagnn {

#define EDGE_UNROLL 4

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

// ---------------------------------------------------------------------------------------------------------------
// forward 1/3: raw scores, one wave per SOURCE node (Q row in registers; gathers K[tgt] and Ek[cls])
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_edge_scores(const int* __restrict__ rowptr_s, const int* __restrict__ tgt_s,
                                                     const int* __restrict__ cls_s, const float* __restrict__ KMQ, int ldk,
                                                     const float* __restrict__ EkEm, int lde, int HP, float qscale,
                                                     float* __restrict__ score, int N) {
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
        if (i + u < cnt && L.j == 0) score[(int64_t)(e0 + i + u) * 4 + L.g] = p;
      }
    }
  }
}

// forward 2/3: softmax over each source segment (PyG softmax: max, exp, sum, / (sum + 1e-16)), then * out-degree
__global__ __launch_bounds__(256) void k_edge_softmax(const int* __restrict__ rowptr_s, const float* __restrict__ score,
                                                      float* __restrict__ a, float* __restrict__ alpha, int N) {
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
                                                        float* __restrict__ aggr, int lda, int N) {
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
    for (int i = 0; i < cnt; i += EDGE_UNROLL) {
      float4 m[EDGE_UNROLL], em[EDGE_UNROLL];
      float wgt[EDGE_UNROLL];
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        const int idx = min(i + u, cnt - 1);
        const int s = __builtin_amdgcn_readlane(sv, idx), c = __builtin_amdgcn_readlane(cv, idx);
        const int p = __builtin_amdgcn_readlane(pv, idx);
        m[u] = L.act ? ld4(KMQ + (int64_t)s * ldk + DP + L.off) : zero4();
        em[u] = L.act ? ld4(EkEm + (int64_t)c * lde + DP + L.off) : zero4();
        wgt[u] = i + u < cnt ? alpha[(int64_t)p * 4 + L.g] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) acc = fma4(wgt[u], add4(m[u], em[u]), acc);
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
    for (int i = 0; i < cnt; i += EDGE_UNROLL) {
      float4 g4[EDGE_UNROLL], em[EDGE_UNROLL];
      float al[EDGE_UNROLL], av[EDGE_UNROLL];
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        const int idx = min(i + u, cnt - 1);
        const int t = __builtin_amdgcn_readlane(tv, idx), c = __builtin_amdgcn_readlane(cv, idx);
        g4[u] = L.act ? ld4(G + (int64_t)t * ldg + L.off) : zero4();
        em[u] = L.act ? ld4(EkEm + (int64_t)c * lde + DP + L.off) : zero4();
        const bool ok = i + u < cnt;
        al[u] = ok ? alpha[(int64_t)(e0 + idx) * 4 + L.g] : 0.f;
        av[u] = ok ? a[(int64_t)(e0 + idx) * 4 + L.g] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        dM = fma4(al[u], g4[u], dM);
        const float gae = deg * row16_sum(dot4(add4(mrow, em[u]), g4[u]));
        r = fmaf(av[u], gae, r);
        if (i + u < cnt && L.j == 0) ga[(int64_t)(e0 + i + u) * 4 + L.g] = gae;
      }
    }
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
  const float r = rs[(int64_t)s * 4 + L.g];
  float4 dQ = zero4();
  for (int e0 = beg; e0 < end; e0 += 64) {
    const int cnt = min(64, end - e0);
    const int tv = lane < cnt ? tgt_s[e0 + lane] : 0;
    const int cv = lane < cnt ? cls_s[e0 + lane] : 0;
    for (int i = 0; i < cnt; i += EDGE_UNROLL) {
      float4 k[EDGE_UNROLL], ek[EDGE_UNROLL];
      float gs[EDGE_UNROLL];
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        const int idx = min(i + u, cnt - 1);
        const int t = __builtin_amdgcn_readlane(tv, idx), c = __builtin_amdgcn_readlane(cv, idx);
        k[u] = L.act ? ld4(KMQ + (int64_t)t * ldk + L.off) : zero4();
        ek[u] = L.act ? ld4(EkEm + (int64_t)c * lde + L.off) : zero4();
        const int64_t o = (int64_t)(e0 + idx) * 4 + L.g;
        gs[u] = i + u < cnt ? qscale * a[o] * (ga[o] - r) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        dQ = fma4(gs[u], add4(k[u], ek[u]), dQ);
        if (i + u < cnt && L.j == 0) ga[(int64_t)(e0 + i + u) * 4 + L.g] = gs[u];  // every lane of the group already read it
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
    for (int i = 0; i < cnt; i += EDGE_UNROLL) {
      float4 qv[EDGE_UNROLL];
      float gs[EDGE_UNROLL];
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        const int idx = min(i + u, cnt - 1);
        const int s = __builtin_amdgcn_readlane(sv, idx), p = __builtin_amdgcn_readlane(pv, idx);
        qv[u] = L.act ? ld4(KMQ + (int64_t)s * ldk + 2 * DP + L.off) : zero4();
        gs[u] = i + u < cnt ? gsb[(int64_t)p * 4 + L.g] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) dK = fma4(gs[u], qv[u], dK);
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
  float4 dEk = zero4(), dEm = zero4();
  for (int i = 0; i < cnt; i += EDGE_UNROLL) {
    float4 qv[EDGE_UNROLL], g4[EDGE_UNROLL];
    float gs[EDGE_UNROLL], al[EDGE_UNROLL];
#pragma unroll
    for (int u = 0; u < EDGE_UNROLL; ++u) {
      const int idx = min(i + u, cnt - 1);
      const int s = __builtin_amdgcn_readlane(sv, idx), t = __builtin_amdgcn_readlane(tv, idx);
      const int p = __builtin_amdgcn_readlane(pv, idx);
      qv[u] = L.act ? ld4(KMQ + (int64_t)s * ldk + 2 * DP + L.off) : zero4();
      g4[u] = L.act ? ld4(G + (int64_t)t * ldg + L.off) : zero4();
      const bool ok = i + u < cnt;
      gs[u] = ok ? gsb[(int64_t)p * 4 + L.g] : 0.f;
      al[u] = ok ? alpha[(int64_t)p * 4 + L.g] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < EDGE_UNROLL; ++u) {
      dEk = fma4(gs[u], qv[u], dEk);
      dEm = fma4(al[u], g4[u], dEm);
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

extern "C" int qagnn_edge_attn_fwd_f32(const qagnn_graph* g, const float* KMQ, int32_t ldk, const float* EkEm, int32_t lde,
                                       int32_t HP, float qscale, float* score, float* a, float* alpha, float* aggr, int32_t lda,
                                       qagnn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_common(g, KMQ, ldk, EkEm, lde, HP, "edge_attn_fwd");
  if (rc != QAGNN_OK) return rc;
  QAGNN_REQUIRE(score && a && alpha && aggr && lda >= 4 * HP && lda % 4 == 0 && aligned16(aggr), QAGNN_EINVAL,
                "edge_attn_fwd: bad output arguments");
  const int nb = cdiv(g->N, 4);
  k_edge_scores<<<nb, 256, 0, stream>>>(g->rowptr_s, g->tgt_s, g->cls_s, KMQ, ldk, EkEm, lde, HP, qscale, score, g->N);
  QAGNN_LAUNCH_CHECK("k_edge_scores");
  k_edge_softmax<<<nb, 256, 0, stream>>>(g->rowptr_s, score, a, alpha, g->N);
  QAGNN_LAUNCH_CHECK("k_edge_softmax");
  k_edge_aggregate<<<nb, 256, 0, stream>>>(g->rowptr_t, g->src_t, g->cls_t, g->pos_t, KMQ, ldk, EkEm, lde, HP, alpha, aggr, lda, g->N);
  QAGNN_LAUNCH_CHECK("k_edge_aggregate");
  return QAGNN_OK;
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
  k_edge_bwd_src1<<<nb, 256, 0, stream>>>(g->rowptr_s, g->tgt_s, g->cls_s, KMQ, ldk, EkEm, lde, HP, a, alpha, G, ldg, dKMQ, ga, rs, g->N);
  QAGNN_LAUNCH_CHECK("k_edge_bwd_src1");
  k_edge_bwd_src2<<<nb, 256, 0, stream>>>(g->rowptr_s, g->tgt_s, g->cls_s, KMQ, ldk, EkEm, lde, HP, qscale, a, dKMQ, ga, rs, g->N);
  QAGNN_LAUNCH_CHECK("k_edge_bwd_src2");
  k_edge_bwd_tgt<<<nb, 256, 0, stream>>>(g->rowptr_t, g->src_t, g->pos_t, KMQ, ldk, HP, ga, dKMQ, g->N);
  QAGNN_LAUNCH_CHECK("k_edge_bwd_tgt");
  k_edge_bwd_cls<<<cdiv(g->max_chunks, 4), 256, 0, stream>>>(g->n_chunks, g->chunk_beg, g->chunk_len, g->src_c, g->tgt_c, g->pos_c,
                                                             KMQ, ldk, HP, alpha, ga, G, ldg, cls_part);
  QAGNN_LAUNCH_CHECK("k_edge_bwd_cls");
  const int P = 1024 / (DP2 / 4);
  k_cls_reduce<<<g->C, 1024, (size_t)P * (DP2 / 4) * sizeof(float4), stream>>>(g->chunkptr, cls_part, dEkEm, lde, DP2);
  QAGNN_LAUNCH_CHECK("k_cls_reduce");
  return QAGNN_OK;
}
