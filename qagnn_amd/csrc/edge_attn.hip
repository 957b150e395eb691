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

#ifndef EDGE_UNROLL
#define EDGE_UNROLL 4  // edges whose row gathers are in flight per wave (tools/build_micro.sh builds -DEDGE_UNROLL=6 / 8 variants for A/B)
#endif
#define TGT_UNROLL 4  // measured: 8 in flight win on balanced 64-edge chunks (gather_micro: -13 %) but lose on real segments (~7 edges: the
                      // clamped tail of a batch is wasted gathers) -- target pass 48 us vs 45, all kernels 4 / 6 / 8: profiles/r1_run70_edge_unroll_ab.txt

// What bounds these kernels (measured, profiles/r1_run56_*, r1_run59_gather_micro.txt, r1_run60_*): NOT the bytes.  One wave
// handles one edge at a time, so every per-edge instruction is paid by a whole wave, and a kernel's time tracks the number
// of instructions it issues per edge (~1 us per instruction per edge at E' = 460 800), whatever they move.  Hence:
//   * rows are read with BUFFER loads: address = descriptor base + per-lane byte offset (VGPR, fixed) + per-edge row offset
//     (SGPR, straight from v_readlane) -- no vector address arithmetic, and the lanes past the head width (13 of 16 lanes
//     carry a head's 52 floats) use an out-of-range offset, for which the hardware returns 0: no predication around the load;
//   * row offsets are pre-multiplied per 64-edge chunk (one v_mul per 64 edges);
//   * per-edge head scalars (a, alpha, ga, gs, score: [E', 4] arrays) are read / written with ONE coalesced 16-byte-per-lane
//     access per chunk and handed between "lane i" and "the 16 lanes of head group g" through a per-wave LDS slab
//     (ds_write_b128 once per chunk, one broadcast ds_read_b32 per edge) instead of one global load or store per edge.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr uint32_t OOB_OFF = 0x7FFFFFF0u;  // >= any num_records we accept: the load returns 0 and touches no memory
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);  // raw buffer, gfx9 dword 3
}
__device__ __forceinline__ float4 buf_ld4(rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}

constexpr int SLAB_ROWS = 72;  // 64 chunk entries + the rows an unrolled batch (<= 8 edges) may touch past the chunk (always zero)

struct Lane {
  int lane, w, g, j;
  bool act;
  uint32_t voff;  // byte offset of this lane's float4 inside a head-padded row; OOB_OFF for the idle lanes
  int off;        // the same in floats (stores)
};
__device__ __forceinline__ Lane lane_info(int HP) {
  Lane L;
  L.lane = threadIdx.x & 63;
  L.w = threadIdx.x >> 6;
  L.g = L.lane >> 4;
  L.j = L.lane & 15;
  L.act = L.j * 4 < HP;
  L.off = L.g * HP + L.j * 4;
  L.voff = L.act ? (uint32_t)L.off * 4u : OOB_OFF;
  return L;
}
// node handled by this wave: 4 waves per block, blocks remapped so that an XCD owns a contiguous node range holding an eighth of the
// batch's edges (xcd_base: qagnn_graph.err + 4, written by the graph preparation); INT_MAX = a block past its XCD's run
__device__ __forceinline__ int wave_node(const int* __restrict__ xcd_base) {
  const int lb = xcd_remap_balanced(blockIdx.x, xcd_base);
  return __builtin_amdgcn_readfirstlane(lb < 0 ? 0x7FFFFFFF : lb * 4 + (threadIdx.x >> 6));
}
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ uint32_t rl(uint32_t v, int i) { return (uint32_t)__builtin_amdgcn_readlane((int)v, i); }
// rows 64.. of a slab are read (as weight 0) by the clamped tail of the last unrolled batch: zero them once per wave
__device__ __forceinline__ void slab_init(float4* slab, int lane) {
  if (lane < SLAB_ROWS - 64) slab[64 + lane] = zero4();
}

// ---------------------------------------------------------------------------------------------------------------
// forward 1/2: scores + softmax over the out-edges, one wave per SOURCE node (Q row in registers; gathers K[tgt], Ek[cls]).
// PyG softmax (max, exp, sum, / (sum + 1e-16)) then * out-degree.  A segment of <= 64 edges (all but the hubs) never leaves
// the wave: its scores sit in the LDS slab, lane i takes the 4 head scores of edge i, the max / sum run over the lanes and
// a, alpha go out as one coalesced float4 per lane.  Longer segments write their raw scores chunk by chunk and are
// normalised by three sweeps over them afterwards (same wave; the sweeps read what other lanes of the wave stored).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 wave_max4(float4 v) {
  return make_float4(wave_max(v.x), wave_max(v.y), wave_max(v.z), wave_max(v.w));
}
__device__ __forceinline__ float4 wave_sum4(float4 v) {
  return make_float4(wave_sum(v.x), wave_sum(v.y), wave_sum(v.z), wave_sum(v.w));
}

// hub segment (> 64 out-edges): raw scores chunk by chunk through the LDS slab into `score`, then max / sum / normalise in three
// sweeps over what this wave stored
__device__ __forceinline__ void scores_hub(float4 (&slab)[SLAB_ROWS], const Lane& L, const int* __restrict__ tgt_s,
                                           const int* __restrict__ cls_s, rsrc_t rK, rsrc_t rE, uint32_t pk, uint32_t pe, float4 q,
                                           float qscale, int beg, int end, float* __restrict__ score, float* __restrict__ a,
                                           float* __restrict__ alpha) {
  const float deg = (float)(end - beg);
  float* const sl = reinterpret_cast<float*>(slab) + L.g;
  for (int e0 = beg; e0 < end; e0 += 64) {
    const int cnt = min(64, end - e0), lc = e0 + min(L.lane, cnt - 1);
    const uint32_t tv = (uint32_t)tgt_s[lc] * pk, cv = (uint32_t)cls_s[lc] * pe;
    for (int i = 0; i < cnt; i += EDGE_UNROLL) {
      float4 k[EDGE_UNROLL], ek[EDGE_UNROLL];
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        const int idx = min(i + u, cnt - 1);
        k[u] = buf_ld4(rK, L.voff, rl(tv, idx));
        ek[u] = buf_ld4(rE, L.voff, rl(cv, idx));
      }
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        const float p = row16_sum(dot4(q, add4(k[u], ek[u]))) * qscale;
        if (L.j == 0) sl[(i + u) * 4] = p;  // rows past cnt take the clamped duplicates: never used
      }
    }
    const float4 sc = slab[L.lane];  // lane i: the 4 head scores of edge e0 + i
    if (L.lane < cnt) st4(score + (int64_t)(e0 + L.lane) * 4, sc);
  }
  // make the raw scores this wave stored visible to all of its lanes
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  float m = -INFINITY;
  for (int e = beg + L.j; e < end; e += 16) m = fmaxf(m, score[(int64_t)e * 4 + L.g]);
  m = row16_max(m);
  float sum = 0.f;
  for (int e = beg + L.j; e < end; e += 16) sum += expf(score[(int64_t)e * 4 + L.g] - m);
  sum = row16_sum(sum);
  for (int e = beg + L.j; e < end; e += 16) {
    const float av = expf(score[(int64_t)e * 4 + L.g] - m) / (sum + 1e-16f);
    a[(int64_t)e * 4 + L.g] = av;
    alpha[(int64_t)e * 4 + L.g] = av * deg;
  }
}

// One wave per SOURCE node.  Three shapes of segment (rocprof of the first form: 64 000 waves x ~300 instructions of per-node work
// -- a 4-component wave-wide softmax with bpermute round trips -- against ~20 per edge; 40 % of the node rows of a CSQA batch are
// PAD rows whose only edge is their self loop):
//   deg == 1   the self loop alone (every PAD row, every isolated node): softmax of one score is exp(0) / (exp(0) + 1e-16) = 1 in
//              fp32 whatever the score, alpha = deg * a = 1 -- two 16-byte stores, no row is read;
//   deg <= 64  scores stay in REGISTERS in (head, edge) layout: the row-of-16 reduction leaves head g's score of edge i in all 16
//              lanes of DPP row g, lane (g, i & 15) keeps it (one v_cndmask); the softmax is then ONE component per lane: an in-row
//              max and an in-row sum (4 DPP steps each, no LDS, no bpermute), one exp and one divide per (edge, head);
//   deg > 64   hubs: scores_hub above.
__global__ __launch_bounds__(256) void k_edge_scores(const int* __restrict__ rowptr_s, const int* __restrict__ tgt_s,
                                                     const int* __restrict__ cls_s, const float* __restrict__ KMQ, int ldk,
                                                     const float* __restrict__ EkEm, int lde, int HP, float qscale,
                                                     float* __restrict__ score, float* __restrict__ a, float* __restrict__ alpha,
                                                     int N, int C, const int* __restrict__ xcd_base) {
  __shared__ float4 slab[4][SLAB_ROWS];
  const int s = wave_node(xcd_base);
  if (s >= N) return;
  const int beg = __builtin_amdgcn_readfirstlane(rowptr_s[s]), end = __builtin_amdgcn_readfirstlane(rowptr_s[s + 1]);
  const int cnt = end - beg;
  const int lane = threadIdx.x & 63;
  if (cnt == 1) {
    if (lane < 2) st4((lane == 0 ? a : alpha) + (int64_t)beg * 4, make_float4(1.f, 1.f, 1.f, 1.f));
    return;
  }
  const Lane L = lane_info(HP);
  const int DP = 4 * HP;
  const uint32_t pk = (uint32_t)ldk * 4u, pe = (uint32_t)lde * 4u;
  const rsrc_t rK = make_rsrc(KMQ, (uint32_t)N * pk), rE = make_rsrc(EkEm, (uint32_t)C * pe);
  const float4 q = buf_ld4(rK, L.voff + 2u * DP * 4u, (uint32_t)s * pk);
  if (cnt > 64) {
    scores_hub(slab[L.w], L, tgt_s, cls_s, rK, rE, pk, pe, q, qscale, beg, end, score, a, alpha);
    return;
  }
  const float deg = (float)cnt;
  const int lc = beg + min(L.lane, cnt - 1);
  const uint32_t tv = (uint32_t)tgt_s[lc] * pk, cv = (uint32_t)cls_s[lc] * pe;
  float sc[4];
#pragma unroll
  for (int grp = 0; grp < 4; ++grp) {
    sc[grp] = -INFINITY;
    if (grp * 16 < cnt) {  // wave-uniform
      const int gcnt = min(16, cnt - grp * 16);
      for (int i = 0; i < gcnt; i += EDGE_UNROLL) {
        float4 k[EDGE_UNROLL], ek[EDGE_UNROLL];
#pragma unroll
        for (int u = 0; u < EDGE_UNROLL; ++u) {
          const int idx = min(grp * 16 + i + u, cnt - 1);
          k[u] = buf_ld4(rK, L.voff, rl(tv, idx));
          ek[u] = buf_ld4(rE, L.voff, rl(cv, idx));
        }
#pragma unroll
        for (int u = 0; u < EDGE_UNROLL; ++u) {
          const float p = row16_sum(dot4(q, add4(k[u], ek[u]))) * qscale;
          sc[grp] = (L.j == i + u) ? p : sc[grp];  // slots past the segment take clamped duplicates: masked below
        }
      }
    }
  }
  // PyG softmax over the segment (max, exp, sum, / (sum + 1e-16)), then * out-degree; lane (g, j) owns edges j, j + 16, ... of head g.
  // Groups past the segment are skipped by wave-uniform branches (most segments fit one or two groups).
  float m = sc[0];  // group 0 is never empty; its slots past the segment hold -inf
  m = (L.j < cnt) ? m : -INFINITY;
#pragma unroll
  for (int grp = 1; grp < 4; ++grp)
    if (grp * 16 < cnt) m = (grp * 16 + L.j < cnt) ? fmaxf(m, sc[grp]) : m;
  m = row16_max(m);
  float ex[4], sum = 0.f;
#pragma unroll
  for (int grp = 0; grp < 4; ++grp) {
    ex[grp] = 0.f;
    if (grp * 16 < cnt) {
      ex[grp] = (grp * 16 + L.j < cnt) ? expf(sc[grp] - m) : 0.f;
      sum += ex[grp];
    }
  }
  sum = row16_sum(sum) + 1e-16f;
#pragma unroll
  for (int grp = 0; grp < 4; ++grp) {
    if (grp * 16 < cnt) {
      if (grp * 16 + L.j < cnt) {
        const float av = ex[grp] / sum;
        const int64_t o = (int64_t)(beg + grp * 16 + L.j) * 4 + L.g;
        a[o] = av;
        alpha[o] = av * deg;
      }
    }
  }
}

// the wave's maximum |.| of the row it wrote -> part[node] (a plain store; k_amax_reduce folds the N entries: common.h says why)
__device__ __forceinline__ void node_amax(float m, float* __restrict__ part, int node) {
  m = wave_amax_lane63(m);
  if ((threadIdx.x & 63) == 63) part[node] = m;
}

// forward 2/2: weighted sum of messages, one wave per TARGET node (gathers M[src] and Em[cls]; no atomics)
__global__ __launch_bounds__(256) void k_edge_aggregate(const int* __restrict__ rowptr_t, const int* __restrict__ src_t,
                                                        const int* __restrict__ cls_t, const int* __restrict__ pos_t,
                                                        const float* __restrict__ KMQ, int ldk, const float* __restrict__ EkEm,
                                                        int lde, int HP, const float* __restrict__ alpha,
                                                        float* __restrict__ aggr, int lda, int N, int C, const int* __restrict__ xcd_base,
                                                        float* __restrict__ amax_part) {
  __shared__ float4 slab[4][SLAB_ROWS];
  const int t = wave_node(xcd_base);
  if (t >= N) return;
  const Lane L = lane_info(HP);
  const int DP = 4 * HP;
  const uint32_t pk = (uint32_t)ldk * 4u, pe = (uint32_t)lde * 4u, vm = L.voff + (uint32_t)DP * 4u;  // M | Em halves
  const rsrc_t rK = make_rsrc(KMQ, (uint32_t)N * pk), rE = make_rsrc(EkEm, (uint32_t)C * pe);
  const int beg = __builtin_amdgcn_readfirstlane(rowptr_t[t]), end = __builtin_amdgcn_readfirstlane(rowptr_t[t + 1]);
  if (end - beg == 1) {
    // one in-edge: the node's own self loop (every PAD row, 40 % of a CSQA batch): no chunk staging, no LDS
    const int sv = __builtin_amdgcn_readfirstlane(src_t[beg]), cv = __builtin_amdgcn_readfirstlane(cls_t[beg]);
    const int pv = __builtin_amdgcn_readfirstlane(pos_t[beg]);
    const float4 m1 = buf_ld4(rK, vm, (uint32_t)sv * pk), em1 = buf_ld4(rE, vm, (uint32_t)cv * pe);
    const float al = alpha[(int64_t)pv * 4 + L.g];
    const float4 o1 = fma4(al, add4(m1, em1), zero4());
    if (L.act) st4(aggr + (int64_t)t * lda + L.off, o1);
    if (amax_part) node_amax(L.act ? absmax4(o1) : 0.f, amax_part, t);  // (the operand maximum of the GEMMs that read aggr: common.h)
    return;
  }
  slab_init(slab[L.w], L.lane);
  const float* const sl = reinterpret_cast<const float*>(slab[L.w]) + L.g;
  float4 acc = zero4();
  for (int e0 = beg; e0 < end; e0 += 64) {
    const int cnt = min(64, end - e0), lc = e0 + min(L.lane, cnt - 1);
    const uint32_t sv = (uint32_t)src_t[lc] * pk, cv = (uint32_t)cls_t[lc] * pe;
    // lane i: alpha of edge e0 + i (one 16-byte gather per lane and chunk); lanes past the chunk hold weight 0
    slab[L.w][L.lane] = L.lane < cnt ? ld4(alpha + (int64_t)pos_t[lc] * 4) : zero4();
    for (int i = 0; i < cnt; i += EDGE_UNROLL) {
      float4 m[EDGE_UNROLL], em[EDGE_UNROLL];
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        const int idx = min(i + u, cnt - 1);
        m[u] = buf_ld4(rK, vm, rl(sv, idx));
        em[u] = buf_ld4(rE, vm, rl(cv, idx));
      }
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) acc = fma4(sl[(i + u) * 4], add4(m[u], em[u]), acc);
    }
  }
  if (L.act) st4(aggr + (int64_t)t * lda + L.off, acc);
  if (amax_part) node_amax(L.act ? absmax4(acc) : 0.f, amax_part, t);
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
                                                       float* __restrict__ ga, float* __restrict__ rs, int N, int C,
                                                       const int* __restrict__ xcd_base, float* __restrict__ amax_part) {
  __shared__ float4 slab[3][4][SLAB_ROWS];  // a | alpha | ga of the current chunk
  const int s = wave_node(xcd_base);
  if (s >= N) return;
  const Lane L = lane_info(HP);
  const int DP = 4 * HP;
  const uint32_t pk = (uint32_t)ldk * 4u, pe = (uint32_t)lde * 4u, pg = (uint32_t)ldg * 4u, vm = L.voff + (uint32_t)DP * 4u;
  const rsrc_t rK = make_rsrc(KMQ, (uint32_t)N * pk), rE = make_rsrc(EkEm, (uint32_t)C * pe), rG = make_rsrc(G, (uint32_t)N * pg);
  const int beg = __builtin_amdgcn_readfirstlane(rowptr_s[s]), end = __builtin_amdgcn_readfirstlane(rowptr_s[s + 1]);
  if (end - beg == 1) {
    // the self loop alone: a = alpha = 1, so dM[s] = G[s]; its softmax gradient gs = qscale * a * (ga - a * ga) is exactly 0, which
    // src pass 2 reproduces from ga = 0, rs = 0 without reading a row (and then leaves dQ[s] = 0, gs = 0 behind)
    const float4 g1 = buf_ld4(rG, L.voff, (uint32_t)s * pg);
    if (L.act) st4(dKMQ + (int64_t)s * ldk + DP + L.off, g1);
    if (amax_part) node_amax(L.act ? absmax4(g1) : 0.f, amax_part, s);
    if (L.lane == 0) {
      st4(ga + (int64_t)beg * 4, zero4());
      st4(rs + (int64_t)s * 4, zero4());
    }
    return;
  }
  const float4 mrow = buf_ld4(rK, vm, (uint32_t)s * pk);
  const float deg = (float)(end - beg);
  slab_init(slab[0][L.w], L.lane);
  slab_init(slab[1][L.w], L.lane);
  const float* const sl_a = reinterpret_cast<const float*>(slab[0][L.w]) + L.g;
  const float* const sl_al = reinterpret_cast<const float*>(slab[1][L.w]) + L.g;
  float* const sl_ga = reinterpret_cast<float*>(slab[2][L.w]) + L.g;
  float4 dM = zero4();
  float r = 0.f;
  for (int e0 = beg; e0 < end; e0 += 64) {
    const int cnt = min(64, end - e0), lc = e0 + min(L.lane, cnt - 1);
    const uint32_t tv = (uint32_t)tgt_s[lc] * pg, cv = (uint32_t)cls_s[lc] * pe;
    slab[0][L.w][L.lane] = L.lane < cnt ? ld4(a + (int64_t)lc * 4) : zero4();
    slab[1][L.w][L.lane] = L.lane < cnt ? ld4(alpha + (int64_t)lc * 4) : zero4();
    for (int i = 0; i < cnt; i += EDGE_UNROLL) {
      float4 g4[EDGE_UNROLL], em[EDGE_UNROLL];
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        const int idx = min(i + u, cnt - 1);
        g4[u] = buf_ld4(rG, L.voff, rl(tv, idx));
        em[u] = buf_ld4(rE, vm, rl(cv, idx));
      }
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {  // clamped duplicates past the chunk enter with a = alpha = 0
        dM = fma4(sl_al[(i + u) * 4], g4[u], dM);
        const float gae = deg * row16_sum(dot4(add4(mrow, em[u]), g4[u]));
        r = fmaf(sl_a[(i + u) * 4], gae, r);
        if (L.j == 0) sl_ga[(i + u) * 4] = gae;
      }
    }
    const float4 gv = slab[2][L.w][L.lane];
    if (L.lane < cnt) st4(ga + (int64_t)lc * 4, gv);
  }
  if (L.act) st4(dKMQ + (int64_t)s * ldk + DP + L.off, dM);
  if (amax_part) node_amax(L.act ? absmax4(dM) : 0.f, amax_part, s);
  if (L.j == 0) rs[(int64_t)s * 4 + L.g] = r;
}

__global__ __launch_bounds__(256) void k_edge_bwd_src2(const int* __restrict__ rowptr_s, const int* __restrict__ tgt_s,
                                                       const int* __restrict__ cls_s, const float* __restrict__ KMQ, int ldk,
                                                       const float* __restrict__ EkEm, int lde, int HP, float qscale,
                                                       const float* __restrict__ a, float* __restrict__ dKMQ,
                                                       float* __restrict__ ga, const float* __restrict__ rs, int N, int C,
                                                       const int* __restrict__ xcd_base, float* __restrict__ amax_part) {
  __shared__ float4 slab[4][SLAB_ROWS];
  const int s = wave_node(xcd_base);
  if (s >= N) return;
  const Lane L = lane_info(HP);
  const int DP = 4 * HP;
  const uint32_t pk = (uint32_t)ldk * 4u, pe = (uint32_t)lde * 4u;
  const rsrc_t rK = make_rsrc(KMQ, (uint32_t)N * pk), rE = make_rsrc(EkEm, (uint32_t)C * pe);
  const int beg = __builtin_amdgcn_readfirstlane(rowptr_s[s]), end = __builtin_amdgcn_readfirstlane(rowptr_s[s + 1]);
  if (end - beg == 1) {  // the self loop alone: gs = 0 (src pass 1 left ga = 0 there), so dQ[s] = 0
    if (L.act) st4(dKMQ + (int64_t)s * ldk + 2 * DP + L.off, zero4());
    if (amax_part && L.lane == 63) amax_part[s] = 0.f;
    return;
  }
  const float4 r4 = ld4(rs + (int64_t)s * 4);  // the node's 4 head values, same address in every lane
  slab_init(slab[L.w], L.lane);
  const float* const sl = reinterpret_cast<const float*>(slab[L.w]) + L.g;
  float4 dQ = zero4();
  for (int e0 = beg; e0 < end; e0 += 64) {
    const int cnt = min(64, end - e0), lc = e0 + min(L.lane, cnt - 1);
    const uint32_t tv = (uint32_t)tgt_s[lc] * pk, cv = (uint32_t)cls_s[lc] * pe;
    // lane i: gs of edge e0 + i from its a and ga (coalesced), written back in place of ga (coalesced) and to the slab
    float4 gs4 = zero4();
    if (L.lane < cnt) {
      const float4 a4 = ld4(a + (int64_t)lc * 4), g4 = ld4(ga + (int64_t)lc * 4);
      gs4 = make_float4(qscale * a4.x * (g4.x - r4.x), qscale * a4.y * (g4.y - r4.y), qscale * a4.z * (g4.z - r4.z),
                        qscale * a4.w * (g4.w - r4.w));
      st4(ga + (int64_t)lc * 4, gs4);
    }
    slab[L.w][L.lane] = gs4;
    for (int i = 0; i < cnt; i += EDGE_UNROLL) {
      float4 k[EDGE_UNROLL], ek[EDGE_UNROLL];
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        const int idx = min(i + u, cnt - 1);
        k[u] = buf_ld4(rK, L.voff, rl(tv, idx));
        ek[u] = buf_ld4(rE, L.voff, rl(cv, idx));
      }
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) dQ = fma4(sl[(i + u) * 4], add4(k[u], ek[u]), dQ);
    }
  }
  if (L.act) st4(dKMQ + (int64_t)s * ldk + 2 * DP + L.off, dQ);
  if (amax_part) node_amax(L.act ? absmax4(dQ) : 0.f, amax_part, s);
}

__global__ __launch_bounds__(256) void k_edge_bwd_tgt(const int* __restrict__ rowptr_t, const int* __restrict__ src_t,
                                                      const int* __restrict__ pos_t, const float* __restrict__ KMQ, int ldk,
                                                      int HP, const float* __restrict__ gsb, float* __restrict__ dKMQ, int N,
                                                      const int* __restrict__ xcd_base, float* __restrict__ amax_part) {
  __shared__ float4 slab[4][SLAB_ROWS];
  const int t = wave_node(xcd_base);
  if (t >= N) return;
  const Lane L = lane_info(HP);
  const int DP = 4 * HP;
  const uint32_t pk = (uint32_t)ldk * 4u, vq = L.voff + 2u * DP * 4u;
  const rsrc_t rK = make_rsrc(KMQ, (uint32_t)N * pk);
  const int beg = __builtin_amdgcn_readfirstlane(rowptr_t[t]), end = __builtin_amdgcn_readfirstlane(rowptr_t[t + 1]);
  if (end - beg == 1) {  // one in-edge (the self loop): dK[t] = gs * Q[src], no staging
    const int sv = __builtin_amdgcn_readfirstlane(src_t[beg]), pv = __builtin_amdgcn_readfirstlane(pos_t[beg]);
    const float4 q1 = buf_ld4(rK, vq, (uint32_t)sv * pk);
    const float gs1 = gsb[(int64_t)pv * 4 + L.g];
    const float4 k1 = fma4(gs1, q1, zero4());
    if (L.act) st4(dKMQ + (int64_t)t * ldk + L.off, k1);
    if (amax_part) node_amax(L.act ? absmax4(k1) : 0.f, amax_part, t);
    return;
  }
  slab_init(slab[L.w], L.lane);
  const float* const sl = reinterpret_cast<const float*>(slab[L.w]) + L.g;
  float4 dK = zero4();
  for (int e0 = beg; e0 < end; e0 += 64) {
    const int cnt = min(64, end - e0), lc = e0 + min(L.lane, cnt - 1);
    const uint32_t sv = (uint32_t)src_t[lc] * pk;
    slab[L.w][L.lane] = L.lane < cnt ? ld4(gsb + (int64_t)pos_t[lc] * 4) : zero4();  // lane i: gs of edge e0 + i
    for (int i = 0; i < cnt; i += TGT_UNROLL) {
      float4 qv[TGT_UNROLL];
#pragma unroll
      for (int u = 0; u < TGT_UNROLL; ++u) qv[u] = buf_ld4(rK, vq, rl(sv, min(i + u, cnt - 1)));
#pragma unroll
      for (int u = 0; u < TGT_UNROLL; ++u) dK = fma4(sl[(i + u) * 4], qv[u], dK);
    }
  }
  if (L.act) st4(dKMQ + (int64_t)t * ldk + L.off, dK);
  if (amax_part) node_amax(L.act ? absmax4(dK) : 0.f, amax_part, t);
}

// one wave per class chunk (<= QAGNN_CLS_CHUNK edges of one class inside one position group), walked 64 edges at a time
// (chunks of 256 edges were measured: 4x fewer partials for the big classes, but the class pass itself went 65 -> 89 us).
__global__ __launch_bounds__(256) void k_edge_bwd_cls(const int* __restrict__ n_chunks, const int* __restrict__ chunk_beg,
                                                      const int* __restrict__ chunk_len, const int* __restrict__ src_c,
                                                      const int* __restrict__ tgt_c, const int* __restrict__ pos_c,
                                                      const float* __restrict__ KMQ, int ldk, int HP,
                                                      const float* __restrict__ alpha, const float* __restrict__ gsb,
                                                      const float* __restrict__ G, int ldg, float* __restrict__ cls_part, int N) {
  __shared__ float4 slab[2][4][SLAB_ROWS];  // gs | alpha of the current 64 edges
  // chunks are in (position group, class) order: the remap gives every XCD a contiguous run of groups, whose node rows its L2
  // then serves (the grid is sized for max_chunks, so the remap runs over the blocks that really have a chunk)
  const int nch = *n_chunks, nb_real = (nch + 3) >> 2;
  if ((int)blockIdx.x >= nb_real) return;
  const int k = __builtin_amdgcn_readfirstlane(xcd_remap(blockIdx.x, nb_real) * 4 + (threadIdx.x >> 6));
  if (k >= nch) return;
  const Lane L = lane_info(HP);
  const int DP = 4 * HP;
  const uint32_t pk = (uint32_t)ldk * 4u, pg = (uint32_t)ldg * 4u, vq = L.voff + 2u * DP * 4u;
  const rsrc_t rK = make_rsrc(KMQ, (uint32_t)N * pk), rG = make_rsrc(G, (uint32_t)N * pg);
  const int beg = __builtin_amdgcn_readfirstlane(chunk_beg[k]), len = __builtin_amdgcn_readfirstlane(chunk_len[k]);
  slab_init(slab[0][L.w], L.lane);
  slab_init(slab[1][L.w], L.lane);
  const float* const sl_gs = reinterpret_cast<const float*>(slab[0][L.w]) + L.g;
  const float* const sl_al = reinterpret_cast<const float*>(slab[1][L.w]) + L.g;
  float4 dEk = zero4(), dEm = zero4();
  for (int b0 = 0; b0 < len; b0 += 64) {
    const int cnt = min(64, len - b0), lc = beg + b0 + min(L.lane, cnt - 1);
    const uint32_t sv = (uint32_t)src_c[lc] * pk, tv = (uint32_t)tgt_c[lc] * pg;
    const int pv = pos_c[lc];
    slab[0][L.w][L.lane] = L.lane < cnt ? ld4(gsb + (int64_t)pv * 4) : zero4();
    slab[1][L.w][L.lane] = L.lane < cnt ? ld4(alpha + (int64_t)pv * 4) : zero4();
    for (int i = 0; i < cnt; i += EDGE_UNROLL) {
      float4 qv[EDGE_UNROLL], g4[EDGE_UNROLL];
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        const int idx = min(i + u, cnt - 1);
        qv[u] = buf_ld4(rK, vq, rl(sv, idx));
        g4[u] = buf_ld4(rG, L.voff, rl(tv, idx));
      }
#pragma unroll
      for (int u = 0; u < EDGE_UNROLL; ++u) {
        dEk = fma4(sl_gs[(i + u) * 4], qv[u], dEk);
        dEm = fma4(sl_al[(i + u) * 4], g4[u], dEm);
      }
    }
  }
  if (L.act) {
    st4(cls_part + (int64_t)k * 2 * DP + L.off, dEk);
    st4(cls_part + (int64_t)k * 2 * DP + DP + L.off, dEm);
  }
}

// ordered sum of a class's chunk partials over all position groups, two stages.  Stage 1: block (c, s) sums the chunks of class c
// in the groups g = s, s + S, ... (S = QAGNN_CLS_SLICES): columns x partitions, partition p takes the slice's chunks number
// p, p + P, ..., partitions added in order.  Stage 2 adds the S slices in order.  (One block per class left the largest class --
// the self loops of the KG nodes, ~900 partials -- as a 100-deep chain of dependent loads: 58 of the stage's 62 us.)
__global__ __launch_bounds__(1024) void k_cls_reduce(const int* __restrict__ chunkptr, const float* __restrict__ cls_part,
                                                     float* __restrict__ slices, int DP2, int C, int NG) {
  extern __shared__ float4 sm4[];
  // the class's chunk ranges of all groups first (one load per thread, not 2 dependent loads per group and thread)
  __shared__ int cp_lo[QAGNN_CLS_GROUPS], cp_hi[QAGNN_CLS_GROUPS];
  const int c = blockIdx.x, sl = blockIdx.y;
  if (threadIdx.x < NG) {
    cp_lo[threadIdx.x] = chunkptr[threadIdx.x * C + c];
    cp_hi[threadIdx.x] = chunkptr[threadIdx.x * C + c + 1];
  }
  __syncthreads();
  const int ncol4 = DP2 >> 2;
  const int P = (int)blockDim.x / ncol4;  // partitions: 9 at 1024 threads and DP2 = 416, 2 at 256 threads (small graphs, see the launch)
  const int col4 = threadIdx.x % ncol4, part = threadIdx.x / ncol4;
  float4 acc = zero4();
  if (part < P) {
    int seen = 0;  // chunks of this class in the slice's groups before g
    for (int g = sl; g < NG; g += QAGNN_CLS_SLICES) {
      const int kb = cp_lo[g], ke = cp_hi[g];
      const int first = (part - seen % P + P) % P;
      for (int k = kb + first; k < ke; k += P) acc = add4(acc, ld4(cls_part + (int64_t)k * DP2 + col4 * 4));
      seen += ke - kb;
    }
    sm4[part * ncol4 + col4] = acc;
  }
  __syncthreads();
  if (threadIdx.x < ncol4) {
    float4 s = sm4[threadIdx.x];
    for (int q = 1; q < P; ++q) s = add4(s, sm4[q * ncol4 + threadIdx.x]);
    st4(slices + ((int64_t)c * QAGNN_CLS_SLICES + sl) * DP2 + threadIdx.x * 4, s);
  }
}
// amax_part != nullptr: the launch also folds the node-side kernels' per-node maxima (npart non-negative floats) into *amax_slot -- every
// block a slice, one atomic per block (common.h): the last small kernel of the edge backward carries the reduction instead of a launch of its own
__global__ __launch_bounds__(256) void k_cls_reduce2(const float* __restrict__ slices, float* __restrict__ dEkEm, int lde, int DP2, int C,
                                                     const float* __restrict__ amax_part, int npart, uint32_t* __restrict__ amax_slot) {
  const int ncol4 = DP2 >> 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < C * ncol4) {
    const int c = i / ncol4, col4 = i - c * ncol4;
    const float* p = slices + (int64_t)c * QAGNN_CLS_SLICES * DP2 + col4 * 4;
    float4 s = ld4(p);
#pragma unroll
    for (int q = 1; q < QAGNN_CLS_SLICES; ++q) s = add4(s, ld4(p + (int64_t)q * DP2));
    st4(dEkEm + (int64_t)c * lde + col4 * 4, s);
  }
  if (amax_part) {
    __shared__ float red[16];
    float m = 0.f;
    for (int k = i; k < npart; k += (int)(gridDim.x * blockDim.x)) m = fmaxf(m, amax_part[k]);
    block_amax_merge(m, amax_slot, red);
  }
}

static int check_common(const qagnn_graph* g, const float* KMQ, int ldk, const float* EkEm, int lde, int HP, const char* who) {
  QAGNN_REQUIRE(g && KMQ && EkEm, QAGNN_EINVAL, "%s: null pointer", who);
  QAGNN_REQUIRE(HP > 0 && HP % 4 == 0 && HP <= 64, QAGNN_EUNSUPPORTED, "%s: head pitch HP=%d must be a multiple of 4, <= 64", who, HP);
  QAGNN_REQUIRE(ldk >= 12 * HP && ldk % 4 == 0 && lde >= 8 * HP && lde % 4 == 0, QAGNN_EINVAL, "%s: ldk=%d lde=%d too small for HP=%d",
                who, ldk, lde, HP);
  QAGNN_REQUIRE(aligned16(KMQ) && aligned16(EkEm), QAGNN_EINVAL, "%s: operands must be 16-byte aligned", who);
  // rows are addressed through 32-bit buffer offsets (idle lanes use OOB_OFF, which must stay out of range)
  QAGNN_REQUIRE((int64_t)g->N * ldk * 4 < (int64_t)OOB_OFF && (int64_t)g->C * lde * 4 < (int64_t)OOB_OFF, QAGNN_EUNSUPPORTED,
                "%s: N=%d x ldk=%d floats exceed the 2 GiB a buffer descriptor of these kernels addresses", who, g->N, ldk);
  return QAGNN_OK;
}

}  // namespace qagnn

using namespace qagnn;

namespace qagnn {
// amax_part != nullptr: [N] floats, max |aggr row| per node (k_edge_aggregate; launch_amax_reduce folds them into the word)
int launch_edge_attn_fwd(const qagnn_graph* g, const float* KMQ, int32_t ldk, const float* EkEm, int32_t lde, int32_t HP, float qscale,
                         float* score, float* a, float* alpha, float* aggr, int32_t lda, float* amax_part, hipStream_t stream) {
  TimedScope timed(2, stream);
  int rc = check_common(g, KMQ, ldk, EkEm, lde, HP, "edge_attn_fwd");
  if (rc != QAGNN_OK) return rc;
  QAGNN_REQUIRE(score && a && alpha && aggr && lda >= 4 * HP && lda % 4 == 0 && aligned16(aggr), QAGNN_EINVAL,
                "edge_attn_fwd: bad output arguments");
  const int nb = 8 * edge_xcd_cap(g->N);  // (8 runs of at most edge_xcd_cap blocks: see k_xcd_partition)
  k_edge_scores<<<nb, 256, 0, stream>>>(g->rowptr_s, g->tgt_s, g->cls_s, KMQ, ldk, EkEm, lde, HP, qscale, score, a, alpha, g->N, g->C, g->err + 4);
  QAGNN_LAUNCH_CHECK("k_edge_scores");
  k_edge_aggregate<<<nb, 256, 0, stream>>>(g->rowptr_t, g->src_t, g->cls_t, g->pos_t, KMQ, ldk, EkEm, lde, HP, alpha, aggr, lda, g->N, g->C, g->err + 4, amax_part);
  QAGNN_LAUNCH_CHECK("k_edge_aggregate");
  return QAGNN_OK;
}
}  // namespace qagnn

extern "C" int qagnn_edge_attn_fwd_f32(const qagnn_graph* g, const float* KMQ, int32_t ldk, const float* EkEm, int32_t lde,
                                       int32_t HP, float qscale, float* score, float* a, float* alpha, float* aggr, int32_t lda,
                                       qagnn_stream_t stream_) {
  return launch_edge_attn_fwd(g, KMQ, ldk, EkEm, lde, HP, qscale, score, a, alpha, aggr, lda, nullptr, (hipStream_t)stream_);
}

extern "C" int qagnn_edge_attn_bwd_f32(const qagnn_graph* g, const float* KMQ, int32_t ldk, const float* EkEm, int32_t lde,
                                       int32_t HP, float qscale, const float* a, const float* alpha, const float* G, int32_t ldg,
                                       float* dKMQ, float* dEkEm, float* ga, float* rs, float* cls_part, qagnn_stream_t stream_) {
  return launch_edge_attn_bwd(g, KMQ, ldk, EkEm, lde, HP, qscale, a, alpha, G, ldg, dKMQ, dEkEm, ga, rs, cls_part, nullptr, nullptr, (hipStream_t)stream_);
}

// amax_part != nullptr: [3 N] floats of scratch for max |d M row|, |d Q row|, |d K row| per node (the three node-side kernels), folded into
// *amax_slot (max |d K|M|Q|, bit pattern; the caller zeroed it) by the last kernel of the call
int qagnn::launch_edge_attn_bwd(const qagnn_graph* g, const float* KMQ, int32_t ldk, const float* EkEm, int32_t lde, int32_t HP, float qscale,
                                const float* a, const float* alpha, const float* G, int32_t ldg, float* dKMQ, float* dEkEm, float* ga, float* rs,
                                float* cls_part, float* amax_part, uint32_t* amax_slot, hipStream_t stream) {
  TimedScope timed(3, stream);
  int rc = check_common(g, KMQ, ldk, EkEm, lde, HP, "edge_attn_bwd");
  if (rc != QAGNN_OK) return rc;
  QAGNN_REQUIRE(a && alpha && G && dKMQ && dEkEm && ga && rs && cls_part, QAGNN_EINVAL, "edge_attn_bwd: null pointer");
  QAGNN_REQUIRE(ldg >= 4 * HP && ldg % 4 == 0 && aligned16(G) && aligned16(dKMQ) && aligned16(dEkEm) && aligned16(cls_part),
                QAGNN_EINVAL, "edge_attn_bwd: bad pitch / alignment");
  QAGNN_REQUIRE((int64_t)g->N * ldg * 4 < (int64_t)OOB_OFF, QAGNN_EUNSUPPORTED, "edge_attn_bwd: G exceeds 2 GiB");
  const int DP2 = 8 * HP;
  QAGNN_REQUIRE(DP2 / 4 <= 1024, QAGNN_EUNSUPPORTED, "edge_attn_bwd: HP too large");
  const int nb = 8 * edge_xcd_cap(g->N);  // (8 runs of at most edge_xcd_cap blocks: see k_xcd_partition)
  k_edge_bwd_src1<<<nb, 256, 0, stream>>>(g->rowptr_s, g->tgt_s, g->cls_s, KMQ, ldk, EkEm, lde, HP, a, alpha, G, ldg, dKMQ, ga, rs, g->N, g->C, g->err + 4, amax_part);
  QAGNN_LAUNCH_CHECK("k_edge_bwd_src1");
  k_edge_bwd_src2<<<nb, 256, 0, stream>>>(g->rowptr_s, g->tgt_s, g->cls_s, KMQ, ldk, EkEm, lde, HP, qscale, a, dKMQ, ga, rs, g->N, g->C, g->err + 4, amax_part ? amax_part + g->N : nullptr);
  QAGNN_LAUNCH_CHECK("k_edge_bwd_src2");
  k_edge_bwd_tgt<<<nb, 256, 0, stream>>>(g->rowptr_t, g->src_t, g->pos_t, KMQ, ldk, HP, ga, dKMQ, g->N, g->err + 4, amax_part ? amax_part + 2 * (int64_t)g->N : nullptr);
  QAGNN_LAUNCH_CHECK("k_edge_bwd_tgt");
  k_edge_bwd_cls<<<cdiv(g->max_chunks, 4), 256, 0, stream>>>(g->n_chunks, g->chunk_beg, g->chunk_len, g->src_c, g->tgt_c, g->pos_c,
                                                             KMQ, ldk, HP, alpha, ga, G, ldg, cls_part, g->N);
  QAGNN_LAUNCH_CHECK("k_edge_bwd_cls");
  // 1024 threads = 9 partitions per (class, slice) where the large classes have hundreds of chunk partials; a small graph (the
  // reference's 10-subgraph mini-batch: a handful of chunks per class) only pays for launching 2 448 x 16 waves: 256 threads there.
  // (The choice follows g->Ep, the capacity the arrays are laid out for: a fixed function of the launch shape, like every other grid here.)
  const int cr_threads = g->Ep < 65536 && DP2 / 4 <= 128 ? 256 : 1024;
  const int P = cr_threads / (DP2 / 4);
  float* slices = cls_part + (int64_t)g->max_chunks * DP2;  // [C][QAGNN_CLS_SLICES][DP2] behind the chunk partials
  k_cls_reduce<<<dim3(g->C, QAGNN_CLS_SLICES), cr_threads, (size_t)P * (DP2 / 4) * sizeof(float4), stream>>>(g->chunkptr, cls_part, slices, DP2,
                                                                                                          g->C, g->n_groups);
  QAGNN_LAUNCH_CHECK("k_cls_reduce");
  k_cls_reduce2<<<cdiv((int64_t)g->C * (DP2 / 4), 256), 256, 0, stream>>>(slices, dEkEm, lde, DP2, g->C, amax_slot ? amax_part : nullptr, 3 * g->N, amax_slot);
  QAGNN_LAUNCH_CHECK("k_cls_reduce2");
  return QAGNN_OK;
}
