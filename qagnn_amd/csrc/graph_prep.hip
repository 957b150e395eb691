// Graph preparation for the batched QA subgraphs (integer work, once per batch).
//
// Builds three deterministic orderings of the E' = E + N edges (caller edges + one self loop per node row):
// grouped by source (softmax segments), by target (aggregation segments) and by edge class (gradients of the
// per-class tables).  Inside every group edges are sorted by edge id, so all floating-point reductions done
// by the edge kernels run in a fixed order that matches the reference's CPU order (index_add_ in edge order).
//
// Reference semantics being replaced: modeling/modeling_qagnn.py:419-438 (one-hots, head/tail type lookup,
// self loops) and :476-479 (out-degree); PyG's implicit grouping in softmax()/scatter().
#include <stdarg.h>

#include <stdlib.h>

#include "common.h"

namespace qagnn {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- decode + histogram ---------------------------------------------------------------------------------
// class ids of tens of thousands of self loops collide on a handful of counters: the class histogram is taken from
// the per-block LDS histograms of k_cls_hist instead of one global atomic per edge.
__global__ void k_decode_count(const int64_t* __restrict__ edge_index, const int64_t* __restrict__ edge_type,
                               const int64_t* __restrict__ node_type, int N, int E, int R, int T, int* __restrict__ es,
                               int* __restrict__ et, int* __restrict__ ec, int* __restrict__ cnt_s, int* __restrict__ cnt_t,
                               int* __restrict__ err, int block_n) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int Ep = E + N;
  if (e >= Ep) return;
  int s, t, c;
  bool bad = false;
  if (e < E) {
    int64_t s64 = edge_index[e], t64 = edge_index[(int64_t)E + e], r64 = edge_type[e];
    bad = s64 < 0 || s64 >= N || t64 < 0 || t64 >= N || r64 < 0 || r64 >= R;
    s = (int)min(max(s64, (int64_t)0), (int64_t)N - 1);
    t = (int)min(max(t64, (int64_t)0), (int64_t)N - 1);
    int r = (int)min(max(r64, (int64_t)0), (int64_t)R - 1);
    int64_t hs = node_type[s], ht = node_type[t];
    bad = bad || hs < 0 || hs >= T || ht < 0 || ht >= T;
    int h = (int)min(max(hs, (int64_t)0), (int64_t)T - 1), tl = (int)min(max(ht, (int64_t)0), (int64_t)T - 1);
    c = r * T * T + h * T + tl;
  } else {
    s = t = e - E;
    int64_t hs = node_type[s];
    bad = hs < 0 || hs >= T;
    c = R * T * T + (int)min(max(hs, (int64_t)0), (int64_t)T - 1);
  }
  if (bad) *err = 1;
  if (block_n > 0 && s / block_n != t / block_n) err[1] = 1;  // an edge leaves its block of block_n node rows
  es[e] = s;
  et[e] = t;
  ec[e] = c;
  atomicAdd(&cnt_s[s], 1);
  atomicAdd(&cnt_t[t], 1);
}

// ---- exclusive scan of up to 3 independent arrays, one 1024-thread block each ---------------------------------
// A thread owns SCAN_ITEMS consecutive elements per round (serial sum, then one block-wide scan of the 1024 thread totals):
// 8192 elements per round instead of 1024, so the 64 000-entry degree arrays take 8 rounds of 3 barriers, not 63.
constexpr int SCAN_ITEMS = 8;
__device__ void block_exclusive_scan(const int* __restrict__ in, int* __restrict__ out, int n) {
  __shared__ int wtot[16];
  __shared__ int carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024 * SCAN_ITEMS) {
    const int i0 = base + tid * SCAN_ITEMS;
    int v[SCAN_ITEMS], tsum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
      v[k] = i0 + k < n ? in[i0 + k] : 0;
      tsum += v[k];
    }
    int incl = tsum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int u = __shfl_up(incl, o, 64);
      if (lane >= o) incl += u;
    }
    if (lane == 63) wtot[wid] = incl;
    __syncthreads();
    if (wid == 0) {
      int w = lane < 16 ? wtot[lane] : 0;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
        int u = __shfl_up(w, o, 64);
        if (lane >= o) w += u;
      }
      if (lane < 16) wtot[lane] = w;  // inclusive over waves
    }
    __syncthreads();
    const int carry = carry_s;
    int run = carry + (wid ? wtot[wid - 1] : 0) + incl - tsum;  // exclusive prefix of this thread's first element
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
      if (i0 + k < n) out[i0 + k] = run;
      run += v[k];
    }
    __syncthreads();
    if (tid == 0) carry_s = carry + wtot[15];
    __syncthreads();
  }
  if (tid == 0) out[n] = carry_s;
}

__global__ __launch_bounds__(1024) void k_scan3(const int* a0, int* o0, int n0, const int* a1, int* o1, int n1, const int* a2,
                                                int* o2, int n2) {
  if (blockIdx.x == 0) block_exclusive_scan(a0, o0, n0);
  else if (blockIdx.x == 1) block_exclusive_scan(a1, o1, n1);
  else block_exclusive_scan(a2, o2, n2);
}

// ---- bucket fill (unordered inside a bucket; fixed up by k_sort_segments) --------------------------------------
__global__ void k_fill(const int* __restrict__ es, const int* __restrict__ et, const int* __restrict__ rowptr_s,
                       const int* __restrict__ rowptr_t, int* __restrict__ cnt_s, int* __restrict__ cnt_t,
                       int* __restrict__ tmp_s, int* __restrict__ tmp_t, int Ep) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= Ep) return;
  const int s = es[e], t = et[e];
  tmp_s[rowptr_s[s] + atomicSub(&cnt_s[s], 1) - 1] = e;
  tmp_t[rowptr_t[t] + atomicSub(&cnt_t[t], 1) - 1] = e;
}

// one wave per (node, direction): rank-by-counting sort of the bucket by edge id
__global__ __launch_bounds__(256) void k_sort_segments(const int* __restrict__ rowptr_s, const int* __restrict__ rowptr_t,
                                                       const int* __restrict__ tmp_s, const int* __restrict__ tmp_t,
                                                       int* __restrict__ eid_s, int* __restrict__ eid_t,
                                                       int* __restrict__ srcpos, int N) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (w >= 2 * N) return;
  const bool by_src = w < N;
  const int v = by_src ? w : w - N;
  const int* rowptr = by_src ? rowptr_s : rowptr_t;
  const int* tmp = by_src ? tmp_s : tmp_t;
  int* out = by_src ? eid_s : eid_t;
  const int beg = rowptr[v], L = rowptr[v + 1] - beg;
  for (int i = lane; i < L; i += 64) {
    const int id = tmp[beg + i];
    int rank = 0;
    for (int j = 0; j < L; ++j) rank += tmp[beg + j] < id;
    out[beg + rank] = id;
    if (by_src) srcpos[id] = beg + rank;
  }
}

__global__ void k_payload(const int* __restrict__ es, const int* __restrict__ et, const int* __restrict__ ec,
                          const int* __restrict__ eid_s, const int* __restrict__ eid_t, const int* __restrict__ srcpos,
                          int* __restrict__ tgt_s, int* __restrict__ src_s, int* __restrict__ cls_s, int* __restrict__ src_t,
                          int* __restrict__ tgt_t, int* __restrict__ cls_t, int* __restrict__ pos_t, int Ep) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= Ep) return;
  const int e = eid_s[p];
  tgt_s[p] = et[e];
  src_s[p] = es[e];
  cls_s[p] = ec[e];
  const int e2 = eid_t[p];
  src_t[p] = es[e2];
  tgt_t[p] = et[e2];
  cls_t[p] = ec[e2];
  pos_t[p] = srcpos[e2];
}


// ---- graph orderings from per-sample blobs built at LOAD time ----------------------------------
// Everything graph_prep derives per batch with five sorting launches -- source order, target order, the position of every edge in
// the source order, class ids, degrees -- is static per dataset sample (the reference re-slices the same lists every
// epoch, utils/data_utils.py:53-76, and offsets them per batch, modeling_qagnn.py:244-251).  The loader therefore stores, per
// (question, choice) sample, a blob of int32 words
//     cnt_s[n] | cnt_t[n] | w0[E_g] | w1[E_g] | w2[E_g]
//     cnt_s / cnt_t   out- / in-degree of the n node slots (real edges only)
//     w0[i] = tgt | cls << 16          edge i of the SOURCE order (by (src, local edge id)): local target, edge class
//     w1[i] = eid | src << 16          eid = local id of source-order edge i;  src = local source of TARGET-order edge i
//     w2[i] = position, in the local source order, of TARGET-order edge i
// = 12 bytes per edge + 8 per node slot on the wire (the int64 edge lists were 24 bytes per edge).  A batch is the plain
// concatenation of its samples' blobs; one workgroup per sample adds the offsets batch_graph would add (node rows g*n, edge ids,
// positions) and inserts the self loops (last in both of a node's segments: their edge id E + v is the largest).  The result is
// bit-identical to qagnn_graph_prep_blocked on the same batch (tests/test_hip_kernels.py).
__device__ __forceinline__ int seg_owner(const int* __restrict__ rp, int n, int i) {  // largest v with rp[v] <= i
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (rp[mid] <= i) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(256) void k_blob_assemble(const int32_t* __restrict__ blobs, const int32_t* __restrict__ blob_off,
                                                       const int32_t* __restrict__ edge_off, const int64_t* __restrict__ node_type,
                                                       int n, int B, int R, int T, int* __restrict__ rowptr_s,
                                                       int* __restrict__ tgt_s, int* __restrict__ src_s, int* __restrict__ cls_s,
                                                       int* __restrict__ eid_s, int* __restrict__ rowptr_t, int* __restrict__ src_t,
                                                       int* __restrict__ tgt_t, int* __restrict__ cls_t, int* __restrict__ pos_t,
                                                       int* __restrict__ err) {
  extern __shared__ int rp[];  // rp_s[n+1] | rp_t[n+1]: local exclusive scans of the degrees
  __shared__ int wsum[2][4];
  int* const rp_s = rp;
  int* const rp_t = rp + n + 1;
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int E = edge_off[B];  // the batch's edge count, read on the device: the host only fixes the CAPACITY of the arrays (hipGraph replay)
  const int32_t* blob = blobs + blob_off[g];
  const int Eoff = edge_off[g], Eg = edge_off[g + 1] - Eoff, node0 = g * n, Epoff = Eoff + node0;
  const int32_t *cnt_s = blob, *cnt_t = blob + n;
  const uint32_t* w0 = reinterpret_cast<const uint32_t*>(blob + 2 * n);
  const uint32_t* w1 = w0 + Eg;
  const int32_t* w2 = reinterpret_cast<const int32_t*>(w1 + Eg);
  // exclusive scans: a thread owns `per` consecutive slots
  const int per = (n + 255) >> 8, v0 = tid * per;
  int sum_s = 0, sum_t = 0;
  for (int k = 0; k < per; ++k)
    if (v0 + k < n) { sum_s += cnt_s[v0 + k]; sum_t += cnt_t[v0 + k]; }
  int inc_s = sum_s, inc_t = sum_t;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int us = __shfl_up(inc_s, o, 64), ut = __shfl_up(inc_t, o, 64);
    if (lane >= o) { inc_s += us; inc_t += ut; }
  }
  if (lane == 63) { wsum[0][wid] = inc_s; wsum[1][wid] = inc_t; }
  __syncthreads();
  int run_s = inc_s - sum_s, run_t = inc_t - sum_t;
  for (int w = 0; w < wid; ++w) { run_s += wsum[0][w]; run_t += wsum[1][w]; }
  for (int k = 0; k < per; ++k)
    if (v0 + k < n) {
      rp_s[v0 + k] = run_s; rp_t[v0 + k] = run_t;
      run_s += cnt_s[v0 + k]; run_t += cnt_t[v0 + k];
    }
  if (tid == 255) { rp_s[n] = run_s; rp_t[n] = run_t; }
  __syncthreads();
  bool bad = rp_s[n] != Eg || rp_t[n] != Eg;
  // node rows: segment starts and the self loops
  for (int v = tid; v < n; v += 256) {
    rowptr_s[node0 + v] = Epoff + rp_s[v] + v;
    rowptr_t[node0 + v] = Epoff + rp_t[v] + v;
    const int64_t ty = node_type[node0 + v];
    bad = bad || ty < 0 || ty >= T;
    const int c = R * T * T + (int)min(max(ty, (int64_t)0), (int64_t)T - 1);
    const int ps = Epoff + rp_s[v + 1] + v, pt = Epoff + rp_t[v + 1] + v;
    tgt_s[ps] = node0 + v; src_s[ps] = node0 + v; cls_s[ps] = c; eid_s[ps] = E + node0 + v;
    src_t[pt] = node0 + v; tgt_t[pt] = node0 + v; cls_t[pt] = c; pos_t[pt] = ps;
  }
  if (g == B - 1 && tid == 0) { rowptr_s[(int64_t)B * n] = E + B * n; rowptr_t[(int64_t)B * n] = E + B * n; }
  const int C_real = R * T * T;
  for (int i = tid; i < Eg; i += 256) {
    const uint32_t a = w0[i], b = w1[i];
    const int pos = w2[i];
    const int tl = (int)(a & 0xFFFFu), cl = (int)(a >> 16), el = (int)(b & 0xFFFFu), sl = (int)(b >> 16);
    bad = bad || tl >= n || cl >= C_real || el >= Eg || sl >= n || pos < 0 || pos >= Eg;
    const int s = seg_owner(rp_s, n, i), t = seg_owner(rp_t, n, i);
    const int p = Epoff + i + s;
    tgt_s[p] = node0 + min(tl, n - 1); src_s[p] = node0 + s; cls_s[p] = min(cl, C_real - 1); eid_s[p] = Eoff + el;
    const int q = Epoff + i + t, posc = min(max(pos, 0), max(Eg - 1, 0)), slc = min(sl, n - 1);
    src_t[q] = node0 + slc; tgt_t[q] = node0 + t; pos_t[q] = Epoff + posc + slc;
    const int ct = min((int)(w0[posc] >> 16), C_real - 1);
    cls_t[q] = ct;
  }
  if (bad) *err = 1;
}

// ---- stable counting sort of the source-ordered positions by class -------------------------------------------
#define CLS_BLK 1024
// (Ep is read from the device -- rowptr_s[N], written by the kernels in front: the launch shapes below are sized by the arrays'
// CAPACITY, so that one captured launch sequence serves every batch of a capacity bucket)
__global__ __launch_bounds__(256) void k_cls_hist(const int* __restrict__ cls_s, int* __restrict__ hist,
                                                  int* __restrict__ cls_count, const int* __restrict__ Ep_dev, int C) {
  extern __shared__ int lh[];
  const int Ep = *Ep_dev;
  for (int c = threadIdx.x; c < C; c += 256) lh[c] = 0;
  __syncthreads();
  const int base = blockIdx.x * CLS_BLK;
  for (int i = threadIdx.x; i < CLS_BLK; i += 256)
    if (base + i < Ep) atomicAdd(&lh[cls_s[base + i]], 1);
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    hist[(int64_t)blockIdx.x * C + c] = lh[c];
    if (lh[c]) atomicAdd(&cls_count[c], lh[c]);
  }
}

// The class order is (position group, class)-major: a group is `gb` consecutive 1024-position blocks of the source order
// (<= QAGNN_CLS_GROUPS groups per batch), i.e. the edges of a few neighbouring subgraphs.  A chunk of the class pass then
// only touches node rows of its group (~1.6 MB of Q and G rows at the CSQA batch), which one XCD's L2 keeps; with a purely
// class-major order every chunk of a small class gathered rows from all over the batch and the pass ran at the fabric's
// random-row rate (profiles/r1_run59_gather_micro.txt: 6.4 TB/s batch-wide vs 18 TB/s subgraph-local).
// counts per (group, class): one thread per pair, class fastest (coalesced over the block histograms)
__global__ __launch_bounds__(256) void k_grp_count(const int* __restrict__ hist, int* __restrict__ gc_cnt, int nblk, int C, int gb, int NG) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NG * C) return;
  const int g = i / C, c = i - g * C;
  int sum = 0;
  for (int b = g * gb; b < min(nblk, (g + 1) * gb); ++b) sum += hist[(int64_t)b * C + c];
  gc_cnt[i] = sum;
}
// hist[b][c] <- first class-order slot of block b's edges of class c (inside the (group, class) range that starts at gcptr)
__global__ __launch_bounds__(256) void k_grp_base(int* __restrict__ hist, const int* __restrict__ gcptr, int nblk, int C, int gb, int NG) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= NG * C) return;
  const int g = i / C, c = i - g * C;
  int run = gcptr[i];
  for (int b = g * gb; b < min(nblk, (g + 1) * gb); ++b) {
    const int v = hist[(int64_t)b * C + c];
    hist[(int64_t)b * C + c] = run;
    run += v;
  }
}

// Stable scatter of one CLS_BLK-position block into the class order, ONE wave per block: the block is walked in 16 chunks of
// 64 positions; inside a chunk a lane's rank among the lanes of its class comes from ballots (one round per distinct class
// of the chunk, no memory traffic), the running per-class slot counters of the block live in LDS and are read once and
// written once per chunk.  (The first version ranked every position against all earlier positions of the block: 512 LDS
// reads per position on average, 80 us per batch.)
__global__ __launch_bounds__(64) void k_cls_scatter(const int* __restrict__ cls_s, const int* __restrict__ src_s,
                                                    const int* __restrict__ tgt_s, const int* __restrict__ hist,
                                                    int* __restrict__ src_c, int* __restrict__ tgt_c, int* __restrict__ pos_c,
                                                    const int* __restrict__ Ep_dev, int C) {
  extern __shared__ int cnt[];  // [C] next class-order slot of class c for this block
  const int Ep = *Ep_dev;
  const int lane = threadIdx.x, base = blockIdx.x * CLS_BLK;
  for (int c = lane; c < C; c += 64) cnt[c] = hist[(int64_t)blockIdx.x * C + c];
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));  // lanes below this one
  for (int q = 0; q < CLS_BLK / 64; ++q) {
    const int p = base + q * 64 + lane;
    const bool live = p < Ep;
    if (__ballot(live) == 0) break;
    const int c = live ? cls_s[p] : -1;
    const int first = live ? cnt[c] : 0;  // slot of the chunk's first position of class c
    int rank = 0, total = 0;
    unsigned long long todo = __ballot(live);
    while (todo) {
      const int cl = __builtin_amdgcn_readlane(c, __builtin_ctzll(todo));
      const unsigned long long m = __ballot(c == cl);
      if (c == cl) {
        rank = __builtin_popcountll(m & lt);
        total = __builtin_popcountll(m);
      }
      todo &= ~m;
    }
    if (live) {
      if (rank == 0) cnt[c] = first + total;  // exactly one lane per class of the chunk
      const int dst = first + rank;
      pos_c[dst] = p;
      src_c[dst] = src_s[p];
      tgt_c[dst] = tgt_s[p];
    }
  }
}

// chunk table: pair i = g * C + c owns chunks [chunkptr[i], chunkptr[i+1]), each <= QAGNN_CLS_CHUNK consecutive class-order slots
__global__ __launch_bounds__(1024) void k_chunk_counts(const int* __restrict__ gc_cnt, int* __restrict__ nch, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) nch[i] = (gc_cnt[i] + QAGNN_CLS_CHUNK - 1) / QAGNN_CLS_CHUNK;
}
// block 0: gcptr = exclusive scan of the pair counts; block 1: chunkptr = exclusive scan of the pairs' chunk counts
__global__ __launch_bounds__(1024) void k_scan_pairs(const int* gc_cnt, int* gcptr, const int* nch, int* chunkptr, int n) {
  if (blockIdx.x == 0) block_exclusive_scan(gc_cnt, gcptr, n);
  else block_exclusive_scan(nch, chunkptr, n);
}
__global__ void k_chunk_fill(const int* __restrict__ gcptr, const int* __restrict__ chunkptr, int* __restrict__ chunk_cls,
                             int* __restrict__ chunk_beg, int* __restrict__ chunk_len, int* __restrict__ n_chunks, int C, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) *n_chunks = chunkptr[n];
  if (i >= n) return;
  const int b = gcptr[i], e = gcptr[i + 1];
  int k = chunkptr[i];
  for (int p = b; p < e; p += QAGNN_CLS_CHUNK, ++k) {
    chunk_cls[k] = i % C;
    chunk_beg[k] = p;
    chunk_len[k] = min(QAGNN_CLS_CHUNK, e - p);
  }
}

// position groups of the class order: gb blocks of CLS_BLK positions each, at most QAGNN_CLS_GROUPS groups
static inline void cls_groups(int Ep, int* nblk, int* gb, int* NG) {
  *nblk = cdiv(Ep, CLS_BLK);
  *gb = cdiv(*nblk, QAGNN_CLS_GROUPS) > 1 ? cdiv(*nblk, QAGNN_CLS_GROUPS) : 1;
  *NG = cdiv(*nblk, *gb);
}

static inline int64_t up4(int64_t x) { return (x + 3) & ~(int64_t)3; }

// The node-side edge kernels walk 4 nodes per block with block b on XCD b % 8 (common.h: xcd_remap gives every XCD one contiguous
// run of blocks, so that a subgraph's rows stay in one L2).  Equal RUNS are not equal WORK: with 400..2000 edges per subgraph the eight
// runs of a 320-subgraph batch differ by +-9 % in edges and every edge kernel lasts as long as its heaviest XCD (measured by re-ordering
// the questions of the bench batch into equal-edge eighths: forward edge stage -5.6 %, backward -3.7 %).  base[0..8]: run k =
// blocks [base[k], base[k + 1]) holds 1/8 of the work, no run longer than `cap` blocks (the grids are 8 x cap).  Work of a node = its
// out-edges + 1/4 for the node itself (a row whose only edge is its self loop takes the degree-1 fast paths): in units of 1/4,
// 4 rowptr_s[i] - 3 i up to node i -- the prefix sum is there already, one binary search per boundary.  On the bench batch the
// heaviest run carries 1.078 x the mean with equal runs and 1.002 x with these.  balance = 0: equal runs.
__global__ void k_xcd_partition(const int* __restrict__ rowptr_s, int N, int cap, int balance, int* __restrict__ base) {
  const int nbk = (N + 3) >> 2, lane = threadIdx.x;
  int cand = 0;
  if (lane >= 1 && lane <= 7) {
    if (balance) {
      const long long W = 4ll * rowptr_s[N] - 3ll * N, target = (W * lane + 7) / 8;
      int lo = 0, hi = nbk;  // smallest lb with work(lb) >= target
      while (lo < hi) {
        const int mid = (lo + hi) >> 1, nd = min(4 * mid, N);
        if (4ll * rowptr_s[nd] - 3ll * nd >= target) hi = mid; else lo = mid + 1;
      }
      cand = lo;
    } else {
      const int q = nbk >> 3, r = nbk & 7;
      cand = lane < r ? lane * (q + 1) : r * (q + 1) + (lane - r) * q;
    }
  }
  __shared__ int c[8];
  if (lane < 8) c[lane] = cand;
  __syncthreads();
  if (lane == 0) {
    int prev = 0;
    base[0] = 0;
    for (int k = 1; k < 8; ++k) {
      int b = c[k];
      b = min(b, prev + cap);            // no run longer than the grid provides for
      b = max(b, nbk - (8 - k) * cap);   // ... and the runs behind this boundary can still cover the rest
      b = max(b, prev);
      base[k] = b;
      prev = b;
    }
    base[8] = nbk;
  }
}

// Zero `bytes` (a multiple of 16, 16-byte aligned) with a kernel of this library, NOT hipMemsetAsync.  Measured (round 5, visits 27-28,
// profiles/r5_run28_race_variants.txt, r5_run31_memset_node_fault.txt): ROCm 7.2 replays a captured hipMemsetAsync node of a LINEAR hipGraph (no forked stream:
// DEBUG_CLR_GRAPH_PACKET_CAPTURE, on by default) with its 16-byte fill pattern read from a kernel-argument slot that later eager launches
// of the same process recycle -- the replay then "zeroes" the range with the head of somebody's argument block.  A kernel has no such
// side buffer.
__global__ void k_zero16(int4* __restrict__ p, int64_t n16) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) p[i] = make_int4(0, 0, 0, 0);
}

}  // namespace qagnn

using namespace qagnn;

static int xcd_partition(qagnn_graph* g, hipStream_t stream) {
  // (work-balanced runs; the equal-node-count partition it replaced: profiles/r4_run15_edge_counters.txt, -5.6 % / -3.7 % on the edge stages)
  k_xcd_partition<<<1, 64, 0, stream>>>(g->rowptr_s, g->N, edge_xcd_cap(g->N), 1, g->err + 4);
  QAGNN_LAUNCH_CHECK("k_xcd_partition");
  return QAGNN_OK;
}

static hipError_t zero_range(void* p, size_t bytes, hipStream_t stream) {
#ifdef QAGNN_PREP_MEMSET_NODE  // the faulty form, only to reproduce the fault (tools/build_micro.sh -> scripts/r5_memset_node_fault.sh)
  return hipMemsetAsync(p, 0, bytes, stream);
#endif
  // (the callers' layout guarantees both: carve() rounds every array to 4 words and the storage is checked for 16-byte alignment; a layout
  // change that broke either would leave a tail of stale counters behind, so it fails here instead)
  if (bytes % 16 != 0 || !aligned16(p)) return hipErrorInvalidValue;
  const int64_t n16 = (int64_t)(bytes / 16);
  if (n16 == 0) return hipSuccess;  // (nothing launched: nothing to ask hipGetLastError about -- it would report and clear somebody else's error)
  int blocks = (int)((n16 + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  k_zero16<<<blocks, 256, 0, stream>>>((int4*)p, n16);
  return hipGetLastError();
}

// carve `storage` into the arrays of *g plus scratch (layout shared by qagnn_graph_prep_blocked and qagnn_graph_from_blobs)
struct carved {
  int32_t *eid_t, *gc_cnt, *gcptr, *nch, *cnt_s, *cnt_t, *es, *et, *ec, *tmp_s, *tmp_t, *srcpos, *hist;
  int nblk, gb, NG, pairs;
};
static carved carve(qagnn_graph* g, int32_t* storage, int N, int E, int R, int T, int block_n) {
  carved cv;
  const int Ep = E + N, C = R * T * T + T;
  cls_groups(Ep, &cv.nblk, &cv.gb, &cv.NG);
  cv.pairs = cv.NG * C;
  const int maxch = Ep / QAGNN_CLS_CHUNK + cv.pairs + 1;
  int32_t* p = storage;
  auto take = [&](int64_t n) { int32_t* r = p; p += up4(n); return r; };
  g->N = N; g->E = E; g->Ep = Ep; g->R = R; g->T = T; g->C = C; g->max_chunks = maxch; g->block_n = block_n;
  g->n_groups = cv.NG;
  g->rowptr_s = take(N + 1); g->rowptr_t = take(N + 1);
  g->tgt_s = take(Ep); g->src_s = take(Ep); g->cls_s = take(Ep); g->eid_s = take(Ep);
  g->src_t = take(Ep); g->tgt_t = take(Ep); g->cls_t = take(Ep); g->pos_t = take(Ep);
  g->src_c = take(Ep); g->tgt_c = take(Ep); g->pos_c = take(Ep);
  cv.eid_t = take(Ep);
  cv.gc_cnt = take(cv.pairs + 1); cv.gcptr = take(cv.pairs + 1);
  g->chunkptr = take(cv.pairs + 1);
  cv.nch = take(cv.pairs + 1);
  g->cls_count = take(C);
  g->chunk_cls = take(maxch); g->chunk_beg = take(maxch); g->chunk_len = take(maxch);
  g->n_chunks = take(4); g->err = take(16);  // err[4 .. 12]: the XCD partition of the node blocks (k_xcd_partition)
  // ---- scratch; the zero-initialised region comes first (cls_count, which sits just before it, must be zero too) ----
  cv.cnt_s = take(N); cv.cnt_t = take(N);
  cv.es = take(Ep); cv.et = take(Ep); cv.ec = take(Ep);
  cv.tmp_s = take(Ep); cv.tmp_t = take(Ep); cv.srcpos = take(Ep);
  cv.hist = take((int64_t)cv.nblk * C);
  return cv;
}

static int class_pass(qagnn_graph* g, int32_t* hist, int32_t* gc_cnt, int32_t* gcptr, int32_t* nch, int nblk, int gb, int NG, int pairs,
                      hipStream_t stream) {
  const int TB = 256, C = g->C;
  const int* Ep = g->rowptr_s + g->N;  // device: the true E' (g->Ep is the capacity the arrays and the grids are sized for)
  k_cls_hist<<<nblk, 256, C * sizeof(int), stream>>>(g->cls_s, hist, g->cls_count, Ep, C);
  QAGNN_LAUNCH_CHECK("k_cls_hist");
  k_grp_count<<<cdiv(pairs, TB), TB, 0, stream>>>(hist, gc_cnt, nblk, C, gb, NG);
  QAGNN_LAUNCH_CHECK("k_grp_count");
  k_chunk_counts<<<cdiv(pairs, 1024), 1024, 0, stream>>>(gc_cnt, nch, pairs);
  QAGNN_LAUNCH_CHECK("k_chunk_counts");
  k_scan_pairs<<<2, 1024, 0, stream>>>(gc_cnt, gcptr, nch, g->chunkptr, pairs);  // first class-order slot / first chunk of every pair
  QAGNN_LAUNCH_CHECK("k_scan_pairs");
  k_grp_base<<<cdiv(pairs, TB), TB, 0, stream>>>(hist, gcptr, nblk, C, gb, NG);
  QAGNN_LAUNCH_CHECK("k_grp_base");
  k_cls_scatter<<<nblk, 64, C * sizeof(int), stream>>>(g->cls_s, g->src_s, g->tgt_s, hist, g->src_c, g->tgt_c, g->pos_c, Ep, C);
  QAGNN_LAUNCH_CHECK("k_cls_scatter");
  k_chunk_fill<<<cdiv(pairs, 256), 256, 0, stream>>>(gcptr, g->chunkptr, g->chunk_cls, g->chunk_beg, g->chunk_len, g->n_chunks, C, pairs);
  QAGNN_LAUNCH_CHECK("k_chunk_fill");
  return QAGNN_OK;
}

extern "C" const char* qagnn_last_error(void) { return g_err; }
extern "C" int qagnn_abi_version(void) { return 21; }

extern "C" int64_t qagnn_graph_storage_elems(int32_t N, int32_t E, int32_t R, int32_t T) {
  const int64_t Ep = (int64_t)E + N, C = (int64_t)R * T * T + T;
  int nblk_i, gb, NG;
  cls_groups((int)Ep, &nblk_i, &gb, &NG);
  const int64_t pairs = (int64_t)NG * C;
  const int64_t maxch = Ep / QAGNN_CLS_CHUNK + pairs + 1;
  const int64_t nblk = nblk_i;
  int64_t tot = 0;
  tot += 2 * up4(N + 1);      // rowptr_s, rowptr_t
  tot += 12 * up4(Ep);        // tgt_s src_s cls_s eid_s src_t tgt_t cls_t pos_t src_c tgt_c pos_c + eid_t
  tot += 4 * up4(pairs + 1);  // gc_cnt, gcptr, chunkptr, nch scratch
  tot += up4(C);              // cls_count
  tot += 3 * up4(maxch);      // chunk tables
  tot += up4(4) + up4(16);    // n_chunks, err (+ the XCD partition)
  tot += 2 * up4(N);          // cnt_s, cnt_t
  tot += 6 * up4(Ep);         // es et ec tmp_s tmp_t srcpos
  tot += up4(nblk * C);       // per-block class histograms
  return tot;
}

extern "C" int qagnn_graph_prep(qagnn_graph* g, int32_t* storage, const int64_t* edge_index, const int64_t* edge_type,
                                const int64_t* node_type, int32_t N, int32_t E, int32_t R, int32_t T, qagnn_stream_t stream_) {
  return qagnn_graph_prep_blocked(g, storage, edge_index, edge_type, node_type, N, E, R, T, 0, stream_);
}

extern "C" int qagnn_graph_prep_blocked(qagnn_graph* g, int32_t* storage, const int64_t* edge_index, const int64_t* edge_type,
                                        const int64_t* node_type, int32_t N, int32_t E, int32_t R, int32_t T, int32_t block_n,
                                        qagnn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  QAGNN_REQUIRE(block_n >= 0 && (block_n == 0 || N % block_n == 0), QAGNN_EINVAL, "graph_prep: N=%d is not a multiple of block_n=%d", N, block_n);
  QAGNN_REQUIRE(g && storage && node_type, QAGNN_EINVAL, "graph_prep: null pointer");
  QAGNN_REQUIRE(N > 0 && E >= 0 && R > 0 && T > 0, QAGNN_EINVAL, "graph_prep: bad sizes N=%d E=%d R=%d T=%d", N, E, R, T);
  QAGNN_REQUIRE(E == 0 || (edge_index && edge_type), QAGNN_EINVAL, "graph_prep: null edge arrays with E=%d", E);
  QAGNN_REQUIRE(aligned16(storage), QAGNN_EINVAL, "graph_prep: storage must be 16-byte aligned");
  const int64_t Ep64 = (int64_t)E + N, C64 = (int64_t)R * T * T + T;
  QAGNN_REQUIRE(Ep64 < (1ll << 30), QAGNN_EUNSUPPORTED, "graph_prep: E+N=%lld too large", (long long)Ep64);
  QAGNN_REQUIRE(C64 <= 8192, QAGNN_EUNSUPPORTED, "graph_prep: %lld edge classes > 8192", (long long)C64);
  const int Ep = (int)Ep64;
  carved cv = carve(g, storage, N, E, R, T, block_n);
  int32_t *cnt_s = cv.cnt_s, *cnt_t = cv.cnt_t, *es = cv.es, *et = cv.et, *ec = cv.ec, *tmp_s = cv.tmp_s, *tmp_t = cv.tmp_t;
  int32_t *srcpos = cv.srcpos, *eid_t = cv.eid_t, *hist = cv.hist, *gc_cnt = cv.gc_cnt, *gcptr = cv.gcptr, *nch = cv.nch;
  const int nblk = cv.nblk, gb = cv.gb, NG = cv.NG, pairs = cv.pairs;

  hipError_t he = zero_range(g->cls_count, (size_t)((char*)es - (char*)g->cls_count), stream);
  if (he != hipSuccess) { set_error("graph_prep: k_zero16 failed: %s", hipGetErrorString(he)); return QAGNN_EHIP; }
  const int TB = 256;
  k_decode_count<<<cdiv(Ep, TB), TB, 0, stream>>>(edge_index, edge_type, node_type, N, E, R, T, es, et, ec, cnt_s, cnt_t,
                                                   g->err, block_n);
  QAGNN_LAUNCH_CHECK("k_decode_count");
  k_scan3<<<2, 1024, 0, stream>>>(cnt_s, g->rowptr_s, N, cnt_t, g->rowptr_t, N, nullptr, nullptr, 0);
  QAGNN_LAUNCH_CHECK("k_scan3");
  k_fill<<<cdiv(Ep, TB), TB, 0, stream>>>(es, et, g->rowptr_s, g->rowptr_t, cnt_s, cnt_t, tmp_s, tmp_t, Ep);
  QAGNN_LAUNCH_CHECK("k_fill");
  k_sort_segments<<<cdiv(2 * (int64_t)N, 4), 256, 0, stream>>>(g->rowptr_s, g->rowptr_t, tmp_s, tmp_t, g->eid_s, eid_t, srcpos, N);
  QAGNN_LAUNCH_CHECK("k_sort_segments");
  k_payload<<<cdiv(Ep, TB), TB, 0, stream>>>(es, et, ec, g->eid_s, eid_t, srcpos, g->tgt_s, g->src_s, g->cls_s, g->src_t,
                                              g->tgt_t, g->cls_t, g->pos_t, Ep);
  QAGNN_LAUNCH_CHECK("k_payload");
  int rc = xcd_partition(g, stream);
  if (rc != QAGNN_OK) return rc;
  return class_pass(g, hist, gc_cnt, gcptr, nch, nblk, gb, NG, pairs, stream);
}

extern "C" int qagnn_graph_from_blobs(qagnn_graph* g, int32_t* storage, const int32_t* blobs, const int32_t* blob_off,
                                      const int32_t* edge_off, const int64_t* node_type, int32_t B, int32_t n, int32_t E, int32_t R,
                                      int32_t T, qagnn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  QAGNN_REQUIRE(g && storage && blobs && blob_off && edge_off && node_type, QAGNN_EINVAL, "graph_from_blobs: null pointer");
  QAGNN_REQUIRE(B > 0 && n > 0 && E >= 0 && R > 0 && T > 0, QAGNN_EINVAL, "graph_from_blobs: bad sizes B=%d n=%d E=%d R=%d T=%d", B, n, E, R, T);
  QAGNN_REQUIRE(n < 65536 && (int64_t)R * T * T < 65536, QAGNN_EUNSUPPORTED, "graph_from_blobs: n=%d or R*T*T=%d does not fit the 16-bit blob fields", n, R * T * T);
  QAGNN_REQUIRE((size_t)(2 * (n + 1)) * sizeof(int) <= 64 * 1024, QAGNN_EUNSUPPORTED, "graph_from_blobs: n=%d node slots per sample exceed the LDS scan", n);
  QAGNN_REQUIRE(aligned16(storage), QAGNN_EINVAL, "graph_from_blobs: storage must be 16-byte aligned");
  const int64_t N64 = (int64_t)B * n, Ep64 = (int64_t)E + N64, C64 = (int64_t)R * T * T + T;
  QAGNN_REQUIRE(Ep64 < (1ll << 30), QAGNN_EUNSUPPORTED, "graph_from_blobs: E+N=%lld too large", (long long)Ep64);
  QAGNN_REQUIRE(C64 <= 8192, QAGNN_EUNSUPPORTED, "graph_from_blobs: %lld edge classes > 8192", (long long)C64);
  carved cv = carve(g, storage, (int)N64, E, R, T, n);
  hipError_t he = zero_range(g->cls_count, (size_t)((char*)cv.es - (char*)g->cls_count), stream);
  if (he != hipSuccess) { set_error("graph_from_blobs: k_zero16 failed: %s", hipGetErrorString(he)); return QAGNN_EHIP; }
  k_blob_assemble<<<B, 256, (size_t)(2 * (n + 1)) * sizeof(int), stream>>>(blobs, blob_off, edge_off, node_type, n, B, R, T, g->rowptr_s,
                                                                           g->tgt_s, g->src_s, g->cls_s, g->eid_s, g->rowptr_t, g->src_t, g->tgt_t,
                                                                           g->cls_t, g->pos_t, g->err);
  QAGNN_LAUNCH_CHECK("k_blob_assemble");
  int rc = xcd_partition(g, stream);
  if (rc != QAGNN_OK) return rc;
  return class_pass(g, cv.hist, cv.gc_cnt, cv.gcptr, cv.nch, cv.nblk, cv.gb, cv.NG, cv.pairs, stream);
}
