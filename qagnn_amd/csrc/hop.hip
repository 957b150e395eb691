// One GATConvE hop of the stack as ONE C call each way: the host-side sequencing of the kernels of this library.
//
// Reference: QAGNN_Message_Passing.mp_helper (modeling/modeling_qagnn.py:45-50) calling GATConvE.forward (:411-452) -> message
// (:455-484) -> mlp (:443, def :408) -> GELU -> dropout, and the autograd backward of all of it.  qagnn_amd/ops.py composes the
// same launches from Python (three autograd operators per hop, ~45 C-ABI calls and ~60 tensor allocations per layer and step);
// at the reference's own mini-batch (2 questions x 5 choices) that host work, not the GPU, bounds the step.  Here the sequence
// is native: 2 calls and a handful of allocations per layer.  The launches, their order and their arguments are exactly those
// of the composed path, so the results are bit-identical (tests/test_hip_kernels.py::test_fused_hop_equals_composed_path).
//
// No allocation, no synchronisation: the caller hands in every buffer, including one scratch region whose size comes from
// qagnn_hop_{fwd,bwd}_workspace_elems().
#include "common.h"

#include <mutex>

namespace qagnn {

static inline int64_t up4(int64_t x) { return (x + 3) & ~(int64_t)3; }
static inline int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }

struct Carver {
  float* p;
  float* end;
  float* take(int64_t n) {
    float* r = p;
    p += up4(n);
    return r;
  }
  bool ok() const { return p <= end; }
};

static int check_hop(const qagnn_hop_args* h, const char* who) {
  QAGNN_REQUIRE(h && h->g, QAGNN_EINVAL, "%s: null argument block / graph", who);
  QAGNN_REQUIRE(h->N == h->g->N && h->N > 0, QAGNN_EINVAL, "%s: N=%d does not match the graph (N=%d)", who, h->N, h->g->N);
  QAGNN_REQUIRE(h->HP > 0 && h->HP % 4 == 0 && h->DP == 4 * h->HP, QAGNN_EINVAL, "%s: DP=%d must be 4*HP (HP=%d)", who, h->DP, h->HP);
  QAGNN_REQUIRE(h->DP % 16 == 0 && h->SP >= 0 && h->SP % 16 == 0, QAGNN_EUNSUPPORTED,
                "%s: DP=%d and SP=%d must be multiples of 16 (GEMM k-tile)", who, h->DP, h->SP);
  QAGNN_REQUIRE(h->T >= 1 && h->T <= 4, QAGNN_EUNSUPPORTED, "%s: %d node types (the type-table gradient handles 1..4)", who, h->T);
  QAGNN_REQUIRE(h->X && h->ntype && h->Wx_t && h->Wx && h->TT && h->EkEm && h->W1t && h->W1 && h->b1 && h->gamma && h->beta &&
                    h->W2t && h->W2 && h->b2,
                QAGNN_EINVAL, "%s: null parameter pointer", who);
  QAGNN_REQUIRE(h->SP == 0 || (h->S && h->Ws_t && h->Ws), QAGNN_EINVAL, "%s: SP=%d but S / Ws_t / Ws missing", who, h->SP);
  QAGNN_REQUIRE(h->KMQ && h->a && h->alpha && h->aggr && h->h1 && h->out && h->stats, QAGNN_EINVAL, "%s: null saved-buffer pointer", who);
  QAGNN_REQUIRE(h->batch_stats || (h->run_mean_p && h->run_var_p), QAGNN_EINVAL, "%s: running statistics missing", who);
  QAGNN_REQUIRE(!h->apply_act || h->y, QAGNN_EINVAL, "%s: apply_act without an output buffer y", who);
  QAGNN_REQUIRE(h->ws, QAGNN_EINVAL, "%s: null workspace", who);
  return QAGNN_OK;
}

// ---- the three-MFMA GEMM form inside a hop (qagnn_hop_args.amax, gemm_split == 2; arithmetic: gemm_nn2.hip's header) -------------
// Word layout of a hop's amax block; X and S may live in another hop's block (a chained stack: the previous hop's y IS this hop's X)
enum { AM_X = 0, AM_S = 1, AM_AGGR = 2, AM_H1 = 3, AM_Y = 4, AM_DOUT = 5, AM_DH1 = 6, AM_DKMQ = 7 };
struct HopAmax {
  bool on;          // this hop runs the form
  uint32_t* x;      // max |X|   (written by the previous hop's GELU / dropout, or by a reduction pass at the start of the call)
  uint32_t* s;      // max |S|
  uint32_t* y;      // where this hop's GELU / dropout leaves max |y|
  uint32_t* own;    // the hop's own block
};
// The form needs batch statistics (the bound of relu(bn(h1)) comes from them) and pays from NN2_PACK_MIN_M rows on (packed B images)
static bool hop_h2(const qagnn_hop_args* h) { return h->gemm_split >= 2 && h->amax != nullptr && h->batch_stats && h->N >= 8192; }
// gemm_split == 3: the reduced-precision form (one fp16 MFMA per product) wherever 2 would take three
static int hop_pieces(const qagnn_hop_args* h) { return h->gemm_split == 3 ? 1 : 0; }
typedef int (*tn_scaled_fn)(const float*, int32_t, int32_t, const float*, int32_t, int32_t, const float*, int32_t, float*, int32_t, int32_t, int32_t,
                            const float*, const float*, const uint32_t*, const uint32_t*, const uint32_t*, float*, qagnn_stream_t);
static tn_scaled_fn hop_tn(const qagnn_hop_args* h) { return h->gemm_split == 3 ? qagnn_gemm_tn_h1_f32 : qagnn_gemm_tn_h2_f32; }
static HopAmax hop_amax_single(const qagnn_hop_args* h) {
  HopAmax m{hop_h2(h), nullptr, nullptr, nullptr, h->amax};
  if (m.on) {  // (x_amax / s_amax: words the caller's producers filled; read-only here)
    m.x = h->x_amax ? const_cast<uint32_t*>(h->x_amax) : h->amax + AM_X;
    m.s = h->s_amax ? const_cast<uint32_t*>(h->s_amax) : h->amax + AM_S;
    m.y = h->amax + AM_Y;
  }
  return m;
}
// hop l of a stack: X / S words shared along the chain where the tensors are
static HopAmax hop_amax_stack(const qagnn_hop_args* hops, int k, int l) {
  HopAmax m = hop_amax_single(&hops[l]);
  if (!m.on) return m;
  if (l > 0 && hop_h2(&hops[l - 1]) && hops[l].X == hops[l - 1].y && hops[l - 1].apply_act && !hops[l].x_amax) m.x = hops[l - 1].amax + AM_Y;
  if (l > 0 && hop_h2(&hops[0]) && hops[l].S == hops[0].S && hops[l].SP == hops[0].SP && !hops[l].s_amax)
    m.s = hops[0].s_amax ? const_cast<uint32_t*>(hops[0].s_amax) : hops[0].amax + AM_S;
  return m;
}

}  // namespace qagnn

using namespace qagnn;

#define HOP_TRY(call)              \
  do {                             \
    int rc__ = (call);             \
    if (rc__ != QAGNN_OK) return rc__; \
  } while (0)

// scratch for the packed B image of a hop's NN products (qagnn_gemm_nn_split_ws_f32), in floats: sized for the projection
// [DP | <= DP] -> 3 DP, the largest of them; one buffer, the products of a hop are in order on the main stream
static int64_t hop_pack_elems(int DP) { return up4((qagnn_gemm_nn_pack_bytes(3 * DP, DP, DP) + 3) / 4); }

extern "C" int64_t qagnn_hop_fwd_workspace_elems(int32_t N, int32_t Ep, int32_t DP) {
  return up4((int64_t)Ep * 4) + up4(max64(qagnn_colreduce_workspace_elems(N, DP, 1), (int64_t)cdiv(N, 128) * 3 * DP)) + hop_pack_elems(DP) +
         up4(max64(N, gelu_amax_scratch_elems((int64_t)N * DP)));  // (+ per-node / per-block maxima: qagnn_hop_args.amax)
}

// NN product through the kernel family the caller asked for (qagnn_hop_args.gemm_split): the bf16-split kernel takes B in its
// [No][K] layout, which the hop holds for every weight (W and W^T both arrive packed)
static int hop_nn(const qagnn_hop_args* h, const qagnn_gemm_nn_args* a, const float* B1n, int ldn1, const float* B2n, int ldn2,
                  float* pkws, int64_t pk_elems, qagnn_stream_t stream) {
  if (h->gemm_split && a->K1 % 4 == 0 && a->K2 % 4 == 0) return qagnn_gemm_nn_split_ws_f32(a, B1n, ldn1, B2n, ldn2, pkws, pk_elems * 4, stream);
  return qagnn_gemm_nn_f32(a, stream);
}

// x_ready / s_ready: the words of X / S already hold this call's maxima (an earlier hop of the stack produced them)
static int hop_fwd_one(const qagnn_hop_args* h, const HopAmax& am, bool x_ready, bool s_ready, qagnn_stream_t stream) {
  HOP_TRY(check_hop(h, "hop_fwd"));
  const int N = h->N, DP = h->DP, SP = h->SP, Ep = h->g->Ep;
  Carver w{h->ws, h->ws + h->ws_elems};
  float* score = w.take((int64_t)Ep * 4);
  float* crws = w.take(max64(qagnn_colreduce_workspace_elems(N, DP, 1), (int64_t)cdiv(N, 128) * 3 * DP));
  const int64_t pk_elems = hop_pack_elems(DP);
  float* pkws = w.take(pk_elems);
  float* ampart = w.take(max64(N, gelu_amax_scratch_elems((int64_t)N * DP)));
  QAGNN_REQUIRE(w.ok(), QAGNN_EINVAL, "hop_fwd: workspace of %lld floats is too small", (long long)h->ws_elems);
  float* mean = h->stats, *var = h->stats + DP, *invstd = h->stats + 2 * DP, *scale = h->stats + 3 * DP, *shift = h->stats + 4 * DP;

  if (am.on) {  // operand maxima nobody left behind: one reduction pass each
    if (!x_ready) HOP_TRY(qagnn_absmax_f32(h->X, (int64_t)N * DP, am.x, stream));
    if (SP > 0 && !s_ready) HOP_TRY(qagnn_absmax_f32(h->S, (int64_t)N * SP, am.s, stream));
  }
  // K | M | Q = [X | S] [Wx ; Ws] + TT[node type]      (project-then-gather: linear_key/msg/query on N node rows, :464-466)
  qagnn_gemm_nn_args ga = {};
  ga.A1 = h->X; ga.lda1 = DP; ga.K1 = DP; ga.B1 = h->Wx_t; ga.ldb1 = 3 * DP;
  if (SP > 0) { ga.A2 = h->S; ga.lda2 = SP; ga.K2 = SP; ga.B2 = h->Ws_t; ga.ldb2 = 3 * DP; }
  ga.C = h->KMQ; ga.ldc = 3 * DP; ga.M = N; ga.No = 3 * DP;
  ga.rowtab = h->TT; ga.ldt = 3 * DP; ga.rowidx = h->ntype;
  if (am.on) { ga.a_amax1 = am.x; ga.a_amax2 = SP > 0 ? am.s : nullptr; ga.pieces = hop_pieces(h); }
  HOP_TRY(hop_nn(h, &ga, h->Wx, DP, SP > 0 ? h->Ws : nullptr, SP, pkws, pk_elems, stream));
  // attention + aggregation (:442, 455-484)
  HOP_TRY(launch_edge_attn_fwd(h->g, h->KMQ, 3 * DP, h->EkEm, 2 * DP, h->HP, h->qscale, score, h->a, h->alpha, h->aggr, DP,
                               am.on ? ampart : nullptr, (hipStream_t)stream));
  if (am.on) HOP_TRY(launch_amax_reduce(ampart, N, am.own + AM_AGGR, (hipStream_t)stream));
  // mlp: Linear -> BatchNorm1d -> ReLU -> Linear (:443, 408); BN + ReLU are folded into the second GEMM's operand load
  qagnn_gemm_nn_args g1 = {};
  g1.A1 = h->aggr; g1.lda1 = DP; g1.K1 = DP; g1.B1 = h->W1t; g1.ldb1 = DP; g1.C = h->h1; g1.ldc = DP; g1.M = N; g1.No = DP; g1.bias = h->b1;
  // batch statistics as a by-product of this GEMM's epilogue where the split kernel can provide them (same rule as ops.GatMlpFn)
  const bool fused_stats = h->batch_stats && h->gemm_split && DP > 192 && DP <= 208;
  if (fused_stats) g1.colstat_part = crws;
  if (am.on) { g1.a_amax1 = am.own + AM_AGGR; g1.pieces = hop_pieces(h); }
  HOP_TRY(hop_nn(h, &g1, h->W1, DP, nullptr, 0, pkws, pk_elems, stream));
  const double Rd = (double)N;
  const float unbias = (float)(Rd / (Rd - 1.0 > 1.0 ? Rd - 1.0 : 1.0));
  const bool h1_bound = am.on && fused_stats;  // (the bound of relu(bn(h1)) rides on the statistics launch)
  if (fused_stats) {
    QAGNN_REQUIRE(!h->run_mean || (h->run_var && h->d > 0 && h->d % 4 == 0 && h->d <= DP), QAGNN_EINVAL, "hop_fwd: running-stat arguments");
    HOP_TRY(launch_bn_stats_finalize(crws, cdiv(N, 128), N, DP, h->gamma, h->beta, h->eps, h->stats, h->run_mean, h->run_var, h->num_batches_tracked,
                                     h->d, h->momentum, unbias, h->ones_col, h1_bound ? am.own + AM_H1 : nullptr, (hipStream_t)stream));
  } else {
    const float* mean_u = mean;
    const float* var_u = var;
    if (h->batch_stats) {
      const float inv_rows = (float)(1.0 / (double)N);  // rounded like the composed path's Python double -> float
      HOP_TRY(qagnn_colreduce_f32(0, h->h1, DP, nullptr, DP, N, DP, nullptr, 1, nullptr, nullptr, nullptr, nullptr, nullptr, inv_rows, mean, crws,
                                  stream));
      HOP_TRY(qagnn_colreduce_f32(1, h->h1, DP, nullptr, DP, N, DP, nullptr, 1, mean, nullptr, nullptr, nullptr, nullptr, inv_rows, var, crws,
                                  stream));
    } else {
      mean_u = h->run_mean_p;
      var_u = h->run_var_p;
    }
    HOP_TRY(qagnn_bn_finalize_f32(mean_u, var_u, h->gamma, h->beta, h->eps, invstd, scale, shift, DP, h->run_mean, h->run_var,
                                  h->num_batches_tracked, h->dense_pos, h->d, h->momentum, unbias, h->ones_col, stream));
  }
  qagnn_gemm_nn_args g2 = {};
  g2.A1 = h->h1; g2.lda1 = DP; g2.K1 = DP; g2.B1 = h->W2t; g2.ldb1 = DP; g2.C = h->out; g2.ldc = DP; g2.M = N; g2.No = DP; g2.bias = h->b2;
  g2.a_scale = scale; g2.a_shift = shift;
  if (h1_bound) { g2.a_amax1 = am.own + AM_H1; g2.pieces = hop_pieces(h); }
  HOP_TRY(hop_nn(h, &g2, h->W2, DP, nullptr, 0, pkws, pk_elems, stream));
  if (h->apply_act) {  // X' = dropout(GELU(out))  (:48-49)
    QAGNN_REQUIRE(h->p_drop >= 0.f && h->p_drop < 1.f, QAGNN_EINVAL, "hop_fwd: p=%f", h->p_drop);
    HOP_TRY(launch_gelu_dropout(h->out, nullptr, h->y, (int64_t)N * DP, h->p_drop, h->seed, am.on ? am.y : nullptr, ampart, (hipStream_t)stream));
  }
  return QAGNN_OK;
}

extern "C" int qagnn_hop_fwd_f32(const qagnn_hop_args* h, qagnn_stream_t stream) {
  QAGNN_REQUIRE(h, QAGNN_EINVAL, "hop_fwd: null argument block");
  const HopAmax am = hop_amax_single(h);
  if (am.on) HOP_TRY(qagnn_zero_words(h->amax, QAGNN_HOP_AMAX_WORDS, stream));
  return hop_fwd_one(h, am, h->x_amax != nullptr, h->s_amax != nullptr, stream);
}

extern "C" int64_t qagnn_hop_bwd_workspace_elems(int32_t N, int32_t Ep, int32_t DP, int32_t SP, int32_t cls_part_rows) {
  int64_t tn = max64(qagnn_gemm_tn_workspace_elems(N, DP, DP), qagnn_gemm_tn_workspace_elems(N, DP, 3 * DP));
  if (SP > 0) tn = max64(tn, qagnn_gemm_tn_workspace_elems(N, DP + SP, 3 * DP));
  // two sets of the buffers the weight-gradient stream reads (d out, d h1, d K|M|Q) + what the main stream keeps to itself
  return 2 * (2 * up4((int64_t)N * DP) + up4((int64_t)N * 3 * DP)) + up4((int64_t)N * DP) + up4((int64_t)Ep * 4) + up4((int64_t)N * 4) +
         up4((int64_t)cls_part_rows * 2 * DP) + up4(tn) + up4(qagnn_colreduce_workspace_elems(N, 3 * DP, 4)) + hop_pack_elems(DP) +
         up4(max64((int64_t)3 * N, gelu_amax_scratch_elems((int64_t)N * DP)));  // (+ per-node / per-block maxima)
}

namespace qagnn {

// Events that order the main stream and the weight-gradient stream (qagnn_hop_args.side_stream): created once per device, reused
// round-robin (a wait refers to the record that preceded it, so an event may be recorded again once its wait is enqueued).
constexpr int HOP_MAX_DEV = 16, HOP_EV_POOL = 64;
static hipEvent_t g_hop_ev[HOP_MAX_DEV][HOP_EV_POOL];
static bool g_hop_ev_made[HOP_MAX_DEV];
static std::mutex g_hop_ev_mu;

struct SideSync {
  hipStream_t main, side;  // side == nullptr: everything on `main`, no events
  hipEvent_t* ev;
  int next;
  hipEvent_t take() { return ev[next++ % HOP_EV_POOL]; }
};

static int side_sync_init(SideSync* s, hipStream_t main, hipStream_t side) {
  s->main = main;
  s->side = (side && side != main) ? side : nullptr;
  s->ev = nullptr;
  s->next = 0;
  if (!s->side) return QAGNN_OK;
  int dev = 0;
  hipError_t he = hipGetDevice(&dev);
  QAGNN_REQUIRE(he == hipSuccess && dev >= 0 && dev < HOP_MAX_DEV, QAGNN_EHIP, "hop_bwd: device %d outside the event table", dev);
  std::lock_guard<std::mutex> lk(g_hop_ev_mu);
  if (!g_hop_ev_made[dev]) {
    for (int i = 0; i < HOP_EV_POOL; ++i) {
      he = hipEventCreateWithFlags(&g_hop_ev[dev][i], hipEventDisableTiming);
      QAGNN_REQUIRE(he == hipSuccess, QAGNN_EHIP, "hop_bwd: hipEventCreate failed: %s", hipGetErrorString(he));
    }
    g_hop_ev_made[dev] = true;
  }
  s->ev = g_hop_ev[dev];
  return QAGNN_OK;
}
// `to` continues after everything enqueued on `from` so far
static int stream_after(hipStream_t to, hipStream_t from, hipEvent_t ev) {
  hipError_t he = hipEventRecord(ev, from);
  if (he == hipSuccess) he = hipStreamWaitEvent(to, ev, 0);
  QAGNN_REQUIRE(he == hipSuccess, QAGNN_EHIP, "hop_bwd: stream fork / join failed: %s", hipGetErrorString(he));
  return QAGNN_OK;
}

// One hop's backward.  The four weight-gradient products (and the gradients read off their rows) feed nothing downstream, they are
// latency-bound split-K launches with tiny outputs, and the data-gradient chain next to them is a serial chain of launches: with a
// side stream they leave the chain (fork after each of their operands is complete, join at the end of the stack).  What they read
// from the workspace (d out, d h1, d K|M|Q) lives in buffer set `set`; the caller alternates sets from hop to hop and makes the
// main stream wait for `*done` before it reuses one, so the side stream may lag the main stream by a whole hop.
static int hop_bwd_one(const qagnn_hop_args* h, const HopAmax& am, SideSync* ss, int set, hipEvent_t* done) {
  HOP_TRY(check_hop(h, "hop_bwd"));
  QAGNN_REQUIRE(h->dy && h->dWx_t && h->dTT && h->dEkEm && h->dW1t && h->db1 && h->dbn && h->dW2t && h->db2, QAGNN_EINVAL,
                "hop_bwd: null gradient pointer");
  QAGNN_REQUIRE(h->SP == 0 || h->dWs_t, QAGNN_EINVAL, "hop_bwd: dWs_t missing");
  const int N = h->N, DP = h->DP, SP = h->SP, Ep = h->g->Ep;
  Carver w{h->ws, h->ws + h->ws_elems};
  float *bufA2[2], *bufC2[2], *dKMQ2[2];
  for (int i = 0; i < 2; ++i) {
    bufA2[i] = w.take((int64_t)N * DP);
    bufC2[i] = w.take((int64_t)N * DP);
    dKMQ2[i] = w.take((int64_t)N * 3 * DP);
  }
  float* bufA = bufA2[set];  // d out
  float* bufC = bufC2[set];  // d h1
  float* dKMQ = dKMQ2[set];
  float* bufB = w.take((int64_t)N * DP);  // d relu(bn(h1)), then d aggr (main stream only)
  float* gab = w.take((int64_t)Ep * 4);
  float* rs = w.take((int64_t)N * 4);
  float* cls_part = w.take(((int64_t)h->g->max_chunks + (int64_t)QAGNN_CLS_SLICES * h->g->C) * 2 * DP);
  int64_t tn = max64(qagnn_gemm_tn_workspace_elems(N, DP, DP), qagnn_gemm_tn_workspace_elems(N, DP, 3 * DP));
  if (SP > 0) tn = max64(tn, qagnn_gemm_tn_workspace_elems(N, DP + SP, 3 * DP));
  float* tnws = w.take(tn);  // split-K partials: used by the weight-gradient products only, which are in order on one stream
  float* crws = w.take(qagnn_colreduce_workspace_elems(N, 3 * DP, 4));  // column-reduction partials: main stream only
  const int64_t pk_elems = hop_pack_elems(DP);
  float* pkws = w.take(pk_elems);  // packed B images of the data-gradient products: main stream only
  float* ampart = w.take(max64((int64_t)3 * N, gelu_amax_scratch_elems((int64_t)N * DP)));
  QAGNN_REQUIRE(w.ok(), QAGNN_EINVAL, "hop_bwd: workspace of %lld floats is too small", (long long)h->ws_elems);
  const float* mean = h->batch_stats ? h->stats : h->run_mean_p;
  const float* invstd = h->stats + 2 * DP, *scale = h->stats + 3 * DP, *shift = h->stats + 4 * DP;
  const qagnn_stream_t stream = (qagnn_stream_t)ss->main;
  const qagnn_stream_t wstream = (qagnn_stream_t)(ss->side ? ss->side : ss->main);  // where the weight gradients go

  // GELU + dropout backward
  const float* dout = h->dy;
  // (the three-MFMA form in the backward needs every maximum the forward left: the same condition, and h1's bound)
  const bool h2 = am.on && DP > 192 && DP <= 208;
  uint32_t* const w_dout = h2 ? am.own + AM_DOUT : nullptr, *const w_dh1 = h2 ? am.own + AM_DH1 : nullptr, *const w_dkmq = h2 ? am.own + AM_DKMQ : nullptr;
  if (h->apply_act) {
    HOP_TRY(launch_gelu_dropout(h->out, h->dy, bufA, (int64_t)N * DP, h->p_drop, h->seed, w_dout, ampart, (hipStream_t)stream));
    dout = bufA;
  } else if (h2) {
    HOP_TRY(qagnn_absmax_f32(dout, (int64_t)N * DP, w_dout, stream));
  }
  if (h->ones_col < 0)
    HOP_TRY(qagnn_colreduce_f32(0, dout, DP, nullptr, DP, N, DP, nullptr, 1, nullptr, nullptr, nullptr, nullptr, nullptr, 1.0f, h->db2, crws, stream));
  // second Linear: dW2^T = relu(bn(h1))^T dout, d r = dout W2
  if (ss->side) HOP_TRY(stream_after(ss->side, ss->main, ss->take()));
  if (h2) HOP_TRY(hop_tn(h)(h->h1, DP, DP, nullptr, 0, 0, dout, DP, h->dW2t, DP, N, DP, scale, shift, am.own + AM_H1, nullptr, w_dout, tnws, wstream));
  else HOP_TRY(qagnn_gemm_tn_f32(h->h1, DP, dout, DP, h->dW2t, DP, N, DP, DP, scale, shift, nullptr, 0, tnws, wstream));
  // relu(bn(h1)) carries a column of ones there: that row of the weight gradient is the bias gradient (no copy when the caller's db2 IS
  // that row)
  if (h->ones_col >= 0 && h->db2 != h->dW2t + (int64_t)h->ones_col * DP) {
    hipError_t he = hipMemcpyAsync(h->db2, h->dW2t + (int64_t)h->ones_col * DP, (size_t)DP * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)wstream);
    if (he != hipSuccess) { set_error("hop_bwd: db2 copy failed: %s", hipGetErrorString(he)); return QAGNN_EHIP; }
  }
  qagnn_gemm_nn_args gr = {};
  gr.A1 = dout; gr.lda1 = DP; gr.K1 = DP; gr.B1 = h->W2; gr.ldb1 = DP; gr.C = bufB; gr.ldc = DP; gr.M = N; gr.No = DP;
  gr.a_amax1 = w_dout; gr.pieces = hop_pieces(h);
  HOP_TRY(hop_nn(h, &gr, h->W2t, DP, nullptr, 0, pkws, pk_elems, stream));
  // BatchNorm + ReLU backward: dbn[0] = d beta, dbn[1] = d gamma, then d h1
  HOP_TRY(qagnn_colreduce_f32(2, bufB, DP, h->h1, DP, N, DP, nullptr, 1, mean, invstd, scale, shift, nullptr, 1.0f, h->dbn, crws, stream));
  HOP_TRY(launch_bn_relu_bwd_colsum(bufB, h->h1, bufC, DP, N, DP, mean, invstd, scale, shift, h->gamma, h->dbn, h->dbn + DP,
                                    h->batch_stats ? (float)(1.0 / (double)N) : 0.f, nullptr, h->db1, crws, w_dh1, (hipStream_t)stream));
  // first Linear (db1 = colsum(d h1) came out of the pass above)
  if (ss->side) HOP_TRY(stream_after(ss->side, ss->main, ss->take()));
  if (h2) HOP_TRY(hop_tn(h)(h->aggr, DP, DP, nullptr, 0, 0, bufC, DP, h->dW1t, DP, N, DP, nullptr, nullptr, am.own + AM_AGGR, nullptr, w_dh1, tnws, wstream));
  else HOP_TRY(qagnn_gemm_tn_f32(h->aggr, DP, bufC, DP, h->dW1t, DP, N, DP, DP, nullptr, nullptr, nullptr, 0, tnws, wstream));
  qagnn_gemm_nn_args gg = {};
  gg.A1 = bufC; gg.lda1 = DP; gg.K1 = DP; gg.B1 = h->W1; gg.ldb1 = DP; gg.C = bufB; gg.ldc = DP; gg.M = N; gg.No = DP;
  gg.a_amax1 = w_dh1; gg.pieces = hop_pieces(h);
  HOP_TRY(hop_nn(h, &gg, h->W1t, DP, nullptr, 0, pkws, pk_elems, stream));
  // attention backward
  HOP_TRY(launch_edge_attn_bwd(h->g, h->KMQ, 3 * DP, h->EkEm, 2 * DP, h->HP, h->qscale, h->a, h->alpha, bufB, DP, dKMQ, h->dEkEm, gab, rs, cls_part,
                               h2 ? ampart : nullptr, w_dkmq, (hipStream_t)stream));
  // projection: weight gradients, node-type-table gradient, data gradients
  if (ss->side) HOP_TRY(stream_after(ss->side, ss->main, ss->take()));
  if (SP > 0 && h->dWs_t == h->dWx_t + (int64_t)DP * 3 * DP) {  // the two gradients are one [DP + SP, 3 DP] matrix: one launch
    if (h2) HOP_TRY(hop_tn(h)(h->X, DP, DP, h->S, SP, SP, dKMQ, 3 * DP, h->dWx_t, 3 * DP, N, 3 * DP, nullptr, nullptr, am.x, am.s, w_dkmq, tnws, wstream));
    else HOP_TRY(qagnn_gemm_tn2_f32(h->X, DP, DP, h->S, SP, SP, dKMQ, 3 * DP, h->dWx_t, 3 * DP, N, 3 * DP, tnws, wstream));
  } else if (h2) {
    HOP_TRY(hop_tn(h)(h->X, DP, DP, nullptr, 0, 0, dKMQ, 3 * DP, h->dWx_t, 3 * DP, N, 3 * DP, nullptr, nullptr, am.x, nullptr, w_dkmq, tnws, wstream));
    if (SP > 0)
      HOP_TRY(hop_tn(h)(h->S, SP, SP, nullptr, 0, 0, dKMQ, 3 * DP, h->dWs_t, 3 * DP, N, 3 * DP, nullptr, nullptr, am.s, nullptr, w_dkmq, tnws, wstream));
  } else {
    HOP_TRY(qagnn_gemm_tn_f32(h->X, DP, dKMQ, 3 * DP, h->dWx_t, 3 * DP, N, DP, 3 * DP, nullptr, nullptr, nullptr, 0, tnws, wstream));
    if (SP > 0)
      HOP_TRY(qagnn_gemm_tn_f32(h->S, SP, dKMQ, 3 * DP, h->dWs_t, 3 * DP, N, SP, 3 * DP, nullptr, nullptr, nullptr, 0, tnws, wstream));
  }
  if (SP > 0 && h->tab_col >= 0) {
    // the type indicators ride in S's padding columns: their rows of dWs_t ARE the type-table gradient
    QAGNN_REQUIRE(h->tab_col + h->T <= SP, QAGNN_EINVAL, "hop_bwd: tab_col=%d + T=%d exceeds SP=%d", h->tab_col, h->T, SP);
    if (h->dTT != h->dWs_t + (int64_t)h->tab_col * 3 * DP) {  // (no copy when the caller's dTT IS those rows)
      hipError_t he = hipMemcpyAsync(h->dTT, h->dWs_t + (int64_t)h->tab_col * 3 * DP, (size_t)h->T * 3 * DP * sizeof(float), hipMemcpyDeviceToDevice,
                                     (hipStream_t)wstream);
      if (he != hipSuccess) { set_error("hop_bwd: dTT copy failed: %s", hipGetErrorString(he)); return QAGNN_EHIP; }
    }
  } else {
    HOP_TRY(qagnn_colreduce_f32(0, dKMQ, 3 * DP, nullptr, 3 * DP, N, 3 * DP, h->ntype, h->T, nullptr, nullptr, nullptr, nullptr, nullptr, 1.0f, h->dTT,
                                crws, stream));
  }
  if (ss->side) {  // the side stream is done with this buffer set (and with everything queued on it before) once this event fires
    *done = ss->take();
    hipError_t he = hipEventRecord(*done, ss->side);
    QAGNN_REQUIRE(he == hipSuccess, QAGNN_EHIP, "hop_bwd: hipEventRecord failed: %s", hipGetErrorString(he));
  }
  if (h->dX) {
    qagnn_gemm_nn_args gx = {};
    gx.A1 = dKMQ; gx.lda1 = 3 * DP; gx.K1 = 3 * DP; gx.B1 = h->Wx; gx.ldb1 = DP; gx.C = h->dX; gx.ldc = DP; gx.M = N; gx.No = DP;
    gx.accumulate = h->accumulate_dX;
    gx.a_amax1 = w_dkmq; gx.pieces = hop_pieces(h);
    HOP_TRY(hop_nn(h, &gx, h->Wx_t, 3 * DP, nullptr, 0, pkws, pk_elems, stream));
  }
  if (SP > 0 && h->dS) {
    qagnn_gemm_nn_args gs = {};
    gs.A1 = dKMQ; gs.lda1 = 3 * DP; gs.K1 = 3 * DP; gs.B1 = h->Ws; gs.ldb1 = SP; gs.C = h->dS; gs.ldc = SP; gs.M = N; gs.No = SP;
    gs.accumulate = h->accumulate_dS;
    gs.a_amax1 = w_dkmq; gs.pieces = hop_pieces(h);
    HOP_TRY(hop_nn(h, &gs, h->Ws_t, 3 * DP, nullptr, 0, pkws, pk_elems, stream));
  }
  return QAGNN_OK;
}

// k hops, last first; the buffer sets alternate and the main stream joins the weight-gradient stream before it returns
static int stack_bwd_impl(const qagnn_hop_args* hops, int k, hipStream_t main) {
  SideSync ss;
  HOP_TRY(side_sync_init(&ss, main, (hipStream_t)hops[k - 1].side_stream));
  hipEvent_t done[2] = {nullptr, nullptr};
  int rc = QAGNN_OK;
  for (int it = 0, l = k - 1; l >= 0 && rc == QAGNN_OK; --l, ++it) {
    const int set = it & 1;
    if (ss.side && done[set]) {  // the hop before last read this set on the side stream
      hipError_t he = hipStreamWaitEvent(ss.main, done[set], 0);
      if (he != hipSuccess) { set_error("stack_bwd: hipStreamWaitEvent failed: %s", hipGetErrorString(he)); rc = QAGNN_EHIP; break; }
    }
    rc = hop_bwd_one(&hops[l], hop_amax_stack(hops, k, l), &ss, set, &done[set]);
  }
  // join even after an error: whatever was forked must not outlive the call (a capture would be left with an unjoined branch)
  if (ss.side) {
    hipEvent_t ev = ss.take();
    hipError_t he = hipEventRecord(ev, ss.side);
    if (he == hipSuccess) he = hipStreamWaitEvent(ss.main, ev, 0);
    if (he != hipSuccess && rc == QAGNN_OK) { set_error("stack_bwd: joining the weight-gradient stream failed: %s", hipGetErrorString(he)); rc = QAGNN_EHIP; }
  }
  return rc;
}

}  // namespace qagnn

extern "C" int qagnn_hop_bwd_f32(const qagnn_hop_args* h, qagnn_stream_t stream) {
  QAGNN_REQUIRE(h, QAGNN_EINVAL, "hop_bwd: null argument block");
  return stack_bwd_impl(h, 1, (hipStream_t)stream);
}

// ---- the whole k-hop stack per call (qagnn_mp_forward / qagnn_mp_backward) -------------------------------------
// hops[l] is a complete qagnn_hop_args; the caller chains them (hops[l+1].X = hops[l].y, hops[l].dy = hops[l+1].dX, shared dS
// with accumulate_dS = 1 on all but the hop whose backward runs first).  Pure sequencing: the same launches, in the same order,
// as k calls of qagnn_hop_fwd_f32 / qagnn_hop_bwd_f32 -- one FFI crossing and one autograd node instead of k for host-bound batches.
extern "C" int qagnn_stack_fwd_f32(const qagnn_hop_args* hops, int32_t k, qagnn_stream_t stream) {
  QAGNN_REQUIRE(hops && k > 0, QAGNN_EINVAL, "stack_fwd: no hops");
  // the hops' amax blocks start at zero: one launch when they are one array (the module mirror's are), one per hop otherwise
  bool any = false, one_array = true;
  for (int l = 0; l < k; ++l) {
    any = any || hop_h2(&hops[l]);
    one_array = one_array && hops[l].amax != nullptr && hops[l].amax == hops[0].amax + (int64_t)l * QAGNN_HOP_AMAX_WORDS;
  }
  if (any && one_array) HOP_TRY(qagnn_zero_words(hops[0].amax, (int64_t)k * QAGNN_HOP_AMAX_WORDS, stream));
  for (int l = 0; l < k; ++l) {
    const HopAmax am = hop_amax_stack(hops, k, l);
    if (am.on && !one_array) HOP_TRY(qagnn_zero_words(hops[l].amax, QAGNN_HOP_AMAX_WORDS, stream));
    HOP_TRY(hop_fwd_one(&hops[l], am, am.on && am.x != hops[l].amax + AM_X, am.on && am.s != hops[l].amax + AM_S, stream));
  }
  return QAGNN_OK;
}

extern "C" int qagnn_stack_bwd_f32(const qagnn_hop_args* hops, int32_t k, qagnn_stream_t stream) {
  QAGNN_REQUIRE(hops && k > 0, QAGNN_EINVAL, "stack_bwd: no hops");
  return stack_bwd_impl(hops, k, (hipStream_t)stream);
}
