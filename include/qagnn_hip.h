/*
 * qagnn_hip.h -- C ABI of libqagnn_hip.so: the MI355X (gfx950) kernels behind QA-GNN's GNN hot path.
 *
 * The reference (michiyasunaga/qagnn @ v1) has no FFI: its boundary for this path is the Python nn.Module
 * contract  QAGNN.forward(sent_vecs, concept_ids, node_type_ids, node_scores, adj_lengths, (edge_index, edge_type))
 * (modeling/modeling_qagnn.py:141) and every native kernel it runs comes from third-party wheels (torch-geometric
 * 1.7.0, torch-scatter 2.0.7, ATen/cuBLAS).  Each entry point below therefore cites the reference call site(s)
 * whose native kernels it replaces.  qagnn_amd/ (Python) mirrors the reference's module interface on top of this
 * ABI; INTEGRATION.md shows the ctypes binding a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers + sizes, all pointers are DEVICE pointers unless named h_*; no allocation inside the library;
 *     every call enqueues on `stream` and returns immediately (graph-capture safe, re-entrant per stream).
 *   - return value: QAGNN_OK or a QAGNN_E* code; qagnn_last_error() gives a message for the calling thread.
 *   - fp32 everywhere ("dtype f32"), int32 graph arrays, int64 only where the reference hands us int64 tensors.
 *   - "head-padded" node rows: a feature row of d = H*dh floats is stored as H groups of HP = roundup4(dh) floats
 *     (pads are zero), DP = H*HP floats per row (d=200,H=4: HP=52, DP=208 -> 832-byte rows, 16-byte aligned heads).
 *     H must be 4 (the reference hard-codes head_count=4, modeling_qagnn.py:387) and dh <= 64.
 */
#ifndef QAGNN_HIP_H
#define QAGNN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QAGNN_OK 0
#define QAGNN_EINVAL 1       /* bad argument (null pointer, misaligned, size constraint violated) */
#define QAGNN_EUNSUPPORTED 2 /* shape outside what the kernels were written for */
#define QAGNN_EHIP 3         /* a HIP runtime call failed; see qagnn_last_error() */

typedef void* qagnn_stream_t; /* hipStream_t */

const char* qagnn_last_error(void);
int qagnn_abi_version(void);

/* ------------------------------------------------------------------------------------------------------------
 * Graph preparation (integer).  Replaces, once per batch instead of once per layer:
 *   modeling_qagnn.py:419-438  make_one_hot x4, node_type[edge_index[0/1]], self-loop append
 *   modeling_qagnn.py:476-479  torch_scatter out-degree
 *   PyG softmax / scatter      the implicit grouping by edge_index[0] (softmax) and edge_index[1] (aggregation)
 * Edge ids: 0..E-1 are the caller's edges, E+v is the self loop of node row v (appended for ALL N rows, :436-438).
 * Edge class  c = etype*T*T + ntype[src]*T + ntype[tgt]  for real edges,  R*T*T + ntype[v]  for self loops;
 * the edge encoder's input one-hot (:419-433) is a function of c alone, C = R*T*T + T classes.
 * All three orders are sorted by (group key, edge id), i.e. deterministic and equal to the reference's CPU
 * summation order inside every group.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct qagnn_graph {
  int32_t N, E, Ep, R, T, C;   /* Ep = E + N */
  /* grouped by SOURCE (softmax segments), position p in [0,Ep) */
  int32_t* rowptr_s;           /* [N+1] */
  int32_t* tgt_s;              /* [Ep] target node of position p */
  int32_t* src_s;              /* [Ep] source node of position p (segment owner, for edge-parallel kernels) */
  int32_t* cls_s;              /* [Ep] edge class */
  int32_t* eid_s;              /* [Ep] edge id (caller order) at position p */
  /* grouped by TARGET (aggregation segments) */
  int32_t* rowptr_t;           /* [N+1] */
  int32_t* src_t;              /* [Ep] */
  int32_t* tgt_t;              /* [Ep] target node of position p (segment owner, for edge-parallel kernels) */
  int32_t* cls_t;              /* [Ep] */
  int32_t* pos_t;              /* [Ep] position of the same edge in the source order */
  /* grouped by (POSITION GROUP, CLASS): group = a run of consecutive source-order positions (the edges of a few neighbouring
   * subgraphs; n_groups <= QAGNN_CLS_GROUPS per batch), then class; cut into chunks of <= QAGNN_CLS_CHUNK edges that never
   * straddle a (group, class) pair.  Keeps the class pass's row gathers inside one XCD's L2. */
  int32_t* cls_count;          /* [C]   edges per class (the count-weighted BatchNorm of the edge encoder needs it) */
  int32_t* src_c;              /* [Ep] */
  int32_t* tgt_c;              /* [Ep] */
  int32_t* pos_c;              /* [Ep] position in the source order */
  int32_t* chunk_cls;          /* [max_chunks] */
  int32_t* chunk_beg;          /* [max_chunks] */
  int32_t* chunk_len;          /* [max_chunks] */
  int32_t* n_chunks;           /* [1] device scalar */
  int32_t* chunkptr;           /* [n_groups*C+1] first chunk of pair g*C + c */
  int32_t max_chunks;          /* Ep / QAGNN_CLS_CHUNK + n_groups*C + 1 */
  int32_t* err;                /* [16]; [0..3] device flags: [0] = 1: an index was out of range (it was clamped);
                                  [1] = 1: some edge leaves its block of block_n consecutive node rows;
                                  [4..12] the XCD partition of the node-side edge kernels: XCD k walks the 4-node blocks
                                  [err[4 + k], err[5 + k]), an eighth of the batch's edge work each (csrc/graph_prep.hip) */
  int32_t block_n;             /* 0, or the node-block size the graph was checked against (subgraph = n consecutive rows) */
  int32_t n_groups;            /* position groups of the class order */
} qagnn_graph;

#define QAGNN_CLS_CHUNK 64
#ifndef QAGNN_CLS_SLICES
#define QAGNN_CLS_SLICES 4   /* the class reduction sums a class's partials in this many independent slices first (1 / 2 / 4 / 8 / 16 measured: 0.318 / 0.298 / 0.293 / 0.296 / 0.306 ms) */
#endif
#ifndef QAGNN_CLS_GROUPS
#define QAGNN_CLS_GROUPS 32 /* measured 16 / 32 / 64 / 128: backward edge stage 0.286 / 0.288 / 0.302 / 0.329 ms per layer at B = 320 */
#endif

/* int32 elements of device storage needed for all arrays of a qagnn_graph plus scratch. */
int64_t qagnn_graph_storage_elems(int32_t N, int32_t E, int32_t R, int32_t T);
/* Carve `storage` (int32, qagnn_graph_storage_elems elements, 16-byte aligned) into *g and build everything. */
int qagnn_graph_prep(qagnn_graph* g, int32_t* storage, const int64_t* edge_index /* [2][E] */, const int64_t* edge_type /* [E] */,
                     const int64_t* node_type /* [N] */, int32_t N, int32_t E, int32_t R, int32_t T, qagnn_stream_t stream);
/* Same; additionally records (device flag err[1]) whether every edge stays inside its block of `block_n` consecutive node
 * rows -- true for batches built by LM_QAGNN.batch_graph (modeling_qagnn.py:244-251: subgraph i owns rows [i*n, (i+1)*n)).
 * The LDS-resident edge kernel below is taken only for such graphs; the decision is made ON THE DEVICE (no host sync). */
int qagnn_graph_prep_blocked(qagnn_graph* g, int32_t* storage, const int64_t* edge_index, const int64_t* edge_type,
                             const int64_t* node_type, int32_t N, int32_t E, int32_t R, int32_t T, int32_t block_n,
                             qagnn_stream_t stream);

/* The same graph from per-sample blobs built once at LOAD time.  Replaces the per-batch work of
 *   utils/data_utils.py:53-76   2*bs*nc individual .to(device) copies of int64 edge lists
 *   modeling_qagnn.py:244-251   LM_QAGNN.batch_graph (offsets + cat), and the five sorting launches of qagnn_graph_prep.
 * `blobs` is the concatenation of the batch's B sample blobs, blob_off[g] the int32-word offset of sample g in it, edge_off[g]
 * the number of real edges of the samples before g (edge_off[B] = E).  One blob = cnt_s[n] | cnt_t[n] | w0[E_g] | w1[E_g] | w2[E_g]:
 *   cnt_s, cnt_t   out- / in-degree of each of the sample's n node slots, real edges only
 *   w0[i] = tgt | cls << 16     edge i of the sample's SOURCE order (sorted by (src, local edge id)): local target, edge class
 *   w1[i] = eid | src << 16     eid: local edge id of source-order edge i;   src: local source of TARGET-order edge i
 *   w2[i]                       position in the local source order of TARGET-order edge i (sorted by (tgt, local edge id))
 * i.e. 12 bytes per edge + 8 per node slot; self loops are not stored (inserted here, one per node row, last in their segments).
 * node_type is the batch's [B*n] int64 tensor (the self loops' classes).  Produces arrays bit-identical to
 * qagnn_graph_prep_blocked(edge_index = batch_graph(...), block_n = n); err[0] is set if a blob field is out of range.
 * `E` is the edge CAPACITY the arrays (and every launch shape that follows: g->E, g->Ep, g->max_chunks) are laid out for; the batch's
 * true edge count is read on the device from edge_off[B] and must not exceed it (the caller packed the batch: it knows).  With
 * E == edge_off[B] this is the plain call; with a bucketed E one hipGraph capture of the whole step serves every batch of the bucket
 * (qagnn_amd/graphed.py).  The true E' = edge_off[B] + B*n stays available on the device as g->rowptr_s[N]. */
int qagnn_graph_from_blobs(qagnn_graph* g, int32_t* storage, const int32_t* blobs, const int32_t* blob_off /* [B+1] */,
                           const int32_t* edge_off /* [B+1] */, const int64_t* node_type /* [B*n] */, int32_t B, int32_t n, int32_t E,
                           int32_t R, int32_t T, qagnn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Dense fp32 MFMA GEMMs (v_mfma_f32_16x16x4_f32).  Replace the cuBLAS SGEMMs of
 *   modeling_qagnn.py:464-466 (linear_key/msg/query, after the project-then-gather rewrite: N rows, not E'),
 *   modeling_qagnn.py:443,408 (GATConvE.mlp), :92 (Vh, Vx), :73 (emb_score) and their autograd backward.
 * NN:  C[M][ldc] (+)= [A1 | A2][M][K1+K2] * [B1 ; B2][K1+K2][No]  + bias[No] + rowtab[rowidx[m]][No]
 *      optional A prologue  a <- max(0, a*a_scale[k] + a_shift[k])   (BatchNorm+ReLU folded into the operand load)
 * TN:  C[Ka][ldc] (+)= A[R][Ka]^T * B[R][No]      (weight gradients; split over rows, deterministic two-stage sum)
 * Constraints: K1, K2, Ka multiples of 16... see each function; all row pitches multiples of 4 floats, 16-byte aligned.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct qagnn_gemm_nn_args {
  const float* A1; int32_t lda1; int32_t K1; const float* B1; int32_t ldb1;
  const float* A2; int32_t lda2; int32_t K2; const float* B2; int32_t ldb2; /* A2 may be NULL (K2 = 0) */
  float* C; int32_t ldc; int32_t M; int32_t No;
  const float* bias;                 /* [No] or NULL */
  const float* rowtab; int32_t ldt;  /* [G][ldt] or NULL */
  const int64_t* rowidx;             /* [M] row -> table row (node_type ids are int64 in the reference) */
  const float* a_scale; const float* a_shift; /* [K1] or NULL; applies to A1 only */
  int32_t accumulate;                /* 1: C += result */
  const int64_t* a_rowidx;           /* [M] or NULL: row m of A1 is A1[a_rowidx[m]] (embedding-table gather fused into the
                                        operand load, utils/layers.py:604-605); a negative index reads a zero row */
  int32_t xcd_remap;                 /* set by the library (XCD-contiguous tile order); callers leave it 0 */
  float* colstat_part;               /* NULL, or [ceil(M/128)][3][No] (qagnn_gemm_nn_split_f32 only, No <= 208, bias-only epilogue):
                                        per 128-row tile t and output column c, over the tile's rows of C:  x0 = C[first row][c],
                                        S1 = sum (C - x0),  S2 = sum (C - x0)^2 -- BatchNorm batch statistics as a by-product of the
                                        GEMM that produces the BatchNorm input (qagnn_bn_stats_finalize_f32 combines the tiles) */
  const uint32_t* a_amax1;           /* NULL, or a device word holding the BIT PATTERN of (an upper bound within 2^8 of) max |A1| -- after the
                                        a_scale / a_shift prologue where there is one -- and likewise a_amax2 for A2 (ignored when K2 = 0).
                                        qagnn_gemm_nn_split*_f32 only: with both known, a large product takes the THREE-MFMA form (scaled
                                        two-piece fp16 split, csrc/gemm_nn2.hip; ~2^-21 per product instead of 2^-23).  A word that
                                        understates the maximum by 2x or more makes the result inf / nan (never silently wrong): the
                                        producers of this library fill it exactly (qagnn_absmax_f32 and the *_amax arguments) */
  const uint32_t* a_amax2;
  int64_t a_rows;                    /* with a_rowidx: the number of rows of A1's storage (the entity table), or 0 = unknown.  Known and below
                                        2 GB, a gathered product of qagnn_gemm_nn_split*_f32 takes the second-generation kernels (and, with
                                        a_amax1 = the bit pattern of an upper bound of max |table| -- the table is frozen, one reduction when it
                                        is loaded --, the three-MFMA form) instead of the first-generation gather kernel */
  int32_t pieces;                    /* 0 (default): full-accuracy arithmetic, see above.  1: the REDUCED-PRECISION form, on request only and only
                                        where the three-MFMA form would run (a_amax known): ONE fp16 MFMA per product -- both operands rounded to
                                        fp16 (11 significant bits) under the same power-of-two scales, fp32 accumulation and fp32 storage; what
                                        torch.autocast makes of the reference's Linear layers (qagnn.py:254-257, --fp16), minus its fp16 outputs */
} qagnn_gemm_nn_args;
int qagnn_gemm_nn_f32(const qagnn_gemm_nn_args* a, qagnn_stream_t stream);
/* The same product on the bf16 matrix cores by EXACT operand splitting (csrc/gemm_split.hip): every fp32 operand is the exact sum
 * of three bf16 numbers; the six partial products of order >= 2^-16 are accumulated in fp32 (six v_mfma_f32_16x16x32_bf16 per
 * tile pair at 16x the fp32-MFMA rate).  Relative error <= 2^-23 per product -- one fp32 rounding -- so the result agrees with
 * qagnn_gemm_nn_f32 to fp32 round-off but not bit for bit.  B is passed in its [No][K] layout (B1n / B2n, pitches ldn1 / ldn2:
 * row j = column j of B1 / B2); a->B1 / a->B2 are ignored.  K1, K2 multiples of 4 (any length; tiles are zero-filled). */
int qagnn_gemm_nn_split_f32(const qagnn_gemm_nn_args* a, const float* B1n, int32_t ldn1, const float* B2n, int32_t ldn2,
                            qagnn_stream_t stream);
/* The same with a caller-provided scratch buffer `ws` of `ws_bytes` >= qagnn_gemm_nn_pack_bytes(No, K1, K2) bytes (16-byte aligned,
 * contents undefined before and after): large products (csrc/gemm_nn2.hip) first write B there ONCE, already split into its three
 * bf16 images in the order the kernel's LDS wants them, and every row tile then streams it into LDS by DMA instead of repeating the
 * split (19 % of the projection [N, 320] x [320, 624] at N = 64 000).  ws = NULL (or too small, or a small product) = qagnn_gemm_nn_split_f32.
 * Same arithmetic per output element as the unpacked route: bit-identical results. */
int64_t qagnn_gemm_nn_pack_bytes(int32_t No, int32_t K1, int32_t K2); /* (sized for either arithmetic form) */
/* Bytes of scratch qagnn_gemm_nn_split_ws_f32 would USE for this very call: 0 when B is a registered (pre-packed) operand, when the product
 * is not one the packed kernels take, or when it has too few rows to pay for a pack launch -- a caller that allocates per call asks first. */
int64_t qagnn_gemm_nn_ws_bytes(const qagnn_gemm_nn_args* a, const float* B1n, int32_t ldn1, const float* B2n, int32_t ldn2);
/* All B operands of a step's large NN products packed in ONE launch, ahead of the products: `d[i]` names a weight in its [No][K]
 * layout(s) exactly as the product will pass it (B1n / ldn1 / K1, B2n / ldn2 / K2, No); the images go to `out`
 * (>= qagnn_gemm_nn_prepack_bytes(d, n) bytes, 16-byte aligned) and are REGISTERED under `tag` (!= 0, replaces the tag's earlier
 * entries): qagnn_gemm_nn_split_f32 / _ws_f32 recognise a registered operand by its pointers and sizes and skip their own packing.
 * Contract: the fp32 weights and `out` stay alive and unchanged until qagnn_gemm_nn_prepack_clear(tag) or the next prepack under the
 * same tag (the module mirror packs behind its operand-packing gather every forward and holds both tensors).  Entries that the
 * packed kernels do not take (K not a multiple of 8, ...) are skipped silently: their products pack per call as before.  Host-side
 * registry, mutex-protected; clear(0) empties it. */
typedef struct qagnn_pack_desc { const float* B1n; int32_t ldn1; int32_t K1; const float* B2n; int32_t ldn2; int32_t K2; int32_t No;
                                 int32_t pieces; /* 0 / 3: the three bf16 images; 2: the two scaled fp16 images of the three-MFMA form (taken by
                                                    products that come with a_amax1 / a_amax2); 1: the one image of the reduced-precision form
                                                    (qagnn_gemm_nn_args.pieces = 1) */ } qagnn_pack_desc;
int64_t qagnn_gemm_nn_prepack_bytes(const qagnn_pack_desc* d, int32_t n);
int qagnn_gemm_nn_prepack_f32(const qagnn_pack_desc* d, int32_t n, void* out, int64_t out_bytes, int64_t tag, qagnn_stream_t stream);
int qagnn_gemm_nn_prepack_clear(int64_t tag);
int qagnn_gemm_nn_split_ws_f32(const qagnn_gemm_nn_args* a, const float* B1n, int32_t ldn1, const float* B2n, int32_t ldn2,
                               void* ws, int64_t ws_bytes, qagnn_stream_t stream);

/* max |x| over n floats (n % 4 == 0, x 16-byte aligned), merged into *slot by an integer atomic max on the bit pattern -- order-independent, hence
 * deterministic; the caller zeroes the word first (qagnn_zero_words).  The operand maxima of the three-MFMA GEMM form (a_amax1 / a_amax2 above,
 * qagnn_gemm_tn_h2_f32 below) for tensors whose producer does not leave one.  NaNs are skipped; an inf gives 0x7F800000 (scale 1). */
int qagnn_absmax_f32(const float* x, int64_t n, uint32_t* slot, qagnn_stream_t stream);
int qagnn_zero_words(uint32_t* p, int64_t n, qagnn_stream_t stream);

/* workspace floats needed by qagnn_gemm_tn_f32 for (R, Ka, No) (includes room for the optional column sums of B) */
int64_t qagnn_gemm_tn_workspace_elems(int32_t R, int32_t Ka, int32_t No);
int qagnn_gemm_tn_f32(const float* A, int32_t lda, const float* B, int32_t ldb, float* C, int32_t ldc, int32_t R, int32_t Ka,
                      int32_t No, const float* a_scale, const float* a_shift /* BN+ReLU prologue on A, or NULL */,
                      const int64_t* a_rowidx /* [R] or NULL: row r of A is A[a_rowidx[r]], negative = zero row */,
                      int32_t accumulate, float* workspace, qagnn_stream_t stream);
/* C [Ka1 + Ka2, No] = [A1 | A2]^T B (rows [0, Ka1) from A1, the rest from A2): the two weight gradients of a product with two A operands
 * (modeling_qagnn.py:464-466 on [x ; extra]: dWx^T = X^T dK|dM|dQ, dWs^T = S^T dK|dM|dQ) in ONE split-K launch and one chunk sum where
 * the bf16-split kernel takes the shapes, two qagnn_gemm_tn_f32 calls otherwise.  workspace: qagnn_gemm_tn_workspace_elems(R, Ka1 + Ka2, No). */
int qagnn_gemm_tn2_f32(const float* A1, int32_t lda1, int32_t Ka1, const float* A2, int32_t lda2, int32_t Ka2, const float* B, int32_t ldb,
                       float* C, int32_t ldc, int32_t R, int32_t No, float* workspace, qagnn_stream_t stream);
/* C [Ka1 + Ka2, No] = [A1 | A2]^T B as qagnn_gemm_tn_f32 (Ka2 = 0) / qagnn_gemm_tn2_f32, in the three-MFMA form: every operand comes with the bit
 * pattern of its max |.| (amax_a1 covers A1 AFTER the a_scale / a_shift prologue).  Falls back to the six-MFMA kernels (same results to fp32
 * round-off, amax ignored) for shapes the split kernels do not take.  workspace: qagnn_gemm_tn_workspace_elems(R, Ka1 + Ka2, No). */
int qagnn_gemm_tn_h2_f32(const float* A1, int32_t lda1, int32_t Ka1, const float* A2, int32_t lda2, int32_t Ka2, const float* B, int32_t ldb,
                         float* C, int32_t ldc, int32_t R, int32_t No, const float* a_scale, const float* a_shift, const uint32_t* amax_a1,
                         const uint32_t* amax_a2, const uint32_t* amax_b, float* workspace, qagnn_stream_t stream);
/* The REDUCED-PRECISION form of the same call (on request only: qagnn_hop_args.gemm_split == 3): ONE fp16 MFMA per product, both operands
 * rounded to fp16 under the same power-of-two scales, fp32 accumulation -- the arithmetic torch.autocast gives the reference's Linear
 * layers in backward (qagnn.py:254-257, --fp16).  Same fallbacks as qagnn_gemm_tn_h2_f32. */
int qagnn_gemm_tn_h1_f32(const float* A1, int32_t lda1, int32_t Ka1, const float* A2, int32_t lda2, int32_t Ka2, const float* B, int32_t ldb,
                         float* C, int32_t ldc, int32_t R, int32_t No, const float* a_scale, const float* a_shift, const uint32_t* amax_a1,
                         const uint32_t* amax_a2, const uint32_t* amax_b, float* workspace, qagnn_stream_t stream);
/* Same, and additionally  bsum[g][no] = sum_r [grp(r) == g] B[r][no]  (groups in 1..4; b_rowidx NULL = one group): the
 * bias gradient (and the node-type-table gradient) of a Linear falls out of the weight-gradient GEMM's B tiles for free
 * instead of costing separate passes over dC. */
int qagnn_gemm_tn_colsum_f32(const float* A, int32_t lda, const float* B, int32_t ldb, float* C, int32_t ldc, int32_t R, int32_t Ka,
                             int32_t No, const float* a_scale, const float* a_shift, const int64_t* a_rowidx, int32_t accumulate,
                             float* bsum /* [groups][No] */, const int64_t* b_rowidx /* [R] or NULL */, int32_t groups,
                             float* workspace, qagnn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Column reductions over rows (bias / BatchNorm gradients, batch statistics).  Replace ATen sum / BatchNorm1d
 * statistics kernels (modeling_qagnn.py:408 BatchNorm1d over all N rows, PAD rows included).
 *   mode 0: out[g][c] = sum_r [grp(r)==g] X[r][c]                 (grp = rowidx or a single group)
 *   mode 1: out[c]    = sum_r (X[r][c] - mean[c])^2               (two-pass variance)
 *   mode 2: out[0][c] = sum_r dY[r][c],  out[1][c] = sum_r dY[r][c] * (H[r][c]-mean[c])*invstd[c]
 *           with dY = dR * [H*scale+shift > 0]   (BatchNorm+ReLU backward reductions; X = dR, X2 = H)
 * workspace: qagnn_colreduce_workspace_elems floats.
 * ------------------------------------------------------------------------------------------------------------ */
int64_t qagnn_colreduce_workspace_elems(int32_t R, int32_t Cc, int32_t groups);
int qagnn_colreduce_f32(int32_t mode, const float* X, int32_t ldx, const float* X2, int32_t ldx2, int32_t R, int32_t Cc,
                        const int64_t* rowidx, int32_t groups, const float* mean, const float* invstd, const float* scale,
                        const float* shift, const float* roww /* optional per-row weight (modes 0, 1): weighted sums */,
                        float out_scale /* multiplies the result, e.g. 1/R for a mean */, float* out, float* workspace,
                        qagnn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Elementwise kernels.
 *   bn_relu_bwd:  dH = gscale[c] * (dY - c1[c] - hhat*c2[c]),  dY = dR*[H*scale+shift>0], hhat=(H-mean)*invstd
 *                 (train: c1 = mean(dY), c2 = mean(dY*hhat); eval: c1 = c2 = 0)       -- BatchNorm1d backward, :408
 *   gelu_dropout: Y = gelu_tanh(X) * keep/(1-p)     (utils/layers.py:10-14 + F.dropout, modeling_qagnn.py:48-49,92-93)
 *                 keep is a counter-based hash of (seed, element index); backward regenerates it.
 *   sin_basis:    out[r][j] = sin(js[j] * score[r])  for j < J, 0 for J <= j < ldo   (modeling_qagnn.py:70-72)
 * ------------------------------------------------------------------------------------------------------------ */
/* BatchNorm1d bookkeeping in one launch: invstd/scale/shift from (mean, var, gamma, beta, eps) on the head-padded axis [Cc]; when
 * run_mean != NULL also the train-mode update of the module's dense running buffers [d] (momentum, unbiased variance = var * unbias)
 * and num_batches_tracked += 1.  dense_pos[k] = padded column of dense feature k. */
int qagnn_bn_finalize_f32(const float* mean, const float* var, const float* gamma, const float* beta, float eps, float* invstd,
                          float* scale, float* shift, int32_t Cc, float* run_mean, float* run_var, int64_t* num_batches_tracked,
                          const int64_t* dense_pos, int32_t d, float momentum, float unbias,
                          int32_t ones_col /* -1, or a zero-padding column c: scale[c] = 0, shift[c] = 1, i.e. relu(bn(h))[:, c] = 1 -- the
                                              weight gradient relu(bn(h))^T dout then carries the bias gradient colsum(dout) in row c */,
                          qagnn_stream_t stream);
int qagnn_bn_relu_bwd_f32(const float* dR, const float* Hh, float* dH, int32_t ld, int32_t R, int32_t Cc, const float* mean,
                          const float* invstd, const float* scale, const float* shift, const float* gamma, const float* sum_dy,
                          const float* sum_dy_hhat, float inv_rows /* 1/R with batch statistics, 0 with running statistics */,
                          const float* roww /* optional: per-row statistics weight instead of inv_rows */, qagnn_stream_t stream);
/* Pooling head (reference utils/layers.py:284-299 inside :344-371, called at modeling_qagnn.py:178), node-sized part, one
 * workgroup per subgraph.  u [B, NH, Cc]: query seen from node space (Wk_h^T w_qs(q)); cvec [B, NH]: <w_qs(q)_h, bk_h>;
 * K [B*n, ldk]: node rows (head-padded GNN output); mask [B, n]: 1 = node excluded (score -inf).
 *   score = (<u, k> + c) * inv_temp;  attn = softmax over the n nodes;  attn_d = dropout(attn, p, seed);  z = sum_l attn_d k
 * NH <= 4, Cc <= 256 (multiple of 4), n <= 1024.  bwd writes dK (not accumulates); dattn_d may be NULL. */
int qagnn_pool_attn_fwd_f32(const float* u, const float* cvec, const float* K, int32_t ldk, const uint8_t* mask, int32_t B, int32_t n,
                            int32_t NH, int32_t Cc, float inv_temp, float p, uint64_t seed, float* attn, float* attn_d, float* z,
                            qagnn_stream_t stream);
int qagnn_pool_attn_bwd_f32(const float* u, const float* K, int32_t ldk, int32_t B, int32_t n, int32_t NH, int32_t Cc, float inv_temp,
                            float p, uint64_t seed, const float* attn, const float* attn_d, const float* dz, const float* dattn_d,
                            float* dK, int32_t lddk, float* du, float* dc, qagnn_stream_t stream);
/* The rest of the head behind the pooling, one workgroup per subgraph (reference utils/layers.py:366-371: the value projection of the
 * pooled rows + `self.dropout`; modeling_qagnn.py:178-182 with fc_layer_num = 0: `concat = dropout_fc(cat(graph_vecs, sent_vecs, Z_vecs))`,
 * `logits = fc(concat)`):
 *   out[b]    = Wv_h z[b,h] + bv_h sum_l attn[b,h,l]                 z, attn: outputs of qagnn_pool_attn_fwd_f32 (attn = its attn_d)
 *   logits[b] = < drop_fc([ drop_pool(out[b]) | sent[b] | Z[b] ]), w_fc > + b_fc
 * BDv [NH*DP][NH*dv]: blockdiag(Wv_h^T) at the head-padded positions of the node features; H + b*ldh = row 0 of subgraph b in the
 * head-padded GNN output (4 GAT heads of d/4 features in DP/4 slots each), Z[b] its d dense features; w_fc [NH*dv + Ds + d].
 * Saved for the backward: out [B][NH*dv] (before dropout), asum [B][NH].  Masks are counter-based (seed_pool / seed_fc + the seed epoch).
 * Backward: dz [B][NH][DP], dattn [B][NH][n] (the gradient through sum_l attn), dout [B][NH*dv] (dBDv = z^T dout), dsent [B][Ds] or
 * NULL, dZ [B][DP] (padded, to be added to row 0 of the pooling's dK: qagnn_add_row0_f32), and part [B][ldp >= L + NH*dv + 1], L = NH*dv + Ds + d:
 * per subgraph the addends of d w_fc | d bv | d b_fc, whose column sums are those gradients.
 * QAGNN_EUNSUPPORTED outside NH <= 4, NH*dv <= 256, DP <= 256. */
int qagnn_head_post_fwd_f32(const float* z, const float* attn, const float* BDv, const float* bv, const float* sent, const float* H, int64_t ldh,
                            const float* w_fc, const float* b_fc, int32_t B, int32_t NH, int32_t DP, int32_t dv, int32_t n, int32_t Ds, int32_t d,
                            float p_pool, float p_fc, uint64_t seed_pool, uint64_t seed_fc, float* out, float* asum, float* logits,
                            qagnn_stream_t stream);
int qagnn_head_post_bwd_f32(const float* dlogits, const float* out, const float* asum, const float* BDv, const float* bv, const float* sent,
                            const float* H, int64_t ldh, const float* w_fc, int32_t B, int32_t NH, int32_t DP, int32_t dv, int32_t n, int32_t Ds,
                            int32_t d, float p_pool, float p_fc, uint64_t seed_pool, uint64_t seed_fc, float* dz, float* dattn, float* dout,
                            float* dsent, float* dZ, float* part, int32_t ldp /* row pitch of part, >= L + NH*dv + 1; extra columns are zeroed */,
                            qagnn_stream_t stream);
/* dK[b*ld_sub + j] += dZ[b*Cc + j]: the gradient of each subgraph's row 0 that the head reads directly, into the pooling's dK */
int qagnn_add_row0_f32(float* dK, int64_t ld_sub, const float* dZ, int32_t B, int32_t Cc, qagnn_stream_t stream);
/* Weight packing of the module mirror (qagnn_amd.ops.GatherPlan; no reference counterpart: the reference multiplies by nn.Linear weights
 * in place, the kernels of this library want them transposed / head-padded / concatenated).  qagnn_gather_multi_f32: out[i] =
 * p[tid[i]][off[i]], 0 where tid[i] < 0 -- which tensor and which element a packed position comes from is fixed when the plan is built,
 * only the tensors' addresses change from call to call.  qagnn_gather_multi_sum_f32: out[s] = sum over k < K, in that order, of
 * p[tid[k][s]][off[k][s]] (tid < 0 or a NULL p[.]: 0), tid / off being [K][S]: the backward of the packing, its packed gradients being
 * separate, possibly absent tensors. */
#define QAGNN_GATHER_MAX 160
typedef struct qagnn_gather_tabs { const float* p[QAGNN_GATHER_MAX]; int32_t n; } qagnn_gather_tabs;
int qagnn_gather_multi_f32(const qagnn_gather_tabs* t, const int32_t* tid, const int32_t* off, float* out, int32_t total, qagnn_stream_t stream);
int qagnn_gather_multi_sum_f32(const qagnn_gather_tabs* t, const int32_t* tid, const int32_t* off, int32_t K, int32_t S, float* out,
                               qagnn_stream_t stream);
/* Launch timing for measurement harnesses (bench.py): while enabled, the NN-product, weight-gradient-product and edge-stage entry points
 * (also when the natively sequenced hop / stack calls them) bracket what they launch with HIP events on their launch stream.
 * qagnn_timing_enable(1) clears the record and starts, (0) stops; qagnn_timing_read synchronises on the recorded events and returns the
 * summed milliseconds and the number of bracketed calls per kind: 0 NN products, 1 weight-gradient products (incl. their chunk sums),
 * 2 edge forward, 3 edge backward.  Off by default; not for use around a stream capture.  (The reference has no counterpart: it is what
 * torch.profiler gives its users for free, modeling_qagnn.py's ops being stock torch kernels.) */
#define QAGNN_TIMING_KINDS 4
int qagnn_timing_enable(int32_t on);
int qagnn_timing_read(double* ms /* [QAGNN_TIMING_KINDS] */, int64_t* calls /* [QAGNN_TIMING_KINDS] */);
int qagnn_gelu_dropout_fwd_f32(const float* X, float* Y, int64_t n, float p, uint64_t seed, qagnn_stream_t stream);
int qagnn_gelu_dropout_bwd_f32(const float* X, const float* dY, float* dX, int64_t n, float p, uint64_t seed, qagnn_stream_t stream);
/* qagnn_gelu_dropout_fwd_f32 that also merges max |Y| into *amax (bit pattern, integer atomic max; the caller zeroed the word): the operand
 * maximum of the three-MFMA GEMM form for a consumer of Y (qagnn_hop_args.x_amax / s_amax, qagnn_gemm_nn_args.a_amax1) without a pass over Y */
int64_t qagnn_gelu_dropout_amax_scratch_elems(int64_t n);
int qagnn_gelu_dropout_fwd_amax_f32(const float* X, float* Y, int64_t n, float p, uint64_t seed, uint32_t* amax,
                                    float* scratch /* qagnn_gelu_dropout_amax_scratch_elems(n) floats */, qagnn_stream_t stream);
/* likewise the backward pass, which leaves max |dX| (the B operand of the weight-gradient product and the A operand of the data-gradient
 * products of the Linear in front of the GELU: modeling_qagnn.py:92-93 behind Vh / Vx) */
int qagnn_gelu_dropout_bwd_amax_f32(const float* X, const float* dY, float* dX, int64_t n, float p, uint64_t seed, uint32_t* amax,
                                    float* scratch, qagnn_stream_t stream);
/* qagnn_bn_relu_bwd_f32 with the column sums of its OUTPUT as a by-product (the bias gradient of the Linear in front of the
 * BatchNorm): one pass instead of an elementwise pass + a column-reduction pass; sums bit-identical to qagnn_colreduce_f32
 * mode 0 on the output.  workspace: qagnn_colreduce_workspace_elems(R, Cc, 1) floats.  (The same fusion for the GELU + dropout
 * backward measured no gain: that kernel is ALU-bound and the reduction's block shape leaves it 2 blocks per CU.) */
int qagnn_bn_relu_bwd_colsum_f32(const float* dR, const float* Hh, float* dH, int32_t ld, int32_t R, int32_t Cc, const float* mean,
                                 const float* invstd, const float* scale, const float* shift, const float* gamma, const float* sum_dy,
                                 const float* sum_dy_hhat, float inv_rows, const float* roww, float* colsum /* [Cc] */, float* workspace,
                                 qagnn_stream_t stream);
int qagnn_sin_basis_f32(const float* score, const float* js, float* out, int32_t ldo, int32_t R, int32_t J, qagnn_stream_t stream);
/* BatchNorm1d batch statistics from the per-tile partials a GEMM left in qagnn_gemm_nn_args.colstat_part, and the bookkeeping of
 * qagnn_bn_finalize_f32, in ONE launch (modeling_qagnn.py:408: BatchNorm1d over all N rows of GATConvE.mlp's first Linear).
 * Tiles are combined with the pairwise update of Chan et al.: mean = sum_t (n_t x0_t + S1_t) / R, M2 = sum_t [S2_t - S1_t^2 / n_t +
 * n_t (mean_t - mean)^2] -- as accurate as the two-pass form (each tile is shifted by one of its own values), in a fixed order.
 * stats: [5][Cc] = mean | biased var | invstd | scale | shift.  Running statistics / batch counter as in qagnn_bn_finalize_f32. */
int qagnn_bn_stats_finalize_f32(const float* part, int32_t n_tiles, int32_t R, int32_t Cc, const float* gamma, const float* beta, float eps,
                                float* stats, float* run_mean, float* run_var, int64_t* num_batches_tracked, const int64_t* dense_pos,
                                int32_t d, float momentum, float unbias, int32_t ones_col, qagnn_stream_t stream);

/* Dropout under hipGraph replay.  Every dropout launch of this library (qagnn_gelu_dropout_*, qagnn_pool_attn_*, the hops) takes
 * its seed by value; a captured graph would replay it verbatim and draw the SAME keep masks in every training step
 * (torch's nn.Dropout solves the same problem with a device-side philox offset).  The kernels therefore add one device-resident
 * word per device, the seed EPOCH (0 until advanced), into the seed.  A captured training step ends with
 * qagnn_seed_epoch_advance(1): replay k uses epoch k, forward and backward of one replay agree, eager callers never notice. */
int qagnn_seed_epoch_advance(uint64_t delta, qagnn_stream_t stream);
int qagnn_seed_epoch_set(uint64_t value, qagnn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * The edge kernels: relation-aware multi-head graph attention over the batched subgraphs.  Replace
 *   PyG propagate/__lift__ gathers (modeling_qagnn.py:442), the concat + per-edge SGEMMs (:464-466), the score
 *   reduction (:469-470), PyG softmax grouped by SOURCE (:472), out-degree scaling (:476-481), msg*alpha (:483)
 *   and the torch_scatter scatter-add by TARGET (:388,442), plus all of their autograd backward.
 * Inputs are the per-NODE projections (head-padded, row pitch ldk = 3*DP: K | M | Q) and the per-CLASS tables
 * (row pitch lde = 2*DP: Ek | Em):
 *   key_e = K[tgt] + Ek[c],  msg_e = M[src] + Em[c],  score_eh = qscale * <Q[src], key_e>_h
 *   a_eh = softmax over the out-edges of src (eps 1e-16),  alpha_eh = deg(src) * a_eh
 *   aggr[tgt] += alpha_eh * msg_e
 * forward writes a[Ep][4], alpha[Ep][4] (source order) and aggr[N][DP]; `score` is scratch [Ep][4].
 * backward takes G = d aggr [N][DP] and writes dKMQ [N][3*DP], dEkEm [C][2*DP]; scratch: ga[Ep][4] (becomes gs),
 * rs[N][4], cls_part[max_chunks + QAGNN_CLS_SLICES*C][2*DP].
 * ------------------------------------------------------------------------------------------------------------ */
int qagnn_edge_attn_fwd_f32(const qagnn_graph* g, const float* KMQ, int32_t ldk, const float* EkEm, int32_t lde, int32_t HP,
                            float qscale, float* score, float* a, float* alpha, float* aggr, int32_t lda,
                            qagnn_stream_t stream);
int qagnn_edge_attn_bwd_f32(const qagnn_graph* g, const float* KMQ, int32_t ldk, const float* EkEm, int32_t lde, int32_t HP,
                            float qscale, const float* a, const float* alpha, const float* G, int32_t ldg, float* dKMQ,
                            float* dEkEm, float* ga, float* rs, float* cls_part, qagnn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * One GATConvE hop of the stack, sequenced natively (csrc/hop.hip).  Replaces, per layer, the reference's
 *   QAGNN_Message_Passing.mp_helper loop body (modeling_qagnn.py:45-50): GATConvE.forward (:411-452) -> message (:455-484)
 *   -> mlp (:443, :408) -> GELU -> dropout, and its autograd backward,
 * by exactly the launches above in a fixed order (results bit-identical to composing the entry points by hand):
 *   fwd:  KMQ = [X | S][Wx ; Ws] + TT[ntype];  (a, alpha, aggr) = edge attention;  h1 = aggr W1^T + b1;  BatchNorm statistics
 *         (batch or running) + bookkeeping;  out = relu(bn(h1)) W2^T + b2;  y = dropout(gelu(out))  (apply_act)
 *   bwd:  the hand-derived backward of the same chain (SURVEY.md 9.2 for the attention part)
 * All operands are in the kernels' packed layout (head-padded, weights pre-transposed; qagnn_amd/modeling_qagnn.py packs
 * them from the reference's state-dict layout).  One struct serves both directions; the forward ignores the gradient fields.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct qagnn_hop_args {
  const qagnn_graph* g;
  int32_t N, DP, SP, HP, T;        /* node rows, padded row width (4*HP), width of S (multiple of 16, may be 0), head pitch, node types */
  float qscale;                    /* 1/sqrt(dim_per_head) */
  const float* X;                  /* [N, DP]   node features entering the hop */
  const float* S;                  /* [N, SP]   layer-invariant score half of node_feature_extra (:86) */
  const int64_t* ntype;            /* [N] */
  const float* Wx_t; const float* Wx;   /* [DP, 3DP], [3DP, DP]  K|M|Q projection of X */
  const float* Ws_t; const float* Ws;   /* [SP, 3DP], [3DP, SP]  ... of S */
  const float* TT;                 /* [T, 3DP]  node-type half of the projection + query bias */
  const float* EkEm;               /* [C, 2DP]  per-class tables Ek | Em */
  const float* W1t; const float* W1; const float* b1;      /* mlp[0]: [DP, DP] x2, [DP] */
  const float* gamma; const float* beta;                   /* mlp[1] affine, head-padded */
  const float* W2t; const float* W2; const float* b2;      /* mlp[3] */
  int32_t batch_stats;             /* 1: BatchNorm uses batch statistics (train mode) */
  float eps;
  const float* run_mean_p; const float* run_var_p;  /* [DP] head-padded running statistics (read when batch_stats == 0) */
  float* run_mean; float* run_var; int64_t* num_batches_tracked;  /* the module's dense buffers, updated in train mode; or NULL */
  const int64_t* dense_pos; int32_t d; float momentum;            /* see qagnn_bn_finalize_f32 */
  int32_t apply_act; float p_drop; uint64_t seed;                 /* y = dropout(gelu(out), p_drop, seed); 0: the hop returns `out` */
  /* forward results, kept by the caller for the backward */
  float* KMQ;                      /* [N, 3DP] */
  float* a; float* alpha;          /* [Ep, 4] each, source order */
  float* aggr; float* h1; float* out; float* y;   /* [N, DP]; y unused when apply_act == 0 */
  float* stats;                    /* [5, DP]: mean, var, invstd, scale, shift */
  /* backward only */
  const float* dy;                 /* [N, DP] */
  float* dX;                       /* [N, DP] or NULL */
  float* dS;                       /* [N, SP] or NULL */
  int32_t accumulate_dS; int32_t accumulate_dX;   /* 1: the gradient is ADDED to what dS / dX hold (running total over the readers) */
  float* dWx_t; float* dWs_t;      /* [DP, 3DP], [SP, 3DP] */
  float* dTT; float* dEkEm;        /* [T, 3DP], [C, 2DP] */
  float* dW1t; float* db1;         /* [DP, DP], [DP] */
  float* dbn;                      /* [2, DP]: d beta, d gamma */
  float* dW2t; float* db2;         /* [DP, DP], [DP] */
  float* ws; int64_t ws_elems;     /* scratch: qagnn_hop_{fwd,bwd}_workspace_elems floats */
  int32_t gemm_split;              /* 1: the NN products run through qagnn_gemm_nn_split_f32 (bf16 matrix cores, exact 3-way split);
                                      2: the same, and the three-MFMA form wherever `amax` (below) makes it possible;
                                      3: REDUCED PRECISION, on request only (never a default): as 2 with ONE fp16 MFMA per product where 2 takes
                                      three (qagnn_gemm_nn_args.pieces = 1, qagnn_gemm_tn_h1_f32) -- the GEMM arithmetic of the reference under
                                      its own --fp16 autocast; statistics, softmax, aggregation and all storage stay fp32 */
  int32_t ones_col;                /* see qagnn_bn_finalize_f32: >= 0 makes db2 = row ones_col of dW2t (no column reduction of d out); a caller
                                      that passes db2 = dW2t + ones_col * DP (and, for tab_col, dTT = dWs_t + tab_col * 3 DP) gets no copy */
  int32_t tab_col;                 /* >= 0: columns [tab_col, tab_col + T) of S hold the node-type indicators (1 at tab_col + ntype[r], S's
                                      zero padding otherwise; the matching rows of Ws_t are zero), so dTT = rows [tab_col, tab_col + T) of
                                      dWs_t = S^T dKMQ: the type-table gradient falls out of the weight-gradient GEMM instead of costing a
                                      grouped column reduction over dKMQ (60 us per layer at 64 000 rows).  -1: reduce dKMQ by node type */
  qagnn_stream_t side_stream;      /* backward only; NULL or a second stream of the same device: the four weight-gradient products of
                                      each hop (and the gradients copied out of their rows) are launched there, forked from `stream`
                                      by events as soon as their operands exist and joined before the call returns -- they feed
                                      nothing downstream and would otherwise sit in the serial data-gradient chain.  Results are
                                      bit-identical either way (same launches).  Capture-safe: the fork makes the side stream part
                                      of a capture in progress on `stream`, the join closes the branch.  In a stack call the field of
                                      the LAST hop is the one that is read */
  uint32_t* amax;                  /* NULL, or QAGNN_HOP_AMAX_WORDS device words that belong to this hop and that the caller keeps, untouched, from
                                      the forward call to the backward call (the library zeroes them at the start of the forward).  With
                                      gemm_split == 2 and batch statistics, the hop's large products then run in the three-MFMA form
                                      (csrc/gemm_nn2.hip): the words hold the bit patterns of max |X|, max |S|, max |aggr|, the bound of
                                      relu(bn(h1)), max |y|, and (backward) max |d out|, max |d h1|, max |d K|M|Q|, each left behind by the
                                      kernel that produces the tensor (X, S of the first hop: one reduction pass each).  In a stack call
                                      whose hops are chained (hops[l + 1].X == hops[l].y, one shared S) the words of X and S are shared too */
  const uint32_t* x_amax;          /* NULL, or a word that already holds max |X| (bit pattern; e.g. left by qagnn_gelu_dropout_fwd_amax_f32 when it
                                      produced X): the hop then skips its own reduction pass over X.  Likewise s_amax for S.  Read-only, read
                                      again by the backward call */
  const uint32_t* s_amax;
} qagnn_hop_args;
#define QAGNN_HOP_AMAX_WORDS 16
int64_t qagnn_hop_fwd_workspace_elems(int32_t N, int32_t Ep, int32_t DP);
int64_t qagnn_hop_bwd_workspace_elems(int32_t N, int32_t Ep, int32_t DP, int32_t SP,
                                      int32_t cls_part_rows /* g->max_chunks + QAGNN_CLS_SLICES * g->C */);
/* (the backward workspace holds TWO sets of the buffers the weight-gradient stream reads -- d out, d h1, d K|M|Q -- so that stream may
 * lag the data-gradient chain by a whole hop; the hops of a stack call share one workspace) */
int qagnn_hop_fwd_f32(const qagnn_hop_args* h, qagnn_stream_t stream);
int qagnn_hop_bwd_f32(const qagnn_hop_args* h, qagnn_stream_t stream);
/* The whole k-hop stack per call (QAGNN_Message_Passing.mp_helper, modeling_qagnn.py:45-50, and its backward): hops[l] is a complete
 * qagnn_hop_args, chained by the caller (hops[l+1].X = hops[l].y; hops[l].dy = hops[l+1].dX; one shared dS with accumulate_dS = 1 on
 * every hop but the last).  Same launches in the same order as k single-hop calls; one FFI crossing for host-bound batches. */
int qagnn_stack_fwd_f32(const qagnn_hop_args* hops, int32_t k, qagnn_stream_t stream);
int qagnn_stack_bwd_f32(const qagnn_hop_args* hops, int32_t k, qagnn_stream_t stream);

/* Per-batch node bookkeeping of QAGNN.forward in one launch.  Replaces ~15 elementwise / reduction kernels of
 *   modeling_qagnn.py:154      concept_ids[:, 1:] - 1            -> ridx [B][n] (int64; -1 on the context node, slot 0)
 *   modeling_qagnn.py:160-167  node-score normalisation          -> score [B][n] (negate, subtract the context node's, mask PAD,
 *                                                                   divide by the mean |.| over the adj_len real nodes + 1e-5)
 *   modeling_qagnn.py:173-177  pooling mask                      -> mask [B][n] (uint8: PAD or context node; slot 0 cleared when
 *                                                                   every slot would be masked)
 * raw_scores [B][n] fp32, adj_len [B], node_type [B][n], concept_ids [B][n] int64 as the reference holds them.
 * table_rows > 0: ids outside [1, table_rows] in the slots 1.. (the reference's nn.Embedding raises on them) become the zero row (-1)
 * and set err[0] = 1 (err may be NULL; the caller zeroes it). */
int qagnn_node_prep_f32(const float* raw_scores, const int64_t* adj_len, const int64_t* node_type, const int64_t* concept_ids,
                        int32_t B, int32_t n, float* score, uint8_t* mask, int64_t* ridx, int64_t table_rows, int32_t* err,
                        qagnn_stream_t stream);

/* ------------------------------------------------------------------------------------------------------------
 * Fused multi-tensor RAdam step.  Replaces the per-parameter Python loop of
 *   utils/optimization_utils.py:31-97  (RAdam.step: ~10 elementwise kernels per tensor, ~70 decoder tensors)
 * p, g, m, v: HOST arrays of n_tensors DEVICE pointers (parameter, gradient, exp_avg, exp_avg_sq; fp32, contiguous, any
 * alignment), numel: host array of element counts.  All tensors share one step count, i.e. one (step_size, mode):
 *   mode 2: N_sma >= 5 (:83-87)   mode 1: SGD-like branch (:89-92)   mode 0: moments only (step_size < 0)
 * Per element:  v = beta2 v + (1-beta2) g g;  m = beta1 m + (1-beta1) g;  p -= weight_decay lr p;  p -= step_size lr m / (sqrt(v)+eps)
 * (or  p -= step_size lr m  in mode 1).  The hyper-parameters are doubles: 1 - beta, weight_decay lr and step_size lr are formed in
 * double and rounded to fp32 once, like the reference's Python scalars.  Pointer tables travel in the kernel arguments: nothing to
 * allocate, capture safe. */
int qagnn_radam_step_f32(int32_t n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v,
                         const int64_t* numel, double beta1, double beta2, double eps, double lr, double weight_decay, double step_size,
                         int32_t mode, qagnn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* QAGNN_HIP_H */
