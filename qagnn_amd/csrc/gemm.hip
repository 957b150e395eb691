// Dense fp32 GEMMs on the CDNA4 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32, 64 FLOP/clk/SIMD).
//
// These replace the cuBLAS SGEMMs the reference launches per EDGE (modeling_qagnn.py:464-466) by per-NODE
// projections (project-then-gather), plus GATConvE.mlp (:443), Vh/Vx (:92) and emb_score (:73),
// and the autograd backward of all of them.
//
// MFMA 16x16x4 f32 operand layout (wave64), from the CDNA4 ISA:
//   A: lane l holds A[i = l & 15][k = l >> 4]       B: lane l holds B[k = l >> 4][j = l & 15]
//   D: lane l, reg r holds D[row = (l >> 4) * 4 + r][col = l & 15]
#include <stdlib.h>

#include "common.h"

namespace qagnn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 16;        // k-tile
constexpr int NN_BM = 128;    // rows per block (4 waves x 2 row tiles of 16)
constexpr int PA_NN = BK + 4; // LDS pitch of the row-major A tile: 16-byte aligned rows (ds_write_b128), 2-way on the 2 A reads per k-step

__host__ __device__ constexpr int pitch_b(int bn) { return (bn % 32 == 16) ? bn : bn + 16; }  // rows k, k+1 land 16 banks apart

// ------------------------------------------------------------------------------------------------------------
// NN:  C[M][No] (+)= [A1|A2] * [B1;B2] + bias + rowtab[rowidx]
// block = WAVES x 64 threads; output tile BM x (NT*16) with BM = WAVES*RT*16: wave w owns row tiles w*RT .. w*RT+RT-1 and
// all NT column tiles.  The block walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... (persistent when the grid is smaller
// than the tile count): the row stores of tile t are fire-and-forget, so they drain to HBM under the MFMAs of tile t+1.
// Two shapes are instantiated: <4 waves, RT=2> (one tile per block, the original mapping) and <8 waves, RT=1> (same
// 128-row tile shared by 8 waves, launched persistent).
// ------------------------------------------------------------------------------------------------------------
template <int V> struct ic { static constexpr int value = V; };  // compile-time buffer index for generic lambdas

// Scheduling shape of one unrolled k-tile (4 k-steps of NM MFMAs each, fragments read from LDS): the fragment reads of
// k-step kk+1 are interleaved 1 : 2 with the MFMAs of k-step kk, so an MFMA never waits on a read issued just before it
// (hipcc's own order is read -> lgkmcnt(0) -> 2 MFMAs, which exposes one LDS latency per MFMA pair).  Speed only: +2-3 % on
// the TN strip kernel (90 registers, room for the fragments in flight); the NN kernel sits at its 128-register cap and loses.
template <int NM>
__device__ __forceinline__ void sched_ktile_pipeline() {
  constexpr int DS = 0x100, MFMA = 0x008, NR = (NM + 1) / 2 + 1;  // ds_read(2) instructions per k-step incl. the A fragment
#pragma unroll
  for (int r = 0; r < NR; ++r) __builtin_amdgcn_sched_group_barrier(DS, 1, 0);
#pragma unroll
  for (int kk = 0; kk < 3; ++kk) {
#pragma unroll
    for (int t = 0; t < NR; ++t) {
      __builtin_amdgcn_sched_group_barrier(MFMA, 2, 0);
      __builtin_amdgcn_sched_group_barrier(DS, 1, 0);
    }
  }
  __builtin_amdgcn_sched_group_barrier(MFMA, 2 * NR, 0);
}

// Register budget: QAGNN_NN_OCC co-resident blocks per CU (LDS allows 2).  The budget is a trade: 2 blocks per CU let one
// block's epilogue stores overlap the other's MFMAs, but cap a wave at 512 / (OCC * WAVES / 4) registers.
#ifndef QAGNN_NN_OCC
#define QAGNN_NN_OCC 2  // 0 = leave it to the compiler (it takes ~300 registers for NT = 13: one block per CU)
#endif
#if QAGNN_NN_OCC > 0
#define QAGNN_NN_ATTR __attribute__((amdgpu_waves_per_eu(QAGNN_NN_OCC * WAVES / 4, QAGNN_NN_OCC * WAVES / 4)))
#else
#define QAGNN_NN_ATTR
#endif

template <int NT, bool AFFINE, int WAVES, int RT>
__global__ __launch_bounds__(WAVES * 64) QAGNN_NN_ATTR void k_gemm_nn(qagnn_gemm_nn_args a, int ntiles) {
  constexpr int NTHR = WAVES * 64, BM = WAVES * RT * 16;
  constexpr int BN = NT * 16;
  constexpr int PB = pitch_b(BN);
  constexpr int B_F4 = BK * BN / 4;                 // float4 per B tile
  constexpr int B_IT = (B_F4 + NTHR - 1) / NTHR;
  // one LDS array: the k-loop tiles (A BM x 20, B 16 x PB) and, afterwards, the epilogue staging slabs (one per wave)
  constexpr int PS = BN + 4;                        // staging pitch: PS % 8 == 4 -> the 4 row groups of a store land 16 banks apart
  constexpr int SLAB_ROWS = WAVES <= 4 ? 16 : 8;    // rows a wave transposes at a time (keeps 8 waves inside 64 KB)
  constexpr int SUB = 16 / SLAB_ROWS;
  constexpr int A_F = BM * PA_NN, B_F = BK * PB, BUF_F = A_F + B_F;
  constexpr int KLOOP_F = 2 * BUF_F, STAGE_F = WAVES * SLAB_ROWS * PS;   // double-buffered k-tiles | epilogue slabs
  __shared__ __attribute__((aligned(16))) float smem[KLOOP_F > STAGE_F ? KLOOP_F : STAGE_F];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ncb = (a.No + BN - 1) / BN;
  const int nk1 = a.K1 / BK, nkt = nk1 + a.K2 / BK;
  const int ar = tid >> 2, ac4 = tid & 3;  // A tile: NTHR/4 rows x 4 float4 per pass, RT passes

  for (int vb = blockIdx.x; vb < ntiles; vb += gridDim.x) {
    // Workgroup b runs on XCD b % 8 (observed; speed only), so tiles are re-indexed with xcd_remap(): each XCD gets a
    // CONTIGUOUS range of tiles, and the column blocks of one row block (consecutive tile ids) read their shared A rows
    // through ONE L2 instead of up to three.  (gridDim.x is a multiple of 8 or equals ntiles, so vb % 8 is the XCD too.)
    const int tile = a.xcd_remap ? xcd_remap(vb, ntiles) : vb;
    const int m0 = (tile / ncb) * BM, n0 = (tile % ncb) * BN;

    f32x4 acc[RT][NT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    float4 raS[2][RT];
    float4 rb[B_IT];
    int64_t arow[RT];  // source row of A1 for this thread's tile rows (gathered or identity); -1 = zero row
#pragma unroll
    for (int p = 0; p < RT; ++p) {
      const int row = m0 + ar + p * (NTHR / 4);
      arow[p] = row < a.M ? (a.a_rowidx ? a.a_rowidx[row] : (int64_t)row) : -1;
    }

    // global -> register staging.  A (the HBM stream: 64-byte row segments, ~2 us under load) is fetched TWO k-tiles ahead
    // into alternating register sets; B (weights, L2-resident) one k-tile ahead.  B is issued before the younger A loads so
    // that the in-order vmcnt wait in front of the LDS store leaves exactly the A loads of tile kt+2 in flight.
    // Every load is unconditional (addresses clamped into the operand; rows >= M and columns >= No only ever feed
    // accumulator rows / columns that are never stored): with branches around the loads hipcc falls back to vmcnt(0).
    auto gloadA = [&](int kt, float4 (&dst)[RT]) {
      const bool first = kt < nk1;
      const float* A = first ? a.A1 : a.A2;
      const int lda = first ? a.lda1 : a.lda2;
      const int k0 = (first ? kt : kt - nk1) * BK;
#pragma unroll
      for (int p = 0; p < RT; ++p) {
        const int64_t srow = first ? (arow[p] >= 0 ? arow[p] : 0) : (int64_t)min(m0 + ar + p * (NTHR / 4), a.M - 1);
        dst[p] = ld4(A + srow * lda + k0 + ac4 * 4);
      }
    };
    auto gloadB = [&](int kt) {
      const bool first = kt < nk1;
      const float* B = first ? a.B1 : a.B2;
      const int ldb = first ? a.ldb1 : a.ldb2;
      const int k0 = (first ? kt : kt - nk1) * BK;
#pragma unroll
      for (int it = 0; it < B_IT; ++it) {
        const int idx = min(tid + it * NTHR, B_F4 - 1);
        const int kr = idx / (BN / 4), c4 = idx % (BN / 4);
        rb[it] = ld4(B + (int64_t)(k0 + kr) * ldb + min(n0 + c4 * 4, a.No - 4));
      }
    };
    auto lstore = [&](auto bufc, int kt, const float4 (&src)[RT]) {  // tile kt -> LDS buffer buf (BN affine + ReLU applied here)
      constexpr int buf = decltype(bufc)::value;
      float* As = smem + buf * BUF_F;
      float* Bs = As + A_F;
#pragma unroll
      for (int p = 0; p < RT; ++p) {
        float4 v = src[p];
        if (AFFINE && kt < nk1) {
          const int k0 = kt * BK;
          const float4 sc = ld4(a.a_scale + k0 + ac4 * 4), sh = ld4(a.a_shift + k0 + ac4 * 4);
          v.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f);
          v.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
          v.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f);
          v.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
        }
        if (a.a_rowidx && kt < nk1 && arow[p] < 0) v = make_float4(0.f, 0.f, 0.f, 0.f);  // gathered "-1" rows are zero rows
        st4(As + (ar + p * (NTHR / 4)) * PA_NN + ac4 * 4, v);
      }
#pragma unroll
      for (int it = 0; it < B_IT; ++it) {
        const int idx = tid + it * NTHR;
        if (idx < B_F4) {
          const int kr = idx / (BN / 4), c4 = idx % (BN / 4);
          st4(Bs + kr * PB + c4 * 4, rb[it]);
        }
      }
    };
    const float* const Aw = smem + (w * RT * 16 + (lane & 15)) * PA_NN + (lane >> 4);
    const float* const Bw = smem + A_F + (lane >> 4) * PB + (lane & 15);
    auto ktile = [&](auto curc, int kt) {  // cur = kt & 1 names both the LDS buffer of tile kt and the A register set of tile kt+2
      constexpr int cur = decltype(curc)::value;
      gloadB(min(kt + 1, nkt - 1));               // past the last tile: a redundant reload, never consumed
      gloadA(min(kt + 2, nkt - 1), raS[cur]);
      __builtin_amdgcn_sched_barrier(0);  // keep the loads up here: the scheduler otherwise sinks them next to the LDS store
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        float av[RT];
#pragma unroll
        for (int i = 0; i < RT; ++i) av[i] = Aw[cur * BUF_F + i * 16 * PA_NN + kk * 4];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float bv = Bw[cur * BUF_F + kk * 4 * PB + j * 16];
#pragma unroll
          for (int i = 0; i < RT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv, acc[i][j], 0, 0, 0);
        }
      }
      lstore(ic<cur ^ 1>{}, min(kt + 1, nkt - 1), raS[cur ^ 1]);  // after the MFMAs: the loads had one k-tile (B) / two (A) to land
      __syncthreads();
    };

    // double-buffered k-loop, ONE barrier per k-tile (it publishes tile kt+1 and retires everybody's reads of tile kt)
    gloadA(0, raS[0]);
    gloadB(0);
    gloadA(min(1, nkt - 1), raS[1]);
    __syncthreads();  // the previous output tile's slab reads are done before the k-loop buffers are overwritten
    lstore(ic<0>{}, 0, raS[0]);
    __syncthreads();
    for (int kt = 0; kt < nkt; kt += 2) {
      ktile(ic<0>{}, kt);
      if (kt + 1 < nkt) ktile(ic<1>{}, kt + 1);
    }

    // epilogue.  The MFMA layout gives a lane ONE column of 4 rows per accumulator; storing that directly is 104 dword
    // stores per lane in 64-byte row fragments (store-issue bound).  Instead each wave transposes SLAB_ROWS rows at a
    // time through its private LDS slab and writes whole rows with 16-byte lanes: 4x fewer store instructions, full
    // 128-byte lines.
    float* const St = smem + w * SLAB_ROWS * PS;
    constexpr int ROW_F4 = BN / 4, TILE_F4 = SLAB_ROWS * ROW_F4, ST_IT = (TILE_F4 + 63) / 64;
#pragma unroll
    for (int i = 0; i < RT; ++i) {
#pragma unroll
      for (int h = 0; h < SUB; ++h) {
        __syncthreads();  // k-loop reads (first pass) / the previous pass's slab reads are done before the slab is overwritten
        const int lr0 = (lane >> 4) * 4 - h * SLAB_ROWS;  // this lane's first row inside the slab (in range for its half only)
        if (lr0 >= 0 && lr0 < SLAB_ROWS) {
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) St[(lr0 + r) * PS + j * 16 + (lane & 15)] = acc[i][j][r];
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ST_IT; ++it) {
          const int idx = lane + it * 64;
          if (idx >= TILE_F4) break;
          const int lr = idx / ROW_F4, c4 = idx % ROW_F4;
          const int row = m0 + (w * RT + i) * 16 + h * SLAB_ROWS + lr, col = n0 + c4 * 4;
          if (row >= a.M || col >= a.No) continue;
          float4 v = ld4(St + lr * PS + c4 * 4);
          if (a.bias) v = add4(v, ld4(a.bias + col));
          if (a.rowtab) v = add4(v, ld4(a.rowtab + (int64_t)a.rowidx[row] * a.ldt + col));
          float* dst = a.C + (int64_t)row * a.ldc + col;
          if (a.accumulate) v = add4(v, ld4(dst));
          st4(dst, v);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// TN:  P[chunk][Ka][No] = sum over the chunk's rows of A[r][ka] * B[r][no]; a second kernel sums the chunks in order.
// One WAVE per 16-row tile of the output: the block has NW = blockDim.x / 64 waves and owns BM = 16*NW output rows x
// NT*16 columns, so the host picks NW to tile Ka exactly (Ka = 208 -> 13 waves, 112 -> 7, 1024 -> 16): no MFMA is
// spent on tile padding (the previous fixed 64-row tile wasted 19 % on 208).  With BM = 16*NW the A k-tile is exactly
// one float4 per thread.  k-tiles of 16 rows, double-buffered, one barrier per tile.
// ------------------------------------------------------------------------------------------------------------
constexpr int TN_RC = 256;  // minimum rows per chunk (and the chunking the workspace query assumes): >= 1 block per CU for the
                            // 208 x 208 gradients at N = 64 000; launches with several tiles per chunk use longer chunks
constexpr int TN_RC_SMALL = 64;  // ... and for reductions over <= 4096 rows (class tables, C = 612): the k-loop of a chunk is serial
__host__ __device__ constexpr int tn_min_chunk(int R) { return R <= 4096 ? TN_RC_SMALL : TN_RC; }
// the bf16-split weight-gradient kernel over <= 4096 rows (10 subgraphs = 2 000 node rows): one 32-row k-tile per chunk puts twice the
// blocks on the idle chip (2 000 x 208 x 208: 126 instead of 64) and halves each block's serial work (A/B against 64-row chunks:
// profiles/r3_run18_small_batch_ab.txt).  The workspace query sizes for 32-row chunks there.
constexpr int TN_RC_SPLIT_SMALL = 32;
static int tn_split_min_chunk(int R) { return R > 4096 ? TN_RC : TN_RC_SPLIT_SMALL; }

__host__ __device__ constexpr int pitch16(int w) { return (w % 32 == 16) ? w : w + 16; }  // rows k, k+1 land 16 banks apart

template <int NT, bool AFFINE>
__global__ __launch_bounds__(1024) void k_gemm_tn(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                  float* __restrict__ P, int R, int Ka, int No, const float* __restrict__ a_scale,
                                                  const float* __restrict__ a_shift, const int64_t* __restrict__ a_rowidx,
                                                  float* __restrict__ Pcs, const int64_t* __restrict__ b_rowidx, int groups,
                                                  int chunk_rows) {
  constexpr int BN = NT * 16;
  constexpr int PB = pitch_b(BN);
  constexpr int B_F4 = BK * BN / 4;
  constexpr int B_IT = (NT + 3) / 4;  // float4 of the B tile per thread when the block has >= 4 waves (fewer waves: loop below)
  extern __shared__ __attribute__((aligned(16))) float smem_tn[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, nthr = blockDim.x, BM = (nthr >> 6) * 16;
  const int PA = pitch16(BM), TA_F = BK * PA, TBUF_F = TA_F + BK * PB + BK;  // + 16 group ids of the tile's rows
  // optional by-product: column sums of B (per row group) taken from the B tiles already sitting in LDS
  const bool do_cs = Pcs != nullptr && blockIdx.y == 0 && tid < BN;
  float cs0 = 0.f, cs1 = 0.f, cs2 = 0.f, cs3 = 0.f;
  int rg = 0;  // group id of tile row `tid` (threads 0..15), staged through LDS with the tile
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM, chunk = blockIdx.z;
  const int r_beg = chunk * chunk_rows, r_end = min(R, r_beg + chunk_rows);
  const int nkt = (r_end - r_beg + BK - 1) / BK;

  f32x4 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 ra, rb[B_IT];
  const int a_f4 = BM / 4;                       // float4 per A tile row; 16 * a_f4 == nthr
  const int akr = tid / a_f4, ac4 = tid % a_f4;  // this thread's (k row, float4 column) of the A tile
  const int acol = m0 + ac4 * 4;
  float4 a_sc = make_float4(0.f, 0.f, 0.f, 0.f), a_sh = a_sc;  // the thread's A columns never change: load the BN affine once
  if (AFFINE && acol < Ka) { a_sc = ld4(a_scale + acol); a_sh = ld4(a_shift + acol); }

  auto gload = [&](int kt) {
    const int r0 = r_beg + kt * BK;
    {
      const int row = r0 + akr;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int64_t srow = (row < r_end && acol < Ka) ? (a_rowidx ? a_rowidx[row] : (int64_t)row) : -1;
      if (srow >= 0) {
        v = ld4(A + srow * lda + acol);
        if (AFFINE) {
          v.x = fmaxf(fmaf(v.x, a_sc.x, a_sh.x), 0.f);
          v.y = fmaxf(fmaf(v.y, a_sc.y, a_sh.y), 0.f);
          v.z = fmaxf(fmaf(v.z, a_sc.z, a_sh.z), 0.f);
          v.w = fmaxf(fmaf(v.w, a_sc.w, a_sh.w), 0.f);
        }
      }
      ra = v;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int idx = tid + it * nthr;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < B_F4) {
        const int kr = idx / (BN / 4), c4 = idx % (BN / 4);
        const int row = r0 + kr, col = n0 + c4 * 4;
        if (row < r_end && col < No) v = ld4(B + (int64_t)row * ldb + col);
      }
      rb[it] = v;
    }
    if (Pcs && b_rowidx && tid < BK) rg = (r0 + tid < r_end) ? (int)b_rowidx[r0 + tid] : 0;
  };
  auto lstore = [&](int buf) {
    float* As = smem_tn + buf * TBUF_F;
    float* Bs = As + TA_F;
    if (Pcs && tid < BK) reinterpret_cast<int*>(Bs + BK * PB)[tid] = rg;
    st4(As + akr * PA + ac4 * 4, ra);
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int idx = tid + it * nthr;
      if (idx < B_F4) {
        const int kr = idx / (BN / 4), c4 = idx % (BN / 4);
        st4(Bs + kr * PB + c4 * 4, rb[it]);
      }
    }
  };
  auto mma = [&](int buf, int kk) {
    const float* Aw = smem_tn + buf * TBUF_F + (lane >> 4) * PA + w * 16 + (lane & 15);
    const float* Bw = smem_tn + buf * TBUF_F + TA_F + (lane >> 4) * PB + (lane & 15);
    const float av = Aw[kk * 4 * PA];
#pragma unroll
    for (int j = 0; j < NT; ++j)
      acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Bw[kk * 4 * PB + j * 16], acc[j], 0, 0, 0);
  };
  auto colsum_tile = [&](int buf) {  // thread `tid` owns column tid of the B tile; rows in tile order = row order
    const float* Bs = smem_tn + buf * TBUF_F + TA_F;
    const int* grp = reinterpret_cast<const int*>(Bs + BK * PB);
#pragma unroll
    for (int k = 0; k < BK; ++k) {
      const float v = Bs[k * PB + tid];
      const int g = groups > 1 ? grp[k] : 0;
      if (g == 0) cs0 += v;
      else if (g == 1) cs1 += v;
      else if (g == 2) cs2 += v;
      else cs3 += v;
    }
  };

  if (nkt > 0) {
    gload(0);
    lstore(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < nkt;
    if (more) gload(kt + 1);
    mma(cur, 0);
    mma(cur, 1);
    if (do_cs) colsum_tile(cur);
    mma(cur, 2);
    mma(cur, 3);
    if (more) lstore(cur ^ 1);  // as late as possible: the global loads get the whole tile's MFMA time to land
    __syncthreads();
  }
  if (do_cs && n0 + tid < No) {
    float* pc = Pcs + (int64_t)chunk * groups * No + n0 + tid;
    pc[0] = cs0;
    if (groups > 1) pc[No] = cs1;
    if (groups > 2) pc[2 * No] = cs2;
    if (groups > 3) pc[3 * No] = cs3;
  }
  float* Pc = P + (int64_t)chunk * Ka * No;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = m0 + w * 16 + (lane >> 4) * 4 + r;
    if (row >= Ka) continue;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = n0 + j * 16 + (lane & 15);
      if (col < No) Pc[(int64_t)row * No + col] = acc[j][r];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// TN, compile-time strip shape: the k_gemm_tn mapping (wave w owns output rows 16w .. 16w+15 and all NT column tiles)
// with the wave count NWT fixed at compile time and the two LDS buffers addressed by constants (k-loop unrolled by two),
// so every LDS fragment read is `base VGPR + immediate` and the k-loop carries no address arithmetic: the run-time
// version spends 2-4 VALU instructions per MFMA on addresses and sits at ~48 % MFMA-busy.
// ------------------------------------------------------------------------------------------------------------
template <int NT, int NWT, bool AFFINE, bool COLSUM>
__global__ __launch_bounds__(NWT * 64) void k_gemm_tn_strip(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                            float* __restrict__ P, int R, int Ka, int No,
                                                            const float* __restrict__ a_scale, const float* __restrict__ a_shift,
                                                            const int64_t* __restrict__ a_rowidx, int chunk_rows,
                                                            float* __restrict__ Pcs, const int64_t* __restrict__ b_rowidx, int groups) {
  constexpr int NTHR = NWT * 64, BM = NWT * 16;
  constexpr int BN = NT * 16;
  constexpr int PA = pitch16(BM), PB = pitch_b(BN);
  constexpr int TA_F = BK * PA, TBUF_F = TA_F + BK * PB;
  constexpr int B_F4 = BK * BN / 4, B_IT = (B_F4 + NTHR - 1) / NTHR;
  constexpr int A_F4 = BM / 4;  // float4 per A tile row; 16 * A_F4 == NTHR: one float4 per thread
  __shared__ __attribute__((aligned(16))) float smem[2 * TBUF_F];
  __shared__ int grp_s[2][BK];  // COLSUM with row groups: group id of the tile's 16 rows
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM, chunk = blockIdx.z;
  const int r_beg = chunk * chunk_rows, r_end = min(R, r_beg + chunk_rows);
  const int nkt = (r_end - r_beg + BK - 1) / BK;

  f32x4 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 ra, rb[B_IT];
  const int akr = tid / A_F4, ac4 = tid % A_F4;
  const int acol = m0 + ac4 * 4;
  // optional by-product (first row block only): column sums of B per row group, taken from the B tiles sitting in LDS --
  // thread `tid` < BN owns column tid; saves the separate pass over B that a bias / type-table gradient would need
  const bool do_cs = COLSUM && blockIdx.y == 0 && tid < BN;
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  int rg = 0;

  auto gload = [&](int kt) {
    const int r0 = r_beg + kt * BK;
    {
      const int row = r0 + akr;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int64_t srow = (row < r_end && acol < Ka) ? (a_rowidx ? a_rowidx[row] : (int64_t)row) : -1;
      if (srow >= 0) {
        v = ld4(A + srow * lda + acol);
        if (AFFINE) {  // BN affine of these 4 columns: re-read per tile (L1 hit) rather than held in 8 registers
          const float4 a_sc = ld4(a_scale + acol), a_sh = ld4(a_shift + acol);
          v.x = fmaxf(fmaf(v.x, a_sc.x, a_sh.x), 0.f);
          v.y = fmaxf(fmaf(v.y, a_sc.y, a_sh.y), 0.f);
          v.z = fmaxf(fmaf(v.z, a_sc.z, a_sh.z), 0.f);
          v.w = fmaxf(fmaf(v.w, a_sc.w, a_sh.w), 0.f);
        }
      }
      ra = v;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int idx = tid + it * NTHR;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < B_F4) {
        const int kr = idx / (BN / 4), c4 = idx % (BN / 4);
        const int row = r0 + kr, col = n0 + c4 * 4;
        if (row < r_end && col < No) v = ld4(B + (int64_t)row * ldb + col);
      }
      rb[it] = v;
    }
    if (COLSUM && b_rowidx && tid < BK) rg = (r0 + tid < r_end) ? (int)b_rowidx[r0 + tid] : 0;
  };
  float* const a_dst = smem + akr * PA + ac4 * 4;
  auto lstore = [&](auto bufc) {
    constexpr int buf = decltype(bufc)::value;
    if (COLSUM && tid < BK) grp_s[buf][tid] = rg;
    st4(a_dst + buf * TBUF_F, ra);
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int idx = tid + it * NTHR;
      if (idx < B_F4) {
        const int kr = idx / (BN / 4), c4 = idx % (BN / 4);
        st4(smem + buf * TBUF_F + TA_F + kr * PB + c4 * 4, rb[it]);
      }
    }
  };
  const float* const a_frag = smem + (lane >> 4) * PA + w * 16 + (lane & 15);
  const float* const b_frag = smem + TA_F + (lane >> 4) * PB + (lane & 15);
  auto ktile = [&](auto curc, int kt) {
    constexpr int cur = decltype(curc)::value;
    const bool more = kt + 1 < nkt;
    if (more) gload(kt + 1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const float av = a_frag[cur * TBUF_F + kk * 4 * PA];
      float bv[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) bv[j] = b_frag[cur * TBUF_F + kk * 4 * PB + j * 16];
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[j], acc[j], 0, 0, 0);
    }
    sched_ktile_pipeline<NT>();
    if (do_cs) {  // rows in tile order = row order: the summation order is fixed
      const float* Bs = smem + cur * TBUF_F + TA_F + tid;
#pragma unroll
      for (int kq = 0; kq < BK; ++kq) {
        const float v = Bs[kq * PB];
        const int g = groups > 1 ? grp_s[cur][kq] : 0;
        if (g == 0) cs[0] += v;
        else if (g == 1) cs[1] += v;
        else if (g == 2) cs[2] += v;
        else cs[3] += v;
      }
    }
    if (more) lstore(ic<cur ^ 1>{});  // as late as possible: the global loads get the whole tile's MFMA time to land
    __syncthreads();
  };

  if (nkt > 0) {
    gload(0);
    lstore(ic<0>{});
  }
  __syncthreads();
  for (int kt = 0; kt < nkt; kt += 2) {
    ktile(ic<0>{}, kt);
    if (kt + 1 < nkt) ktile(ic<1>{}, kt + 1);
  }
  if (do_cs && n0 + tid < No) {
    float* pc = Pcs + (int64_t)chunk * groups * No + n0 + tid;
    pc[0] = cs[0];
    if (groups > 1) pc[No] = cs[1];
    if (groups > 2) pc[2 * No] = cs[2];
    if (groups > 3) pc[3 * No] = cs[3];
  }
  float* Pc = P + (int64_t)chunk * Ka * No;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = m0 + w * 16 + (lane >> 4) * 4 + r;
    if (row >= Ka) continue;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = n0 + j * 16 + (lane & 15);
      if (col < No) Pc[(int64_t)row * No + col] = acc[j][r];
    }
  }
}

__global__ void k_sum_chunks(const float* __restrict__ P, float* __restrict__ C, int ldc, int Ka, int No, int nchunks,
                             int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)Ka * No) return;
  const int row = (int)(i / No), col = (int)(i % No);
  float s = 0.f;
#pragma unroll 8
  for (int c = 0; c < nchunks; ++c) s += P[(int64_t)c * Ka * No + i];
  float* d = C + (int64_t)row * ldc + col;
  *d = accumulate ? *d + s : s;
}

// Same reduction, latency-shaped: a thread owns one float4 column, the block's 16 waves split the chunks (wave g takes chunks g,
// g+16, ...) and every wave issues FOUR of its loads back to back before it adds anything, so the ~84 partials of a 64 000-row
// weight gradient are fetched in two round trips per thread instead of five (the partials were written by the kernel in front and
// sit in L2 / Infinity Cache: the kernel is bound by load latency, not by bytes -- it ran at 1.5 TB/s with 2 loads in flight
// over 8 waves: profiles/r2_run60_by_shape.txt).  The 16 partial sums meet in LDS and are added in wave order.  The summation
// order is fixed by (nchunks), never by timing.
constexpr int SC_G = 16;
__global__ __launch_bounds__(SC_G * 64) void k_sum_chunks4(const float* __restrict__ P, float* __restrict__ C, int ldc, int Ka, int No,
                                                           int nchunks, int accumulate) {
  __shared__ float4 part[SC_G][64];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t n4 = (int64_t)Ka * No / 4, i4 = (int64_t)blockIdx.x * 64 + lane;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i4 < n4) {
    const float4* src = reinterpret_cast<const float4*>(P) + i4;
    for (int c0 = g; c0 < nchunks; c0 += 4 * SC_G) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = src[(int64_t)min(c0 + u * SC_G, nchunks - 1) * n4];  // clamped: unconditional, in flight together
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (c0 + u * SC_G < nchunks) s = add4(s, v[u]);
    }
  }
  part[g][lane] = s;
  __syncthreads();
  if (g == 0 && i4 < n4) {
    float4 t = part[0][lane];
#pragma unroll
    for (int k = 1; k < SC_G; ++k) t = add4(t, part[k][lane]);
    const int64_t e = i4 * 4;
    float* d = C + (e / No) * ldc + (e % No);
    if (accumulate) t = add4(t, ld4(d));
    st4(d, t);
  }
}

// NN launch shape: 8-wave blocks walking the tiles persistently, QAGNN_NN_OCC (= 2) blocks per CU.  Measured at M = 64000
// (profiles/r1_gemm_micro.txt): 10-18 % faster than one 128-row tile per 4-wave block; the other forms were removed in round 5.
static int num_cus() {
  static int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
      v = 256;
    return v;
  }();
  return n;
}

template <int NT>
static int launch_nn(const qagnn_gemm_nn_args& a, hipStream_t stream) {
  // persistent 8-wave blocks, one per CU (QAGNN_NN_OCC per CU in the micro-benchmark builds), XCD-aware tile order: the forms these
  // replaced (4-wave two-row-tile blocks, one block per tile, launch-order tiles) were A/B-ed in profiles/r1_run23_nn_xcd_ab.txt, r1_run32_gemm_ab.txt
  constexpr int per_cu = QAGNN_NN_OCC > 0 ? QAGNN_NN_OCC : 1;
  qagnn_gemm_nn_args b = a;
  b.xcd_remap = 1;
  const int ntiles = cdiv(a.No, NT * 16) * cdiv(a.M, NN_BM);
  const int cap = (num_cus() * per_cu) & ~7;  // multiple of 8: block -> XCD mapping survives the tile walk
  const int grid = ntiles < cap ? ntiles : cap;
  if (a.a_scale) k_gemm_nn<NT, true, 8, 1><<<grid, 512, 0, stream>>>(b, ntiles);
  else k_gemm_nn<NT, false, 8, 1><<<grid, 512, 0, stream>>>(b, ntiles);
  QAGNN_LAUNCH_CHECK("k_gemm_nn");
  return QAGNN_OK;
}

// waves per block = 16-row output tiles per block: the count in [4, 16] that wastes the fewest rows of Ka (ties -> more waves)
static int pick_tn_waves(int Ka) {
  int best = 4, best_waste = INT32_MAX;
  for (int nw = 16; nw >= 4; --nw) {
    const int bm = nw * 16, waste = cdiv(Ka, bm) * bm - Ka;
    if (waste < best_waste) { best_waste = waste; best = nw; }
  }
  return best;
}

template <int NT>
static int launch_tn(const float* A, int lda, const float* B, int ldb, float* P, int R, int Ka, int No, const float* sc,
                     const float* sh, const int64_t* ridx, int chunk_rows, float* Pcs, const int64_t* bidx, int groups,
                     hipStream_t stream) {
  const int nw = pick_tn_waves(Ka), bm = nw * 16, nchunks = cdiv(R, chunk_rows);
  // B tile loop covers B_IT * nthreads float4: needs (NT + 3) / 4 * nw * 64 >= 16 * NT * 4  <=>  nw >= 4  (guaranteed)
  dim3 grid(cdiv(No, NT * 16), cdiv(Ka, bm), nchunks);
  const size_t lds = 2 * (size_t)(BK * pitch16(bm) + BK * pitch_b(NT * 16) + BK) * sizeof(float);
  if (sc) k_gemm_tn<NT, true><<<grid, nw * 64, lds, stream>>>(A, lda, B, ldb, P, R, Ka, No, sc, sh, ridx, Pcs, bidx, groups, chunk_rows);
  else k_gemm_tn<NT, false><<<grid, nw * 64, lds, stream>>>(A, lda, B, ldb, P, R, Ka, No, sc, sh, ridx, Pcs, bidx, groups, chunk_rows);
  QAGNN_LAUNCH_CHECK("k_gemm_tn");
  return QAGNN_OK;
}

static bool tn_strip_enabled() { return true; }  // (the run-time-shaped k_gemm_tn serves the widths the strip kernel is not compiled for)

// compile-time strip launch (NT = 13 and 7 / 13 / 16 waves: every weight gradient of the stack at d = 200)
template <int NWT>
static void launch_tn_strip_i(dim3 grid, hipStream_t stream, const float* A, int lda, const float* B, int ldb, float* P, int R, int Ka, int No,
                              const float* sc, const float* sh, const int64_t* ridx, int chunk_rows, float* Pcs, const int64_t* bidx,
                              int groups) {
#define QAGNN_STRIP_GO(AFF, CS) \
  k_gemm_tn_strip<13, NWT, AFF, CS><<<grid, NWT * 64, 0, stream>>>(A, lda, B, ldb, P, R, Ka, No, sc, sh, ridx, chunk_rows, Pcs, bidx, groups)
  if (Pcs) {
    if (sc) QAGNN_STRIP_GO(true, true);
    else QAGNN_STRIP_GO(false, true);
  } else {
    if (sc) QAGNN_STRIP_GO(true, false);
    else QAGNN_STRIP_GO(false, false);
  }
#undef QAGNN_STRIP_GO
}
static bool tn_strip_ok(int Ka) { const int nw = pick_tn_waves(Ka); return nw == 7 || nw == 13 || nw == 16; }
static int launch_tn_strip(const float* A, int lda, const float* B, int ldb, float* P, int R, int Ka, int No, const float* sc,
                           const float* sh, const int64_t* ridx, int chunk_rows, float* Pcs, const int64_t* bidx, int groups,
                           hipStream_t stream) {
  const int nw = pick_tn_waves(Ka);
  dim3 grid(cdiv(No, 13 * 16), cdiv(Ka, nw * 16), cdiv(R, chunk_rows));
  if (nw == 7) launch_tn_strip_i<7>(grid, stream, A, lda, B, ldb, P, R, Ka, No, sc, sh, ridx, chunk_rows, Pcs, bidx, groups);
  else if (nw == 13) launch_tn_strip_i<13>(grid, stream, A, lda, B, ldb, P, R, Ka, No, sc, sh, ridx, chunk_rows, Pcs, bidx, groups);
  else launch_tn_strip_i<16>(grid, stream, A, lda, B, ldb, P, R, Ka, No, sc, sh, ridx, chunk_rows, Pcs, bidx, groups);
  QAGNN_LAUNCH_CHECK("k_gemm_tn_strip");
  return QAGNN_OK;
}

// Rows per split-K chunk.  Every chunk costs one Ka x No partial (written, then re-read by k_sum_chunks), so a launch whose
// chunk already spans several blocks (column blocks x row blocks) takes longer chunks: just enough blocks to fill the CUs
// once (twice for blocks of <= 8 waves).  Never below TN_RC, which is what qagnn_gemm_tn_workspace_elems() sizes for.
static int pick_tn_chunk_rows(int R, int Ka, int No, int nt) {
  const int lo = tn_min_chunk(R);
  const int rb = pick_tn_waves(Ka), nw = rb;
  const int blocks_per_chunk = cdiv(No, nt * 16) * cdiv(Ka, rb * 16);
  const int target = (num_cus() * (nw <= 8 ? 2 : 1)) / blocks_per_chunk;
  const int rows = (cdiv(R, target > 0 ? target : 1) + 15) & ~15;
  return rows > lo ? rows : lo;
}

// column-tile count per block: the widest instantiation that divides No, else the one wasting the least
static int pick_nt(int No) {
  const int cands[5] = {13, 7, 8, 4, 2};
  for (int c : cands)
    if (No % (c * 16) == 0) return c;
  int best = 4;
  int64_t best_cost = INT64_MAX;
  for (int c : cands) {
    const int64_t cost = (int64_t)cdiv(No, c * 16) * c * 16;
    if (cost < best_cost) { best_cost = cost; best = c; }
  }
  return best;
}

}  // namespace qagnn

using namespace qagnn;

extern "C" int qagnn_gemm_nn_f32(const qagnn_gemm_nn_args* a, qagnn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TimedScope timed(0, stream);
  QAGNN_REQUIRE(a && a->A1 && a->B1 && a->C, QAGNN_EINVAL, "gemm_nn: null pointer");
  QAGNN_REQUIRE(a->M > 0 && a->No > 0 && a->K1 > 0, QAGNN_EINVAL, "gemm_nn: bad sizes M=%d No=%d K1=%d", a->M, a->No, a->K1);
  QAGNN_REQUIRE(a->K1 % BK == 0 && a->K2 % BK == 0 && a->K2 >= 0, QAGNN_EINVAL, "gemm_nn: K1=%d K2=%d must be multiples of %d",
                a->K1, a->K2, BK);
  QAGNN_REQUIRE(a->No % 4 == 0, QAGNN_EINVAL, "gemm_nn: No=%d must be a multiple of 4", a->No);
  QAGNN_REQUIRE(a->lda1 % 4 == 0 && a->ldb1 % 4 == 0 && aligned16(a->A1) && aligned16(a->B1), QAGNN_EINVAL,
                "gemm_nn: operand 1 must be 16-byte aligned with pitches multiple of 4");
  QAGNN_REQUIRE(a->K2 == 0 || (a->A2 && a->B2 && a->lda2 % 4 == 0 && a->ldb2 % 4 == 0 && aligned16(a->A2) && aligned16(a->B2)),
                QAGNN_EINVAL, "gemm_nn: operand 2 must be 16-byte aligned with pitches multiple of 4");
  QAGNN_REQUIRE(!a->rowtab || a->rowidx, QAGNN_EINVAL, "gemm_nn: rowtab without rowidx");
  QAGNN_REQUIRE(a->ldc % 4 == 0 && aligned16(a->C) && (!a->bias || aligned16(a->bias)) &&
                    (!a->rowtab || (aligned16(a->rowtab) && a->ldt % 4 == 0)),
                QAGNN_EINVAL, "gemm_nn: C / bias / rowtab must be 16-byte aligned with pitches multiple of 4");
  QAGNN_REQUIRE(!a->a_scale || (a->a_shift && aligned16(a->a_scale) && aligned16(a->a_shift)), QAGNN_EINVAL,
                "gemm_nn: a_scale/a_shift must both be given and 16-byte aligned");
  switch (pick_nt(a->No)) {
    case 13: return launch_nn<13>(*a, stream);
    case 8: return launch_nn<8>(*a, stream);
    case 7: return launch_nn<7>(*a, stream);
    case 4: return launch_nn<4>(*a, stream);
    default: return launch_nn<2>(*a, stream);
  }
}

extern "C" int64_t qagnn_gemm_tn_workspace_elems(int32_t R, int32_t Ka, int32_t No) {
  return (int64_t)cdiv(R, R <= 4096 ? TN_RC_SPLIT_SMALL : tn_min_chunk(R)) * ((int64_t)Ka * No + 4 * (int64_t)No);
}

extern "C" int qagnn_gemm_tn_colsum_f32(const float* A, int32_t lda, const float* B, int32_t ldb, float* C, int32_t ldc, int32_t R,
                                        int32_t Ka, int32_t No, const float* a_scale, const float* a_shift, const int64_t* a_rowidx,
                                        int32_t accumulate, float* bsum, const int64_t* b_rowidx, int32_t groups, float* workspace,
                                        qagnn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TimedScope timed(1, stream);
  QAGNN_REQUIRE(A && B && C && workspace, QAGNN_EINVAL, "gemm_tn: null pointer");
  QAGNN_REQUIRE(R > 0 && Ka > 0 && No > 0 && Ka % 4 == 0 && No % 4 == 0, QAGNN_EINVAL,
                "gemm_tn: bad sizes R=%d Ka=%d No=%d (Ka, No multiples of 4)", R, Ka, No);
  QAGNN_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && aligned16(A) && aligned16(B), QAGNN_EINVAL,
                "gemm_tn: operands must be 16-byte aligned with pitches multiple of 4");
  QAGNN_REQUIRE(!a_scale || (a_shift && aligned16(a_scale) && aligned16(a_shift)), QAGNN_EINVAL,
                "gemm_tn: a_scale/a_shift must both be given and 16-byte aligned");
  QAGNN_REQUIRE(!bsum || (groups >= 1 && groups <= 4 && (groups == 1 || b_rowidx)), QAGNN_EINVAL, "gemm_tn: colsum groups=%d (1..4)", groups);
  const int nt = pick_nt(No);
  const bool split = !bsum && tn_split_ok(R, Ka, No, lda, ldb, a_rowidx != nullptr, a_scale != nullptr);
  const bool strip = nt == 13 && tn_strip_enabled() && tn_strip_ok(Ka);
  const int crows = split ? tn_split_chunk_rows(R, Ka, No, tn_split_min_chunk(R)) : pick_tn_chunk_rows(R, Ka, No, nt);
  const int nchunks = cdiv(R, crows);
  float* Pcs = bsum ? workspace + (int64_t)nchunks * Ka * No : nullptr;
  int rc;
  if (split) rc = launch_tn_split(A, lda, B, ldb, workspace, R, Ka, No, a_scale, a_shift, a_rowidx, crows, stream);
  else if (strip) rc = launch_tn_strip(A, lda, B, ldb, workspace, R, Ka, No, a_scale, a_shift, a_rowidx, crows, Pcs, b_rowidx, groups, stream);
  else switch (nt) {
    case 13: rc = launch_tn<13>(A, lda, B, ldb, workspace, R, Ka, No, a_scale, a_shift, a_rowidx, crows, Pcs, b_rowidx, groups, stream); break;
    case 8: rc = launch_tn<8>(A, lda, B, ldb, workspace, R, Ka, No, a_scale, a_shift, a_rowidx, crows, Pcs, b_rowidx, groups, stream); break;
    case 7: rc = launch_tn<7>(A, lda, B, ldb, workspace, R, Ka, No, a_scale, a_shift, a_rowidx, crows, Pcs, b_rowidx, groups, stream); break;
    case 4: rc = launch_tn<4>(A, lda, B, ldb, workspace, R, Ka, No, a_scale, a_shift, a_rowidx, crows, Pcs, b_rowidx, groups, stream); break;
    default: rc = launch_tn<2>(A, lda, B, ldb, workspace, R, Ka, No, a_scale, a_shift, a_rowidx, crows, Pcs, b_rowidx, groups, stream); break;
  }
  if (rc != QAGNN_OK) return rc;
  const int64_t tot = (int64_t)Ka * No;
  if (ldc % 4 == 0 && aligned16(C)) k_sum_chunks4<<<cdiv(tot / 4, 64), SC_G * 64, 0, stream>>>(workspace, C, ldc, Ka, No, nchunks, accumulate);
  else k_sum_chunks<<<cdiv(tot, 256), 256, 0, stream>>>(workspace, C, ldc, Ka, No, nchunks, accumulate);
  QAGNN_LAUNCH_CHECK("k_sum_chunks");
  if (bsum) {
    k_sum_chunks<<<cdiv((int64_t)groups * No, 256), 256, 0, stream>>>(Pcs, bsum, No, groups, No, nchunks, 0);
    QAGNN_LAUNCH_CHECK("k_sum_chunks(colsum)");
  }
  return QAGNN_OK;
}

// C [Ka1 + Ka2, No] = [A1 | A2]^T B: the two weight gradients that share their B operand (X^T dK|dM|dQ and S^T dK|dM|dQ of a hop) as ONE
// split-K launch and ONE chunk sum where the bf16-split kernel takes the shapes; otherwise two qagnn_gemm_tn_f32 calls.
extern "C" int qagnn_gemm_tn2_f32(const float* A1, int32_t lda1, int32_t Ka1, const float* A2, int32_t lda2, int32_t Ka2, const float* B,
                                  int32_t ldb, float* C, int32_t ldc, int32_t R, int32_t No, float* workspace, qagnn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TimedScope timed(1, stream);
  QAGNN_REQUIRE(A1 && A2 && B && C && workspace, QAGNN_EINVAL, "gemm_tn2: null pointer");
  QAGNN_REQUIRE(R > 0 && Ka1 > 0 && Ka2 > 0 && No > 0 && Ka1 % 4 == 0 && Ka2 % 4 == 0 && No % 4 == 0, QAGNN_EINVAL,
                "gemm_tn2: bad sizes R=%d Ka1=%d Ka2=%d No=%d (Ka, No multiples of 4)", R, Ka1, Ka2, No);
  QAGNN_REQUIRE(lda1 % 4 == 0 && lda2 % 4 == 0 && ldb % 4 == 0 && aligned16(A1) && aligned16(A2) && aligned16(B), QAGNN_EINVAL,
                "gemm_tn2: operands must be 16-byte aligned with pitches multiple of 4");
  const bool merged = tn_split_ok(R, Ka1, No, lda1, ldb, false, false) && tn_split_ok(R, Ka2, No, lda2, ldb, false, false) && ldc % 4 == 0 &&
                      aligned16(C);
  if (!merged) {
    int rc = qagnn_gemm_tn_f32(A1, lda1, B, ldb, C, ldc, R, Ka1, No, nullptr, nullptr, nullptr, 0, workspace, stream_);
    if (rc != QAGNN_OK) return rc;
    return qagnn_gemm_tn_f32(A2, lda2, B, ldb, C + (int64_t)Ka1 * ldc, ldc, R, Ka2, No, nullptr, nullptr, nullptr, 0, workspace, stream_);
  }
  const int crows = tn_split2_chunk_rows(R, Ka1, Ka2, No, tn_split_min_chunk(R));
  const int nchunks = cdiv(R, crows), Ka = Ka1 + Ka2;
  int rc = launch_tn_split2(A1, lda1, Ka1, A2, lda2, Ka2, B, ldb, workspace, R, No, crows, stream);
  if (rc != QAGNN_OK) return rc;
  const int64_t tot = (int64_t)Ka * No;
  k_sum_chunks4<<<cdiv(tot / 4, 64), SC_G * 64, 0, stream>>>(workspace, C, ldc, Ka, No, nchunks, 0);
  QAGNN_LAUNCH_CHECK("k_sum_chunks");
  return QAGNN_OK;
}

extern "C" int qagnn_gemm_tn_f32(const float* A, int32_t lda, const float* B, int32_t ldb, float* C, int32_t ldc, int32_t R,
                                 int32_t Ka, int32_t No, const float* a_scale, const float* a_shift, const int64_t* a_rowidx,
                                 int32_t accumulate, float* workspace, qagnn_stream_t stream_) {
  return qagnn_gemm_tn_colsum_f32(A, lda, B, ldb, C, ldc, R, Ka, No, a_scale, a_shift, a_rowidx, accumulate, nullptr, nullptr, 0,
                                  workspace, stream_);
}

// The weight-gradient products in the three-MFMA form (scaled two-piece fp16 split: gemm_nn2.hip's header, gemm_split.hip's NP = 2 kernels).
// Same chunking, same ordered chunk sum as the six-MFMA route; shapes the split kernels do not take fall back to it (amax unused).
static int gemm_tn_scaled(const float* A1, int32_t lda1, int32_t Ka1, const float* A2, int32_t lda2, int32_t Ka2, const float* B,
                          int32_t ldb, float* C, int32_t ldc, int32_t R, int32_t No, const float* a_scale, const float* a_shift,
                          const uint32_t* amax_a1, const uint32_t* amax_a2, const uint32_t* amax_b, float* workspace, int np,
                          qagnn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TimedScope timed(1, stream);
  const bool two = A2 != nullptr && Ka2 > 0;
  QAGNN_REQUIRE(A1 && B && C && workspace, QAGNN_EINVAL, "gemm_tn_h2: null pointer");
  QAGNN_REQUIRE(!two || !a_scale, QAGNN_EINVAL, "gemm_tn_h2: the two-operand product has no BatchNorm prologue");
  const bool have = amax_a1 && amax_b && (!two || amax_a2);
  const bool ok = have && ldc % 4 == 0 && aligned16(C) && tn_split_ok(R, Ka1, No, lda1, ldb, false, a_scale != nullptr) &&
                  (!two || tn_split_ok(R, Ka2, No, lda2, ldb, false, false));
  if (!ok) {
    if (two) return qagnn_gemm_tn2_f32(A1, lda1, Ka1, A2, lda2, Ka2, B, ldb, C, ldc, R, No, workspace, stream_);
    return qagnn_gemm_tn_f32(A1, lda1, B, ldb, C, ldc, R, Ka1, No, a_scale, a_shift, nullptr, 0, workspace, stream_);
  }
  QAGNN_REQUIRE(R > 0 && Ka1 % 4 == 0 && No % 4 == 0 && (!two || Ka2 % 4 == 0), QAGNN_EINVAL, "gemm_tn_h2: bad sizes");
  QAGNN_REQUIRE(lda1 % 4 == 0 && ldb % 4 == 0 && aligned16(A1) && aligned16(B) && (!two || (lda2 % 4 == 0 && aligned16(A2))), QAGNN_EINVAL,
                "gemm_tn_h2: operands must be 16-byte aligned with pitches multiple of 4");
  QAGNN_REQUIRE(!a_scale || (a_shift && aligned16(a_scale) && aligned16(a_shift)), QAGNN_EINVAL, "gemm_tn_h2: a_scale/a_shift must both be given");
  const uint32_t* am[3] = {amax_a1, amax_a2, amax_b};
  const int Ka = Ka1 + (two ? Ka2 : 0);
  const int crows = two ? tn_split2_chunk_rows(R, Ka1, Ka2, No, tn_split_min_chunk(R)) : tn_split_chunk_rows(R, Ka1, No, tn_split_min_chunk(R));
  const int nchunks = cdiv(R, crows);
  int rc = two ? launch_tn_split2(A1, lda1, Ka1, A2, lda2, Ka2, B, ldb, workspace, R, No, crows, stream, am, np)
               : launch_tn_split(A1, lda1, B, ldb, workspace, R, Ka1, No, a_scale, a_shift, nullptr, crows, stream, am, np);
  if (rc != QAGNN_OK) return rc;
  k_sum_chunks4<<<cdiv((int64_t)Ka * No / 4, 64), SC_G * 64, 0, stream>>>(workspace, C, ldc, Ka, No, nchunks, 0);
  QAGNN_LAUNCH_CHECK("k_sum_chunks");
  return QAGNN_OK;
}

extern "C" int qagnn_gemm_tn_h2_f32(const float* A1, int32_t lda1, int32_t Ka1, const float* A2, int32_t lda2, int32_t Ka2, const float* B,
                                    int32_t ldb, float* C, int32_t ldc, int32_t R, int32_t No, const float* a_scale, const float* a_shift,
                                    const uint32_t* amax_a1, const uint32_t* amax_a2, const uint32_t* amax_b, float* workspace,
                                    qagnn_stream_t stream_) {
  return gemm_tn_scaled(A1, lda1, Ka1, A2, lda2, Ka2, B, ldb, C, ldc, R, No, a_scale, a_shift, amax_a1, amax_a2, amax_b, workspace, 2, stream_);
}

// The reduced-precision form of the same products (ONE fp16 MFMA per product, operands rounded to fp16 under the same scales: see
// qagnn_gemm_nn_args.pieces); on request only -- qagnn_hop_args.gemm_split == 3
extern "C" int qagnn_gemm_tn_h1_f32(const float* A1, int32_t lda1, int32_t Ka1, const float* A2, int32_t lda2, int32_t Ka2, const float* B,
                                    int32_t ldb, float* C, int32_t ldc, int32_t R, int32_t No, const float* a_scale, const float* a_shift,
                                    const uint32_t* amax_a1, const uint32_t* amax_a2, const uint32_t* amax_b, float* workspace,
                                    qagnn_stream_t stream_) {
  return gemm_tn_scaled(A1, lda1, Ka1, A2, lda2, Ka2, B, ldb, C, ldc, R, No, a_scale, a_shift, amax_a1, amax_a2, amax_b, workspace, 1, stream_);
}
