// Column reductions and elementwise kernels around the dense part of a GATConvE layer (HBM-bound, float4 lanes).
//
// Reference ops replaced: BatchNorm1d statistics / backward inside GATConvE.mlp (modeling_qagnn.py:408,443; over ALL
// N rows, PAD rows included), tanh-GELU (utils/layers.py:10-14) + F.dropout (modeling_qagnn.py:48-49, 92-93),
// the sin basis of the node-score embedding (modeling_qagnn.py:70-72), and bias / type-table gradient sums.
#include "common.h"

namespace qagnn {

// rows per wave (a block = 4 waves): 32 -> >= 2 blocks per CU at N = 64 000.  Below 32 768 rows that shape leaves CUs idle while
// every wave walks 4 dependent batches of row loads (2 000 rows: 16 blocks, 12-15 us per pass, pure latency): 8 rows per wave = ONE
// batch in flight per wave and 4x the blocks.  The choice depends on R only, so every pass over the same matrix (and the fused
// column-sum pass below) keeps the same partial sums.
constexpr int CR_WR_BIG = 32, CR_WR_SMALL = 8;
static inline int cr_wr(int R) { return R < 32768 ? CR_WR_SMALL : CR_WR_BIG; }

// MODE 0: grouped column sums   MODE 1: sum (x-mean)^2   MODE 2: BN+ReLU backward reductions (2 outputs)
// roww (modes 0, 1): optional per-row weight -> weighted sums (count-weighted BatchNorm statistics of the edge-class table)
template <int MODE, int CR_WR>
__global__ __launch_bounds__(256) void k_colreduce(const float* __restrict__ X, int ldx, const float* __restrict__ X2, int ldx2,
                                                   int R, int Cc, const int64_t* __restrict__ rowidx, int groups,
                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   const float* __restrict__ roww, float* __restrict__ part) {
  constexpr int NOUT = MODE == 0 ? 4 : (MODE == 2 ? 2 : 1);
  __shared__ __attribute__((aligned(16))) float red[4][NOUT][256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = blockIdx.x * 256 + lane * 4;
  const bool act = col < Cc;
  constexpr int CR_ROWS = 4 * CR_WR;
  const int r0 = blockIdx.y * CR_ROWS + w * CR_WR;
  float4 acc[NOUT];
#pragma unroll
  for (int o = 0; o < NOUT; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = mu, sc = mu, sh = mu;
  if (act && MODE >= 1) mu = ld4(mean + col);
  if (act && MODE == 2) { is = ld4(invstd + col); sc = ld4(scale + col); sh = ld4(shift + col); }
  if (act) {
    const int rend = min(R, r0 + CR_WR);
    // 8 rows per step: the 8 (16 in mode 2) row loads are issued back to back before any of them is consumed
    for (int rb = r0; rb < rend; rb += 8) {
      float4 xs[8], hs[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = min(rb + u, rend - 1);
        xs[u] = ld4(X + (int64_t)r * ldx + col);
        if (MODE == 2) hs[u] = ld4(X2 + (int64_t)r * ldx2 + col);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = rb + u;
        if (r >= rend) break;
        float4 x = xs[u];
        const float wr = (MODE != 2 && roww) ? roww[r] : 1.f;  // wave-uniform
        if (MODE == 0) {
          x = make_float4(x.x * wr, x.y * wr, x.z * wr, x.w * wr);
          const int g = rowidx ? (int)rowidx[r] : 0;
          // wave-uniform branch: every lane of the wave is on the same row
          if (g == 0) acc[0] = add4(acc[0], x);
          else if (g == 1) acc[1] = add4(acc[1], x);
          else if (g == 2) acc[2] = add4(acc[2], x);
          else acc[3] = add4(acc[3], x);
        } else if (MODE == 1) {
          const float4 d = make_float4(x.x - mu.x, x.y - mu.y, x.z - mu.z, x.w - mu.w);
          acc[0] = make_float4(fmaf(d.x * wr, d.x, acc[0].x), fmaf(d.y * wr, d.y, acc[0].y), fmaf(d.z * wr, d.z, acc[0].z),
                               fmaf(d.w * wr, d.w, acc[0].w));
        } else {
          const float4 h = hs[u];
          float4 dy;
          dy.x = fmaf(h.x, sc.x, sh.x) > 0.f ? x.x : 0.f;
          dy.y = fmaf(h.y, sc.y, sh.y) > 0.f ? x.y : 0.f;
          dy.z = fmaf(h.z, sc.z, sh.z) > 0.f ? x.z : 0.f;
          dy.w = fmaf(h.w, sc.w, sh.w) > 0.f ? x.w : 0.f;
          acc[0] = add4(acc[0], dy);
          acc[1].x = fmaf(dy.x, (h.x - mu.x) * is.x, acc[1].x);
          acc[1].y = fmaf(dy.y, (h.y - mu.y) * is.y, acc[1].y);
          acc[1].z = fmaf(dy.z, (h.z - mu.z) * is.z, acc[1].z);
          acc[1].w = fmaf(dy.w, (h.w - mu.w) * is.w, acc[1].w);
        }
      }
    }
  }
#pragma unroll
  for (int o = 0; o < NOUT; ++o) st4(&red[w][o][lane * 4], acc[o]);
  __syncthreads();
  const int nout = MODE == 0 ? groups : NOUT;
  for (int i = threadIdx.x; i < nout * 256; i += 256) {
    const int o = i >> 8, c = i & 255;
    if (blockIdx.x * 256 + c < Cc) {
      const float s = (red[0][o][c] + red[1][o][c]) + (red[2][o][c] + red[3][o][c]);
      part[((int64_t)blockIdx.y * nout + o) * Cc + blockIdx.x * 256 + c] = s;
    }
  }
}

// ordered sum of the chunk partials: 64 outputs x 16 chunk partitions per block (partition q takes chunks q, q+16, ...),
// partitions combined in fixed order.  16 partitions because the kernel is pure latency: 500 partials per output behind
// 4 partitions took 8 us, 31 times per step.
constexpr int CF_Q = 16;
__global__ __launch_bounds__(64 * CF_Q) void k_colreduce_final(const float* __restrict__ part, float* __restrict__ out, int nchunks,
                                                               int tot, float out_scale) {
  __shared__ float red[CF_Q][64];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int o = blockIdx.x * 64 + lane;
  float s0 = 0.f, s1 = 0.f;
  if (o < tot) {
    int c = q;
#pragma unroll 4
    for (; c + CF_Q < nchunks; c += 2 * CF_Q) {
      s0 += part[(int64_t)c * tot + o];
      s1 += part[(int64_t)(c + CF_Q) * tot + o];
    }
    if (c < nchunks) s0 += part[(int64_t)c * tot + o];
  }
  red[q][lane] = s0 + s1;
  __syncthreads();
  if (q == 0 && o < tot) {
    float s = red[0][lane];
#pragma unroll
    for (int k = 1; k < CF_Q; ++k) s += red[k][lane];
    out[o] = s * out_scale;
  }
}


__global__ void k_bn_relu_bwd(const float* __restrict__ dR, const float* __restrict__ Hh, float* __restrict__ dH, int ld, int R,
                              int Cc, const float* __restrict__ mean, const float* __restrict__ invstd,
                              const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ gamma,
                              const float* __restrict__ sum_dy, const float* __restrict__ sum_dy_hhat, float inv_rows,
                              const float* __restrict__ roww) {
  const int c4n = Cc >> 2;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)R * c4n) return;
  const int r = (int)(i / c4n), col = (int)(i % c4n) * 4;
  if (roww) inv_rows = roww[r];  // weighted statistics: row r entered mean / variance with weight roww[r] instead of 1/R
  const int64_t off = (int64_t)r * ld + col;
  const float4 g = ld4(dR + off), h = ld4(Hh + off);
  const float4 mu = ld4(mean + col), is = ld4(invstd + col), sc = ld4(scale + col), sh = ld4(shift + col);
  const float4 ga = ld4(gamma + col), k1 = ld4(sum_dy + col), k2 = ld4(sum_dy_hhat + col);
  float4 o;
#define ONE(f)                                              \
  {                                                         \
    const float dy = fmaf(h.f, sc.f, sh.f) > 0.f ? g.f : 0.f; \
    const float hh = (h.f - mu.f) * is.f;                   \
    o.f = ga.f * is.f * (dy - k1.f * inv_rows - hh * (k2.f * inv_rows)); \
  }
  ONE(x) ONE(y) ONE(z) ONE(w)
#undef ONE
  st4(dH + off, o);
}

// BatchNorm1d bookkeeping of GATConvE.mlp in ONE launch (torch.nn.BatchNorm1d semantics, modeling_qagnn.py:408):
//   invstd = 1/sqrt(var + eps), scale = gamma * invstd, shift = beta - mean * scale   (the GEMM operand prologue's affine)
//   training: running_mean/var <- lerp(., batch mean / UNBIASED batch var, momentum), num_batches_tracked += 1
// mean/var/gamma/beta are head-padded [Cc]; the module's running buffers are dense [d], dense_pos[k] = padded index of k.
__global__ void k_bn_finalize(const float* __restrict__ mean, const float* __restrict__ var, const float* __restrict__ gamma,
                              const float* __restrict__ beta, float eps, float* __restrict__ invstd, float* __restrict__ scale,
                              float* __restrict__ shift, int Cc, float* __restrict__ run_mean, float* __restrict__ run_var,
                              int64_t* __restrict__ nbt, const int64_t* __restrict__ dense_pos, int d, float momentum, float unbias,
                              int ones_col) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < Cc) {
    const float is = rsqrtf(var[i] + eps), sc = gamma[i] * is;
    invstd[i] = is;
    // ones_col (a zero-padding column of the head-padded layout, or -1): relu(h * 0 + 1) = 1 there, so the operand relu(bn(h1))
    // carries a column of ones -- its row of the weight gradient relu(bn(h1))^T dout is then the bias gradient colsum(dout) for free,
    // while the forward product is unchanged (the matching weight row is zero padding)
    scale[i] = i == ones_col ? 0.f : sc;
    shift[i] = i == ones_col ? 1.f : beta[i] - mean[i] * sc;
  }
  if (run_mean && i < d) {
    const int64_t p = dense_pos[i];
    run_mean[i] += momentum * (mean[p] - run_mean[i]);
    run_var[i] += momentum * (var[p] * unbias - run_var[i]);
  }
  if (nbt && i == 0) *nbt += 1;
}

// BatchNorm batch statistics from the per-tile partials of the GEMM that wrote the BatchNorm input (qagnn_gemm_nn_args.colstat_part:
// per 128-row tile t and column c: x0 = the tile's first value, S1 = sum (x - x0), S2 = sum (x - x0)^2), combined by the pairwise
// update of Chan, Golub & LeVeque -- every tile is shifted by one of its own values, so nothing cancels -- plus the whole
// bookkeeping of k_bn_finalize, in one launch and ONE pass over the partials.  A block owns 16 columns; its 1024 threads are
// 16 columns x 64 partitions (partition q merges tiles q, q + 64, ... in order), the 64 partition results are merged in a fixed binary
// tree: a fixed order, whatever the timing.
constexpr int ST_TILE = 128;  // = SBM of gemm_split.hip
constexpr int BF_COLS = 16, BF_PARTS = 64;
struct Moments { float n, mean, m2; };
__device__ __forceinline__ void merge(Moments& a, float nb, float mb, float m2b) {
  if (nb <= 0.f) return;
  const float n = a.n + nb, d = mb - a.mean;
  a.mean += d * (nb / n);
  a.m2 += m2b + d * d * (a.n * nb / n);
  a.n = n;
}
__global__ __launch_bounds__(BF_COLS * BF_PARTS) void k_bn_stats_finalize(const float* __restrict__ part, int nt, int R, int Cc,
                                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                        float eps, float* __restrict__ stats, float* __restrict__ run_mean,
                                                                        float* __restrict__ run_var, int64_t* __restrict__ nbt, int d,
                                                                        float momentum, float unbias, int ones_col,
                                                                        uint32_t* __restrict__ amax_bound) {
  __shared__ float sn[BF_PARTS][BF_COLS], sm[BF_PARTS][BF_COLS], s2[BF_PARTS][BF_COLS];
  const int cl = threadIdx.x & (BF_COLS - 1), q = threadIdx.x / BF_COLS;
  const int c = blockIdx.x * BF_COLS + cl;
  Moments a = {0.f, 0.f, 0.f};
  if (c < Cc)
    for (int t0 = q; t0 < nt; t0 += 4 * BF_PARTS) {  // four tiles' partials in flight together (the kernel is load latency: 500 tiles = 8 per thread)
      float x0[4], S1[4], S2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t t = min(t0 + u * BF_PARTS, nt - 1);  // clamped: unconditional loads
        x0[u] = part[(t * 3 + 0) * Cc + c]; S1[u] = part[(t * 3 + 1) * Cc + c]; S2[u] = part[(t * 3 + 2) * Cc + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + u * BF_PARTS;
        if (t < nt) {
          const float n_t = (float)min(ST_TILE, R - t * ST_TILE);
          merge(a, n_t, x0[u] + S1[u] / n_t, S2[u] - S1[u] * S1[u] / n_t);
        }
      }
    }
  sn[q][cl] = a.n; sm[q][cl] = a.mean; s2[q][cl] = a.m2;
  __syncthreads();
  // the 64 partitions meet in a fixed binary tree (6 levels; partition q absorbs q + stride): the column's first thread ends with all of them
#pragma unroll
  for (int stride = 1; stride < BF_PARTS; stride *= 2) {
    if ((q & (2 * stride - 1)) == 0 && c < Cc) {
      merge(a, sn[q + stride][cl], sm[q + stride][cl], s2[q + stride][cl]);
      sn[q][cl] = a.n; sm[q][cl] = a.mean; s2[q][cl] = a.m2;
    }
    __syncthreads();
  }
  if (q == 0 && c < Cc) {
    const float mean = a.mean, var = fmaxf(a.m2 / (float)R, 0.f);  // biased, as BatchNorm normalises with
    const float is = rsqrtf(var + eps), sc = gamma[c] * is;
    stats[c] = mean;
    stats[Cc + c] = var;
    stats[2 * Cc + c] = is;
    stats[3 * Cc + c] = c == ones_col ? 0.f : sc;                     // (see k_bn_finalize: the column of ones)
    stats[4 * Cc + c] = c == ones_col ? 1.f : beta[c] - mean * sc;
    // amax_bound: an upper bound of max |relu(bn(h1))| without a pass over h1 (the operand maximum the three-MFMA GEMM form asks for,
    // gemm_nn2.hip): (x - mean)^2 <= sum_r (x_r - mean)^2 = R var, so |x - mean| invstd <= sqrt(R var / (var + eps)) <= sqrt(R) and
    // |bn(x)| <= |gamma| sqrt(R) + |beta|; 1 % on top for the rounding of the fp32 statistics.  It overstates the true maximum by
    // sqrt(R) / (the largest |z| of the batch) ~ 2^6 at 64 000 rows: inside the form's 2^19 window.
    if (amax_bound) {
      const float bnd = c == ones_col ? 1.f : 1.01f * (fabsf(gamma[c]) * sqrtf((float)R) + fabsf(beta[c]));
      atomicMax(amax_bound, __builtin_bit_cast(uint32_t, bnd));
    }
    if (run_mean) {  // head-padded column c = h * HP + j is dense feature h * dh + j when j < dh (4 heads: ops.HeadLayout)
      const int HP = Cc >> 2, dh = d >> 2, h = c / HP, j = c - h * HP;
      if (j < dh) {
        const int i = h * dh + j;
        run_mean[i] += momentum * (mean - run_mean[i]);
        run_var[i] += momentum * (var * unbias - run_var[i]);
      }
    }
  }
  if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
}

// ---- GELU (tanh form) + dropout -----------------------------------------------------------------------------
__device__ __forceinline__ float gelu_f(float x) {
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  return 0.5f * x * (1.f + tanhf(u));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float x2 = x * x;
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x2);
  const float t = tanhf(u);
  return 0.5f * (1.f + t) + 0.5f * x * (1.f - t * t) * 0.7978845608028654f * (1.f + 3.f * 0.044715f * x2);
}
// AMAX: the launch also leaves max |out| of every block in amax_part[blockIdx.x] (a plain store; launch_amax_reduce folds them).  The
// kernel keeps its shape -- one float4 per thread, 13 000 blocks at 64 000 rows: it is ALU-bound (the counter hash) and wants every wave
// slot; two fat-block forms with one atomic per block (1 024 / 2 048 blocks) measured 20.6 -> 28.7 / 35.8 us, the atomics alone ~8 ns each
// on the kernel's tail (round 6, visits 4 and 10).
template <bool BWD>
__device__ __forceinline__ float4 gelu_dropout_val(float4 x, float4 g, int64_t i, float p, uint64_t seed, float inv) {
  float4 o;
#define ONE(f, k)                                                                  \
  {                                                                                \
    const float keep = (p > 0.f && uniform01(seed, (uint64_t)i * 4 + k) < p) ? 0.f : inv; \
    o.f = BWD ? g.f * keep * gelu_grad_f(x.f) : keep * gelu_f(x.f);                \
  }
  ONE(x, 0) ONE(y, 1) ONE(z, 2) ONE(w, 3)
#undef ONE
  return o;
}
template <bool BWD>
__device__ __forceinline__ float4 gelu_dropout_one(const float* __restrict__ X, const float* __restrict__ dY, int64_t i, float p, uint64_t seed, float inv) {
  return gelu_dropout_val<BWD>(ld4(X + i * 4), BWD ? ld4(dY + i * 4) : make_float4(1.f, 1.f, 1.f, 1.f), i, p, seed, inv);
}
template <bool BWD, bool AMAX = false>
__global__ __launch_bounds__(256) void k_gelu_dropout(const float* __restrict__ X, const float* __restrict__ dY, float* __restrict__ out, int64_t n4,
                                                      float p, uint64_t seed, const unsigned long long* __restrict__ epoch,
                                                      float* __restrict__ amax_part = nullptr) {
  if (p > 0.f) seed = epoch_seed(seed, epoch);
  const float inv = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (!AMAX && i >= n4) return;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    o = gelu_dropout_one<BWD>(X, dY, i, p, seed, inv);
    st4(out + i * 4, o);
  }
  if constexpr (AMAX) {  // (whole waves: idle lanes carry 0)
    __shared__ float red[4];
    const float m = wave_amax_lane63(absmax4(o));
    if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) amax_part[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  }
}

// max |x| of a tensor whose producer leaves none (qagnn_absmax_f32); N1: n counts single floats (k_amax_reduce: the edge kernels' per-node maxima)
template <bool N1>
__global__ __launch_bounds__(256) void k_absmax(const float* __restrict__ x, int64_t n, uint32_t* __restrict__ slot) {
  __shared__ float red[16];
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    if constexpr (N1) m = fmaxf(m, fabsf(x[i]));
    else m = fmaxf(m, absmax4(ld4(x + i * 4)));
  }
  block_amax_merge(m, slot, red);
}
__global__ void k_zero_words(uint32_t* __restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0u;
}

// ---- BatchNorm + ReLU backward that also leaves the column sums of what it writes -------------------------------------------
// d h1 is the B operand of the first Linear's weight gradient AND that Linear's bias gradient is its column sum; taking the
// sum in a separate pass re-reads 53 MB per layer.  Same block shape and the same summation order as k_colreduce<0> (32 rows
// per wave in row order, (w0 + w1) + (w2 + w3), chunk partials summed by k_colreduce_final), so the sums are bit-identical
// to the separate pass.  (The same fusion for the GELU + dropout backward was measured and dropped: 42 us against 27 + 18 --
// that kernel is ALU-bound and this block shape gives it only 2 blocks per CU.)
template <int CR_WR>
__global__ __launch_bounds__(256) void k_bn_relu_bwd_colsum(const float* __restrict__ dR, const float* __restrict__ Hh, float* __restrict__ dH,
                                                            int ld, int R, int Cc, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const float* __restrict__ gamma,
                                                            const float* __restrict__ k1v, const float* __restrict__ k2v, float inv_rows,
                                                            const float* __restrict__ roww, float* __restrict__ part,
                                                            uint32_t* __restrict__ amax) {
  __shared__ __attribute__((aligned(16))) float red[4][256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = blockIdx.x * 256 + lane * 4;
  const bool act = col < Cc;
  constexpr int CR_ROWS = 4 * CR_WR;
  const int r0 = blockIdx.y * CR_ROWS + w * CR_WR;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float am = 0.f;  // max |dH| of this thread's elements (amax != nullptr: merged per wave below)
  if (act) {
    const float4 mu = ld4(mean + col), is = ld4(invstd + col), sc = ld4(scale + col), sh = ld4(shift + col);
    const float4 ga = ld4(gamma + col), k1 = ld4(k1v + col), k2 = ld4(k2v + col);
    const int rend = min(R, r0 + CR_WR);
    for (int rb = r0; rb < rend; rb += 8) {
      float4 xs[8], hs[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = min(rb + u, rend - 1);
        xs[u] = ld4(dR + (int64_t)r * ld + col);
        hs[u] = ld4(Hh + (int64_t)r * ld + col);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = rb + u;
        if (r >= rend) break;
        const float4 x = xs[u], h = hs[u];
        const float ir = roww ? roww[r] : inv_rows;  // wave-uniform
        float4 o;
#define ONE(f)                                                          \
  {                                                                     \
    const float dy = fmaf(h.f, sc.f, sh.f) > 0.f ? x.f : 0.f;           \
    const float hh = (h.f - mu.f) * is.f;                               \
    o.f = ga.f * is.f * (dy - k1.f * ir - hh * (k2.f * ir));            \
  }
        ONE(x) ONE(y) ONE(z) ONE(w)
#undef ONE
        // the sums must add the ROUNDED outputs, like a separate pass over dH would: the empty asm makes o opaque, so the
        // compiler cannot contract its last multiply into the accumulation (fp-contract is on by default in HIP)
        asm volatile("" : "+v"(o.x), "+v"(o.y), "+v"(o.z), "+v"(o.w));
        st4(dH + (int64_t)r * ld + col, o);
        acc = add4(acc, o);
        am = fmaxf(am, absmax4(o));
      }
    }
  }
  st4(&red[w][lane * 4], acc);
  __syncthreads();
  const int c = threadIdx.x;
  if (blockIdx.x * 256 + c < Cc)
    part[(int64_t)blockIdx.y * Cc + blockIdx.x * 256 + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
  if (amax) {  // (kernel argument: uniform; idle lanes carry 0.  One atomic per block -- 500 at 64 000 rows)
    __shared__ float ared[16];
    block_amax_merge(am, amax, ared);
  }
}

// ---- the dropout seed epoch (see common.h) --------------------------------------------------------------------
__device__ unsigned long long g_seed_epoch = 0;  // one instance per device (module globals are per device)
__global__ void k_seed_epoch(unsigned long long* w, unsigned long long v, int add) { *w = add ? *w + v : v; }

const unsigned long long* seed_epoch_ptr() {
  static const unsigned long long* cache[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const unsigned long long*& p = cache[dev & 63];
  if (!p) {
    void* a = nullptr;
    if (hipGetSymbolAddress(&a, HIP_SYMBOL(g_seed_epoch)) != hipSuccess) return nullptr;
    p = static_cast<const unsigned long long*>(a);
  }
  return p;
}

__global__ void k_sin_basis(const float* __restrict__ score, const float* __restrict__ js, float* __restrict__ out, int ldo, int R,
                            int J) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)R * ldo) return;
  const int r = (int)(i / ldo), j = (int)(i % ldo);
  out[i] = j < J ? sinf(js[j] * score[r]) : 0.f;
}

}  // namespace qagnn

using namespace qagnn;

extern "C" int64_t qagnn_colreduce_workspace_elems(int32_t R, int32_t Cc, int32_t groups) {
  return (int64_t)cdiv(R, 4 * cr_wr(R)) * (groups < 2 ? 2 : groups) * Cc;
}

extern "C" int qagnn_colreduce_f32(int32_t mode, const float* X, int32_t ldx, const float* X2, int32_t ldx2, int32_t R, int32_t Cc,
                                   const int64_t* rowidx, int32_t groups, const float* mean, const float* invstd,
                                   const float* scale, const float* shift, const float* roww, float out_scale, float* out, float* workspace, qagnn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  QAGNN_REQUIRE(X && out && workspace, QAGNN_EINVAL, "colreduce: null pointer");
  QAGNN_REQUIRE(R > 0 && Cc > 0 && Cc % 4 == 0 && ldx % 4 == 0 && aligned16(X), QAGNN_EINVAL, "colreduce: bad sizes/alignment");
  QAGNN_REQUIRE(mode >= 0 && mode <= 2, QAGNN_EINVAL, "colreduce: bad mode %d", mode);
  const bool small = cr_wr(R) == CR_WR_SMALL;
  dim3 grid(cdiv(Cc, 256), cdiv(R, 4 * cr_wr(R)));
  int nout;
  if (mode == 0) {
    QAGNN_REQUIRE(groups >= 1 && groups <= 4 && (groups == 1 || rowidx), QAGNN_EINVAL, "colreduce: groups=%d (1..4)", groups);
    nout = groups;
    if (small) k_colreduce<0, CR_WR_SMALL><<<grid, 256, 0, stream>>>(X, ldx, X2, ldx2, R, Cc, rowidx, groups, mean, invstd, scale, shift, roww, workspace);
    else k_colreduce<0, CR_WR_BIG><<<grid, 256, 0, stream>>>(X, ldx, X2, ldx2, R, Cc, rowidx, groups, mean, invstd, scale, shift, roww, workspace);
  } else if (mode == 1) {
    QAGNN_REQUIRE(mean && aligned16(mean), QAGNN_EINVAL, "colreduce: mode 1 needs mean");
    nout = 1;
    if (small) k_colreduce<1, CR_WR_SMALL><<<grid, 256, 0, stream>>>(X, ldx, X2, ldx2, R, Cc, rowidx, groups, mean, invstd, scale, shift, roww, workspace);
    else k_colreduce<1, CR_WR_BIG><<<grid, 256, 0, stream>>>(X, ldx, X2, ldx2, R, Cc, rowidx, groups, mean, invstd, scale, shift, roww, workspace);
  } else {
    QAGNN_REQUIRE(X2 && mean && invstd && scale && shift && ldx2 % 4 == 0 && aligned16(X2), QAGNN_EINVAL,
                  "colreduce: mode 2 needs X2, mean, invstd, scale, shift");
    nout = 2;
    if (small) k_colreduce<2, CR_WR_SMALL><<<grid, 256, 0, stream>>>(X, ldx, X2, ldx2, R, Cc, rowidx, groups, mean, invstd, scale, shift, roww, workspace);
    else k_colreduce<2, CR_WR_BIG><<<grid, 256, 0, stream>>>(X, ldx, X2, ldx2, R, Cc, rowidx, groups, mean, invstd, scale, shift, roww, workspace);
  }
  QAGNN_LAUNCH_CHECK("k_colreduce");
  const int tot = nout * Cc;
  k_colreduce_final<<<cdiv(tot, 64), 64 * CF_Q, 0, stream>>>(workspace, out, grid.y, tot, out_scale);
  QAGNN_LAUNCH_CHECK("k_colreduce_final");
  return QAGNN_OK;
}

extern "C" int qagnn_bn_relu_bwd_f32(const float* dR, const float* Hh, float* dH, int32_t ld, int32_t R, int32_t Cc,
                                     const float* mean, const float* invstd, const float* scale, const float* shift,
                                     const float* gamma, const float* sum_dy, const float* sum_dy_hhat, float inv_rows,
                                     const float* roww, qagnn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  QAGNN_REQUIRE(dR && Hh && dH && mean && invstd && scale && shift && gamma && sum_dy && sum_dy_hhat, QAGNN_EINVAL,
                "bn_relu_bwd: null pointer");
  QAGNN_REQUIRE(R > 0 && Cc > 0 && Cc % 4 == 0 && ld % 4 == 0, QAGNN_EINVAL, "bn_relu_bwd: bad sizes");
  const int64_t tot = (int64_t)R * (Cc / 4);
  k_bn_relu_bwd<<<cdiv(tot, 256), 256, 0, stream>>>(dR, Hh, dH, ld, R, Cc, mean, invstd, scale, shift, gamma, sum_dy, sum_dy_hhat,
                                                    inv_rows, roww);
  QAGNN_LAUNCH_CHECK("k_bn_relu_bwd");
  return QAGNN_OK;
}

extern "C" int qagnn_bn_finalize_f32(const float* mean, const float* var, const float* gamma, const float* beta, float eps, float* invstd,
                                     float* scale, float* shift, int32_t Cc, float* run_mean, float* run_var, int64_t* num_batches_tracked,
                                     const int64_t* dense_pos, int32_t d, float momentum, float unbias, int32_t ones_col,
                                     qagnn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  QAGNN_REQUIRE(mean && var && gamma && beta && invstd && scale && shift && Cc > 0, QAGNN_EINVAL, "bn_finalize: null pointer");
  QAGNN_REQUIRE(!run_mean || (run_var && dense_pos && d > 0 && d <= Cc), QAGNN_EINVAL, "bn_finalize: running-stat arguments");
  k_bn_finalize<<<cdiv(Cc, 256), 256, 0, stream>>>(mean, var, gamma, beta, eps, invstd, scale, shift, Cc, run_mean, run_var,
                                                   num_batches_tracked, dense_pos, d, momentum, unbias, ones_col);
  QAGNN_LAUNCH_CHECK("k_bn_finalize");
  return QAGNN_OK;
}

namespace qagnn {
// (validated by the entry point / by hop.hip's own argument checks)
int launch_bn_stats_finalize(const float* part, int n_tiles, int R, int Cc, const float* gamma, const float* beta, float eps, float* stats,
                             float* run_mean, float* run_var, int64_t* nbt, int d, float momentum, float unbias, int ones_col, uint32_t* amax_bound,
                             hipStream_t stream) {
  k_bn_stats_finalize<<<cdiv(Cc, BF_COLS), BF_COLS * BF_PARTS, 0, stream>>>(part, n_tiles, R, Cc, gamma, beta, eps, stats, run_mean, run_var, nbt, d,
                                                                            momentum, unbias, ones_col, amax_bound);
  QAGNN_LAUNCH_CHECK("k_bn_stats_finalize");
  return QAGNN_OK;
}
}  // namespace qagnn

extern "C" int qagnn_bn_stats_finalize_f32(const float* part, int32_t n_tiles, int32_t R, int32_t Cc, const float* gamma, const float* beta,
                                           float eps, float* stats, float* run_mean, float* run_var, int64_t* num_batches_tracked,
                                           const int64_t* dense_pos, int32_t d, float momentum, float unbias, int32_t ones_col,
                                           qagnn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  QAGNN_REQUIRE(part && gamma && beta && stats && R > 0 && Cc > 0 && n_tiles == cdiv(R, ST_TILE), QAGNN_EINVAL,
                "bn_stats_finalize: bad arguments (R=%d needs %d tiles of %d rows, got %d)", R, cdiv(R, ST_TILE), ST_TILE, n_tiles);
  QAGNN_REQUIRE(!run_mean || (run_var && d > 0 && d % 4 == 0 && Cc % 4 == 0 && d <= Cc), QAGNN_EINVAL, "bn_stats_finalize: running-stat arguments");
  (void)dense_pos;  // the head-padded layout is implied by (Cc, d); kept in the signature for symmetry with qagnn_bn_finalize_f32
  return launch_bn_stats_finalize(part, n_tiles, R, Cc, gamma, beta, eps, stats, run_mean, run_var, num_batches_tracked, d, momentum, unbias, ones_col,
                                  nullptr, stream);
}

// ---- weight packing without the concatenations (qagnn_amd.ops.GatherPlan) ------------------------------------------------------
// Forward: out[i] = p[tid[i]][off[i]] (tid < 0: 0) -- the source tensors stay where they are, their addresses travel in the kernel
// arguments, and which tensor / which element a packed position comes from was worked out once when the plan was built.  Backward: for
// every source element s the sum, in fixed order k = 0..K-1, of the packed-gradient elements (tid[k][s], off[k][s]) (tid < 0 or an
// absent gradient tensor: 0), the packed gradients being separate tensors as autograd hands them over.
// Was: torch.cat + index_select forward; cat of ~90 gradient slices + K index_selects + K - 1 adds backward.
__global__ __launch_bounds__(256) void k_gather_multi(qagnn_gather_tabs t, const int* __restrict__ tid, const int* __restrict__ off,
                                                     float* __restrict__ out, int total) {
  __shared__ const float* ps[QAGNN_GATHER_MAX];
  for (int i = threadIdx.x; i < t.n; i += blockDim.x) ps[i] = t.p[i];
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int k = tid[i];
  out[i] = k >= 0 ? ps[k][off[i]] : 0.f;
}

__global__ __launch_bounds__(256) void k_gather_multi_sum(qagnn_gather_tabs t, const int* __restrict__ tid, const int* __restrict__ off, int K, int S,
                                                         float* __restrict__ out) {
  __shared__ const float* ps[QAGNN_GATHER_MAX];
  for (int i = threadIdx.x; i < t.n; i += blockDim.x) ps[i] = t.p[i];
  __syncthreads();
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) {
    const int g = tid[(int64_t)k * S + s];
    const float* p = g >= 0 ? ps[g] : nullptr;
    const float v = p != nullptr ? p[off[(int64_t)k * S + s]] : 0.f;
    acc = k == 0 ? v : acc + v;
  }
  out[s] = acc;
}

extern "C" int qagnn_gather_multi_f32(const qagnn_gather_tabs* t, const int32_t* tid, const int32_t* off, float* out, int32_t total,
                                      qagnn_stream_t stream_) {
  QAGNN_REQUIRE(t && t->n > 0 && t->n <= QAGNN_GATHER_MAX && tid && off && out && total > 0, QAGNN_EINVAL, "gather_multi: bad arguments");
  for (int i = 0; i < t->n; ++i) QAGNN_REQUIRE(t->p[i] != nullptr, QAGNN_EINVAL, "gather_multi: null source %d", i);
  k_gather_multi<<<cdiv(total, 256), 256, 0, (hipStream_t)stream_>>>(*t, tid, off, out, total);
  QAGNN_LAUNCH_CHECK("k_gather_multi");
  return QAGNN_OK;
}

extern "C" int qagnn_gather_multi_sum_f32(const qagnn_gather_tabs* t, const int32_t* tid, const int32_t* off, int32_t K, int32_t S, float* out,
                                          qagnn_stream_t stream_) {
  QAGNN_REQUIRE(t && t->n > 0 && t->n <= QAGNN_GATHER_MAX && tid && off && out && K > 0 && S > 0, QAGNN_EINVAL, "gather_multi_sum: bad arguments");
  k_gather_multi_sum<<<cdiv(S, 256), 256, 0, (hipStream_t)stream_>>>(*t, tid, off, K, S, out);
  QAGNN_LAUNCH_CHECK("k_gather_multi_sum");
  return QAGNN_OK;
}

namespace qagnn {
// dY == nullptr: forward.  amax != nullptr: max |out| is merged into that word; amax_part: scratch of gelu_amax_scratch_elems(n) floats
int64_t gelu_amax_scratch_elems(int64_t n) { return cdiv(n / 4, 256); }
int launch_gelu_dropout(const float* X, const float* dY, float* out, int64_t n, float p, uint64_t seed, uint32_t* amax, float* amax_part,
                        hipStream_t stream) {
  const int grid = cdiv(n / 4, 256);
  if (amax && !amax_part) { set_error("gelu_dropout: the maximum needs its scratch"); return QAGNN_EINVAL; }
  if (dY) {
    if (amax) k_gelu_dropout<true, true><<<grid, 256, 0, stream>>>(X, dY, out, n / 4, p, seed, seed_epoch_ptr(), amax_part);
    else k_gelu_dropout<true><<<grid, 256, 0, stream>>>(X, dY, out, n / 4, p, seed, seed_epoch_ptr());
  } else {
    if (amax) k_gelu_dropout<false, true><<<grid, 256, 0, stream>>>(X, nullptr, out, n / 4, p, seed, seed_epoch_ptr(), amax_part);
    else k_gelu_dropout<false><<<grid, 256, 0, stream>>>(X, nullptr, out, n / 4, p, seed, seed_epoch_ptr());
  }
  QAGNN_LAUNCH_CHECK("k_gelu_dropout");
  return amax ? launch_amax_reduce(amax_part, grid, amax, stream) : QAGNN_OK;
}
}  // namespace qagnn

extern "C" int qagnn_gelu_dropout_fwd_f32(const float* X, float* Y, int64_t n, float p, uint64_t seed, qagnn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  QAGNN_REQUIRE(X && Y && n > 0 && n % 4 == 0 && aligned16(X) && aligned16(Y), QAGNN_EINVAL, "gelu_dropout_fwd: bad args");
  QAGNN_REQUIRE(p >= 0.f && p < 1.f, QAGNN_EINVAL, "gelu_dropout_fwd: p=%f", p);
  return launch_gelu_dropout(X, nullptr, Y, n, p, seed, nullptr, nullptr, stream);
}

extern "C" int64_t qagnn_gelu_dropout_amax_scratch_elems(int64_t n) { return gelu_amax_scratch_elems(n); }
extern "C" int qagnn_gelu_dropout_fwd_amax_f32(const float* X, float* Y, int64_t n, float p, uint64_t seed, uint32_t* amax, float* scratch,
                                               qagnn_stream_t stream_) {
  QAGNN_REQUIRE(X && Y && amax && scratch && n > 0 && n % 4 == 0 && aligned16(X) && aligned16(Y), QAGNN_EINVAL, "gelu_dropout_fwd_amax: bad args");
  QAGNN_REQUIRE(p >= 0.f && p < 1.f, QAGNN_EINVAL, "gelu_dropout_fwd_amax: p=%f", p);
  return launch_gelu_dropout(X, nullptr, Y, n, p, seed, amax, scratch, (hipStream_t)stream_);
}

// the backward pass that leaves max |dX| behind (a consumer of dX in the three-MFMA form: the weight-gradient and data-gradient products of the
// Linear in front of the GELU); same scratch as the forward form
extern "C" int qagnn_gelu_dropout_bwd_amax_f32(const float* X, const float* dY, float* dX, int64_t n, float p, uint64_t seed, uint32_t* amax,
                                               float* scratch, qagnn_stream_t stream_) {
  QAGNN_REQUIRE(X && dY && dX && amax && scratch && n > 0 && n % 4 == 0 && aligned16(X) && aligned16(dY) && aligned16(dX), QAGNN_EINVAL,
                "gelu_dropout_bwd_amax: bad args");
  QAGNN_REQUIRE(p >= 0.f && p < 1.f, QAGNN_EINVAL, "gelu_dropout_bwd_amax: p=%f", p);
  return launch_gelu_dropout(X, dY, dX, n, p, seed, amax, scratch, (hipStream_t)stream_);
}

extern "C" int qagnn_gelu_dropout_bwd_f32(const float* X, const float* dY, float* dX, int64_t n, float p, uint64_t seed,
                                          qagnn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  QAGNN_REQUIRE(X && dY && dX && n > 0 && n % 4 == 0 && aligned16(X) && aligned16(dY) && aligned16(dX), QAGNN_EINVAL,
                "gelu_dropout_bwd: bad args");
  QAGNN_REQUIRE(p >= 0.f && p < 1.f, QAGNN_EINVAL, "gelu_dropout_bwd: p=%f", p);
  return launch_gelu_dropout(X, dY, dX, n, p, seed, nullptr, nullptr, stream);
}

extern "C" int qagnn_absmax_f32(const float* x, int64_t n, uint32_t* slot, qagnn_stream_t stream_) {
  QAGNN_REQUIRE(x && slot && n > 0 && n % 4 == 0 && aligned16(x), QAGNN_EINVAL, "absmax: bad arguments");
  const int64_t n4 = n / 4;
  const int grid = (int)(n4 / 256 + 1 < 1024 ? n4 / 256 + 1 : 1024);
  k_absmax<false><<<grid, 256, 0, (hipStream_t)stream_>>>(x, n4, slot);
  QAGNN_LAUNCH_CHECK("k_absmax");
  return QAGNN_OK;
}

// the per-node maxima of an edge kernel (non-negative floats) -> the word: 64 blocks, one atomic each
int qagnn::launch_amax_reduce(const float* part, int64_t n, uint32_t* slot, hipStream_t stream) {
  const int grid = (int)(n / 1024 + 1 < 64 ? n / 1024 + 1 : 64);
  k_absmax<true><<<grid, 256, 0, stream>>>(part, n, slot);
  QAGNN_LAUNCH_CHECK("k_amax_reduce");
  return QAGNN_OK;
}

extern "C" int qagnn_zero_words(uint32_t* p, int64_t n, qagnn_stream_t stream_) {
  QAGNN_REQUIRE(p && n > 0, QAGNN_EINVAL, "zero_words: bad arguments");
  k_zero_words<<<(int)(n / 256 + 1 < 256 ? n / 256 + 1 : 256), 256, 0, (hipStream_t)stream_>>>(p, n);
  QAGNN_LAUNCH_CHECK("k_zero_words");
  return QAGNN_OK;
}

extern "C" int qagnn_seed_epoch_advance(uint64_t delta, qagnn_stream_t stream_) {
  unsigned long long* w = const_cast<unsigned long long*>(seed_epoch_ptr());
  QAGNN_REQUIRE(w, QAGNN_EHIP, "seed_epoch: cannot resolve the epoch word on this device");
  k_seed_epoch<<<1, 1, 0, (hipStream_t)stream_>>>(w, delta, 1);
  QAGNN_LAUNCH_CHECK("k_seed_epoch");
  return QAGNN_OK;
}

extern "C" int qagnn_seed_epoch_set(uint64_t value, qagnn_stream_t stream_) {
  unsigned long long* w = const_cast<unsigned long long*>(seed_epoch_ptr());
  QAGNN_REQUIRE(w, QAGNN_EHIP, "seed_epoch: cannot resolve the epoch word on this device");
  k_seed_epoch<<<1, 1, 0, (hipStream_t)stream_>>>(w, value, 0);
  QAGNN_LAUNCH_CHECK("k_seed_epoch");
  return QAGNN_OK;
}

extern "C" int qagnn_sin_basis_f32(const float* score, const float* js, float* out, int32_t ldo, int32_t R, int32_t J,
                                   qagnn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  QAGNN_REQUIRE(score && js && out && R > 0 && J > 0 && ldo >= J, QAGNN_EINVAL, "sin_basis: bad args");
  const int64_t tot = (int64_t)R * ldo;
  k_sin_basis<<<cdiv(tot, 256), 256, 0, stream>>>(score, js, out, ldo, R, J);
  QAGNN_LAUNCH_CHECK("k_sin_basis");
  return QAGNN_OK;
}

// qagnn_bn_relu_bwd_f32 that also returns colsum[c] = sum_r dH[r][c] (the bias gradient of the Linear in front of the BatchNorm)
namespace qagnn {
// amax != nullptr: the launch also merges max |dH| into that word
int launch_bn_relu_bwd_colsum(const float* dR, const float* Hh, float* dH, int ld, int R, int Cc, const float* mean, const float* invstd,
                              const float* scale, const float* shift, const float* gamma, const float* sum_dy, const float* sum_dy_hhat,
                              float inv_rows, const float* roww, float* colsum, float* workspace, uint32_t* amax, hipStream_t stream) {
  dim3 grid(cdiv(Cc, 256), cdiv(R, 4 * cr_wr(R)));
  if (cr_wr(R) == CR_WR_SMALL)
    k_bn_relu_bwd_colsum<CR_WR_SMALL><<<grid, 256, 0, stream>>>(dR, Hh, dH, ld, R, Cc, mean, invstd, scale, shift, gamma, sum_dy, sum_dy_hhat,
                                                                inv_rows, roww, workspace, amax);
  else
    k_bn_relu_bwd_colsum<CR_WR_BIG><<<grid, 256, 0, stream>>>(dR, Hh, dH, ld, R, Cc, mean, invstd, scale, shift, gamma, sum_dy, sum_dy_hhat,
                                                              inv_rows, roww, workspace, amax);
  QAGNN_LAUNCH_CHECK("k_bn_relu_bwd_colsum");
  k_colreduce_final<<<cdiv(Cc, 64), 64 * CF_Q, 0, stream>>>(workspace, colsum, grid.y, Cc, 1.0f);
  QAGNN_LAUNCH_CHECK("k_colreduce_final");
  return QAGNN_OK;
}
}  // namespace qagnn

extern "C" int qagnn_bn_relu_bwd_colsum_f32(const float* dR, const float* Hh, float* dH, int32_t ld, int32_t R, int32_t Cc, const float* mean,
                                            const float* invstd, const float* scale, const float* shift, const float* gamma,
                                            const float* sum_dy, const float* sum_dy_hhat, float inv_rows, const float* roww, float* colsum,
                                            float* workspace, qagnn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  QAGNN_REQUIRE(dR && Hh && dH && mean && invstd && scale && shift && gamma && sum_dy && sum_dy_hhat && colsum && workspace, QAGNN_EINVAL,
                "bn_relu_bwd_colsum: null pointer");
  QAGNN_REQUIRE(R > 0 && Cc > 0 && Cc % 4 == 0 && ld % 4 == 0, QAGNN_EINVAL, "bn_relu_bwd_colsum: bad sizes");
  return launch_bn_relu_bwd_colsum(dR, Hh, dH, ld, R, Cc, mean, invstd, scale, shift, gamma, sum_dy, sum_dy_hhat, inv_rows, roww, colsum, workspace,
                                   nullptr, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Per-batch node bookkeeping of QAGNN.forward in one launch (reference modeling_qagnn.py:154, 160-167, 173-177):
//   ridx[g][v]   = concept_ids[g][v] - 1, and -1 for the context node v = 0 (its feature comes from svec2nvec, :153)
//   score[g][v]  = (-(s[g][v]) - (-(s[g][0]))) * [v < adj_len[g]]  /  (sum_v |.| / adj_len[g] + 1e-5)          (:160-167)
//   mask[g][v]   = v >= adj_len[g]  or  node_type[g][v] == 3;   mask[g][0] cleared when every slot of g is masked   (:173-177)
// One workgroup per subgraph.  The row sum of |score| is taken in float64 and rounded once: the correctly rounded fp32 sum, which is
// what ANY summation order gives when the sum is exactly representable (the reference's own order is a CPU / GPU library detail,
// and sin(1.1^j * score) downstream amplifies a 1-ulp difference by 1e4) -- and within 1 ulp of every fp32 order otherwise.
// ---------------------------------------------------------------------------------------------------------------
namespace qagnn {
__global__ __launch_bounds__(256) void k_node_prep(const float* __restrict__ raw, const int64_t* __restrict__ adj_len,
                                                   const int64_t* __restrict__ node_type, const int64_t* __restrict__ concept_ids, int n,
                                                   float* __restrict__ score, uint8_t* __restrict__ mask, int64_t* __restrict__ ridx,
                                                   int64_t table_rows, int32_t* __restrict__ err) {
  __shared__ double wsum[4];
  __shared__ int wall[4];
  const int g = blockIdx.x, tid = threadIdx.x;
  const int64_t len = adj_len[g];
  const float s0 = -raw[(int64_t)g * n];
  double part = 0.0;
  int unmasked = 0;
  for (int v = tid; v < n; v += 256) {
    const float d = (-raw[(int64_t)g * n + v] - s0) * (v < len ? 1.0f : 0.0f);
    part += (double)fabsf(d);
    unmasked |= !(v >= len || node_type[(int64_t)g * n + v] == 3);
  }
  for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
  unmasked = __any(unmasked);
  if ((tid & 63) == 0) { wsum[tid >> 6] = part; wall[tid >> 6] = unmasked; }
  __syncthreads();
  const float total = (float)(wsum[0] + wsum[1] + wsum[2] + wsum[3]);
  const bool all_masked = !(wall[0] | wall[1] | wall[2] | wall[3]);
  const float denom = total / (float)len + 1e-05f;
  for (int v = tid; v < n; v += 256) {
    const int64_t i = (int64_t)g * n + v;
    const float d = (-raw[i] - s0) * (v < len ? 1.0f : 0.0f);
    score[i] = d / denom;
    const bool m = v >= len || node_type[i] == 3;
    mask[i] = (v == 0 && all_masked) ? 0 : (m ? 1 : 0);
    int64_t r = v == 0 ? -1 : concept_ids[i] - 1;
    if (table_rows > 0 && v != 0 && (r < 0 || r >= table_rows)) {  // nn.Embedding raises on such an id (utils/layers.py:604); here: zero row + flag
      r = -1;
      if (err) err[0] = 1;
    }
    ridx[i] = r;
  }
}
}  // namespace qagnn

extern "C" int qagnn_node_prep_f32(const float* raw_scores, const int64_t* adj_len, const int64_t* node_type, const int64_t* concept_ids,
                                   int32_t B, int32_t n, float* score, uint8_t* mask, int64_t* ridx, int64_t table_rows, int32_t* err,
                                   qagnn_stream_t stream_) {
  QAGNN_REQUIRE(raw_scores && adj_len && node_type && concept_ids && score && mask && ridx && B > 0 && n > 0, QAGNN_EINVAL, "node_prep: bad arguments");
  qagnn::k_node_prep<<<B, 256, 0, (hipStream_t)stream_>>>(raw_scores, adj_len, node_type, concept_ids, n, score, mask, ridx, table_rows, err);
  QAGNN_LAUNCH_CHECK("k_node_prep");
  return QAGNN_OK;
}
