// Pooling head of QAGNN.forward: masked multi-head attention pooling of the GNN output over the nodes of one subgraph
// (reference: utils/layers.py:284-299 MatrixVectorScaledDotProductAttention inside :344-371 MultiheadAttPoolLayer, called at
// modeling_qagnn.py:178).  The host re-associates the projections (qagnn_amd/layers.py): per subgraph b and head h it hands in
// the query seen from node space u[b,h] (= Wk_h^T qs[b,h]) and the bias term c[b,h]; this file does the node-sized part,
//     score[b,h,l] = (<u[b,h], k[b,l]> + c[b,h]) / temperature,  masked -> -inf
//     attn = softmax_l(score),  attn_d = dropout(attn, p),  z[b,h] = sum_l attn_d[b,h,l] k[b,l]
// with ONE workgroup per subgraph that reads the subgraph's rows twice (second sweep out of L2) instead of the two batched
// mat-vec GEMMs + masked_fill + softmax + dropout (+ 4 GEMMs, an [B,n,d] add and ~20 small kernels in backward).
// k is the head-padded GNN output [B*n, ldk]; rows are read with one float4 per lane (Cc <= 256).
#include "common.h"

namespace qagnn {

constexpr int POOL_MAXH = 4;     // attention heads (reference: 2)
constexpr int POOL_MAXN = 1024;  // node slots per subgraph (reference: 200)

constexpr int POOL_W = 8;  // waves per workgroup; a wave takes 4 consecutive rows per step (4 row loads in flight)

// grid = B, block = 512 (8 waves)
__global__ __launch_bounds__(64 * POOL_W) void k_pool_fwd(const float* __restrict__ u, const float* __restrict__ cvec, const float* __restrict__ K,
                                                  int ldk, const uint8_t* __restrict__ mask, int n, int NH, int Cc, float inv_temp,
                                                  float p, uint64_t seed, const unsigned long long* __restrict__ epoch, float* __restrict__ attn,
                                                  float* __restrict__ attn_d, float* __restrict__ z) {
  if (p > 0.f) seed = epoch_seed(seed, epoch);
  __shared__ float sc[POOL_MAXH * POOL_MAXN];
  __shared__ __attribute__((aligned(16))) float red[POOL_W][POOL_MAXH][256];
  const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = lane * 4;
  const bool act = col < Cc;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 u4[POOL_MAXH];
#pragma unroll
  for (int h = 0; h < POOL_MAXH; ++h) u4[h] = (h < NH && act) ? ld4(u + ((int64_t)b * NH + h) * Cc + col) : zero;
  const float* Kb = K + (int64_t)b * n * ldk;

  // sweep 1: scores
  for (int l0 = w * 4; l0 < n; l0 += 4 * POOL_W) {
    float4 kv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) kv[j] = (act && l0 + j < n) ? ld4(Kb + (int64_t)(l0 + j) * ldk + col) : zero;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int l = l0 + j;
      if (l >= n) break;
      const bool out = mask[(int64_t)b * n + l] != 0;
#pragma unroll
      for (int h = 0; h < POOL_MAXH; ++h) {
        if (h >= NH) break;
        const float d = wave_sum(dot4(kv[j], u4[h]));
        if (lane == 0) sc[h * n + l] = out ? -INFINITY : (d + cvec[(int64_t)b * NH + h]) * inv_temp;
      }
    }
  }
  __syncthreads();
  // softmax over the nodes + dropout: wave h owns head h
  if (w < NH) {
    const int h = w;
    float m = -INFINITY;
    for (int l = lane; l < n; l += 64) m = fmaxf(m, sc[h * n + l]);
    m = wave_max(m);
    float s = 0.f;
    for (int l = lane; l < n; l += 64) {
      const float e = __expf(sc[h * n + l] - m);
      sc[h * n + l] = e;
      s += e;
    }
    s = wave_sum(s);
    const float inv = 1.f / s, keep_scale = 1.f / (1.f - p);
    const int64_t base = ((int64_t)b * NH + h) * n;
    for (int l = lane; l < n; l += 64) {
      const float a = sc[h * n + l] * inv;
      const float ad = (p > 0.f && uniform01(seed, (uint64_t)(base + l)) < p) ? 0.f : a * keep_scale;
      attn[base + l] = a;
      attn_d[base + l] = ad;
      sc[h * n + l] = ad;
    }
  }
  __syncthreads();
  // sweep 2: z[h] = sum_l attn_d[h][l] k[l]
  float4 acc[POOL_MAXH];
#pragma unroll
  for (int h = 0; h < POOL_MAXH; ++h) acc[h] = zero;
  for (int l0 = w * 4; l0 < n; l0 += 4 * POOL_W) {
    float4 kv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) kv[j] = (act && l0 + j < n) ? ld4(Kb + (int64_t)(l0 + j) * ldk + col) : zero;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (l0 + j >= n) break;
#pragma unroll
      for (int h = 0; h < POOL_MAXH; ++h) {
        if (h >= NH) break;
        acc[h] = fma4(sc[h * n + l0 + j], kv[j], acc[h]);
      }
    }
  }
#pragma unroll
  for (int h = 0; h < POOL_MAXH; ++h) st4(&red[w][h][col], acc[h]);
  __syncthreads();
  if (w == 0 && act) {
#pragma unroll
    for (int h = 0; h < POOL_MAXH; ++h) {
      if (h >= NH) break;
      float4 s = ld4(&red[0][h][col]);
#pragma unroll
      for (int k = 1; k < POOL_W; ++k) s = add4(s, ld4(&red[k][h][col]));
      st4(z + ((int64_t)b * NH + h) * Cc + col, s);
    }
  }
}

// backward: given dz [B, NH, Cc] and (optionally) d attn_d [B, NH, n] -> dK rows (written, not accumulated), du, dc
__global__ __launch_bounds__(64 * POOL_W) void k_pool_bwd(const float* __restrict__ u, const float* __restrict__ K, int ldk, int n, int NH, int Cc,
                                                  float inv_temp, float p, uint64_t seed, const unsigned long long* __restrict__ epoch,
                                                  const float* __restrict__ attn, const float* __restrict__ attn_d,
                                                  const float* __restrict__ dz, const float* __restrict__ dattn_d,
                                                  float* __restrict__ dK, int lddk, float* __restrict__ du, float* __restrict__ dc) {
  if (p > 0.f) seed = epoch_seed(seed, epoch);
  __shared__ float ga[POOL_MAXH * POOL_MAXN];  // sweep 1: d attn_d from the z path; then ds (already x 1/temperature)
  __shared__ __attribute__((aligned(16))) float red[POOL_W][POOL_MAXH][256];
  const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = lane * 4;
  const bool act = col < Cc;
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 u4[POOL_MAXH], g4[POOL_MAXH];
#pragma unroll
  for (int h = 0; h < POOL_MAXH; ++h) {
    const bool on = h < NH && act;
    u4[h] = on ? ld4(u + ((int64_t)b * NH + h) * Cc + col) : zero;
    g4[h] = on ? ld4(dz + ((int64_t)b * NH + h) * Cc + col) : zero;
  }
  const float* Kb = K + (int64_t)b * n * ldk;
  for (int l0 = w * 4; l0 < n; l0 += 4 * POOL_W) {
    float4 kv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) kv[j] = (act && l0 + j < n) ? ld4(Kb + (int64_t)(l0 + j) * ldk + col) : zero;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (l0 + j >= n) break;
#pragma unroll
      for (int h = 0; h < POOL_MAXH; ++h) {
        if (h >= NH) break;
        const float d = wave_sum(dot4(kv[j], g4[h]));
        if (lane == 0) ga[h * n + l0 + j] = d;
      }
    }
  }
  __syncthreads();
  if (w < NH) {
    const int h = w;
    const int64_t base = ((int64_t)b * NH + h) * n;
    const float keep_scale = 1.f / (1.f - p);
    float sdot = 0.f;
    for (int l = lane; l < n; l += 64) {
      const float a = attn[base + l];
      const float keep = (p > 0.f && uniform01(seed, (uint64_t)(base + l)) < p) ? 0.f : keep_scale;
      const float dat = (ga[h * n + l] + (dattn_d ? dattn_d[base + l] : 0.f)) * keep;  // gradient w.r.t. the softmax output
      ga[h * n + l] = dat;
      sdot += a * dat;
    }
    sdot = wave_sum(sdot);
    float dcs = 0.f;
    for (int l = lane; l < n; l += 64) {
      const float ds = attn[base + l] * (ga[h * n + l] - sdot) * inv_temp;  // d score-before-temperature
      ga[h * n + l] = ds;
      dcs += ds;
    }
    dcs = wave_sum(dcs);
    if (lane == 0) dc[(int64_t)b * NH + h] = dcs;
  }
  __syncthreads();
  float4 acc[POOL_MAXH];
#pragma unroll
  for (int h = 0; h < POOL_MAXH; ++h) acc[h] = zero;
  for (int l0 = w * 4; l0 < n; l0 += 4 * POOL_W) {
    float4 kv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) kv[j] = (act && l0 + j < n) ? ld4(Kb + (int64_t)(l0 + j) * ldk + col) : zero;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int l = l0 + j;
      if (l >= n) break;
      float4 o = zero;
#pragma unroll
      for (int h = 0; h < POOL_MAXH; ++h) {
        if (h >= NH) break;
        const float ds = ga[h * n + l];
        o = fma4(attn_d[((int64_t)b * NH + h) * n + l], g4[h], o);  // attention after dropout, as the forward stored it
        o = fma4(ds, u4[h], o);
        acc[h] = fma4(ds, kv[j], acc[h]);
      }
      if (act) st4(dK + ((int64_t)b * n + l) * lddk + col, o);
    }
  }
#pragma unroll
  for (int h = 0; h < POOL_MAXH; ++h) st4(&red[w][h][col], acc[h]);
  __syncthreads();
  if (w == 0 && act) {
#pragma unroll
    for (int h = 0; h < POOL_MAXH; ++h) {
      if (h >= NH) break;
      float4 s = ld4(&red[0][h][col]);
#pragma unroll
      for (int k = 1; k < POOL_W; ++k) s = add4(s, ld4(&red[k][h][col]));
      st4(du + ((int64_t)b * NH + h) * Cc + col, s);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// The rest of the head behind the pooling (reference utils/layers.py:366-371 + modeling_qagnn.py:178-182 with fc_layer_num = 0):
//     out[b]    = dropout_p1( Wv_h z[b,h] + bv_h sum_l attn[b,h,l] )                     (the value projection of the pooled rows)
//     logits[b] = < dropout_p2( [ out[b] | sent_vecs[b] | Z[b] ] ), w_fc > + b_fc        (Z = row 0 of the subgraph, dense part)
// One workgroup of 256 threads per subgraph each way instead of ~10 + ~20 stock elementwise / reduction / cat kernels: at the
// reference's mini-batch of 10 subgraphs a training step IS its ~300 kernel launches.  BDv is the block-diagonal value projection
// in the padded node layout ([NH * DP][NO], NO = NH * dv <= 256; qagnn_amd/layers.py), read column-wise by consecutive threads.
// The dropout masks are counter-based (uniform01) like every other mask of this library: backward regenerates them from the seeds.
// ------------------------------------------------------------------------------------------------------------
constexpr int HEAD_T = 256;

__device__ __forceinline__ float block_sum_256(float v, float* red /* [4] */) {
  v = wave_sum(v);
  __syncthreads();  // (red may still be read from the previous reduction)
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// dense feature k of a GNN row -> its position in the head-padded row (4 GAT heads of dh features, HP slots each)
__device__ __forceinline__ int padded_pos(int k, int dh, int HP) { return (k / dh) * HP + (k % dh); }

__global__ __launch_bounds__(HEAD_T) void k_head_post_fwd(const float* __restrict__ z, const float* __restrict__ attn, const float* __restrict__ BDv,
                                                         const float* __restrict__ bv, const float* __restrict__ sent, const float* __restrict__ H,
                                                         int64_t ldh, const float* __restrict__ w, const float* __restrict__ bfc, int NH, int DP, int dv,
                                                         int n, int Ds, int d, int dh, int HP, float p1, float p2, uint64_t seed1, uint64_t seed2,
                                                         const unsigned long long* __restrict__ epoch, float* __restrict__ out,
                                                         float* __restrict__ asum, float* __restrict__ logits) {
  __shared__ float zs[POOL_MAXH * 256], outd[HEAD_T], as[POOL_MAXH], red[4];
  const int b = blockIdx.x, tid = threadIdx.x, NO = NH * dv, L = NO + Ds + d;
  if (p1 > 0.f) seed1 = epoch_seed(seed1, epoch);
  if (p2 > 0.f) seed2 = epoch_seed(seed2, epoch);
  for (int i = tid; i < NH * DP; i += HEAD_T) zs[i] = z[(int64_t)b * NH * DP + i];
  for (int h = 0; h < NH; ++h) {
    float s_ = 0.f;
    for (int l = tid; l < n; l += HEAD_T) s_ += attn[((int64_t)b * NH + h) * n + l];
    s_ = block_sum_256(s_, red);
    if (tid == 0) { as[h] = s_; asum[b * NH + h] = s_; }
  }
  __syncthreads();
  if (tid < NO) {
    const int h = tid / dv;
    const float* col = BDv + (int64_t)h * DP * NO + tid;
    float acc = 0.f;
    for (int j = 0; j < DP; ++j) acc = fmaf(zs[h * DP + j], col[(int64_t)j * NO], acc);
    acc = fmaf(bv[tid], as[h], acc);
    out[(int64_t)b * NO + tid] = acc;
    const float keep = (p1 > 0.f && uniform01(seed1, (uint64_t)b * NO + tid) < p1) ? 0.f : 1.f / (1.f - p1);
    outd[tid] = p1 > 0.f ? acc * keep : acc;
  }
  __syncthreads();
  const float inv2 = p2 > 0.f ? 1.f / (1.f - p2) : 1.f;
  const float* Hb = H + (int64_t)b * ldh;
  float part = 0.f;
  for (int k = tid; k < L; k += HEAD_T) {
    const float v = k < NO ? outd[k] : k < NO + Ds ? sent[(int64_t)b * Ds + (k - NO)] : Hb[padded_pos(k - NO - Ds, dh, HP)];
    const float keep = (p2 > 0.f && uniform01(seed2, (uint64_t)b * L + k) < p2) ? 0.f : inv2;
    part = fmaf(v * keep, w[k], part);
  }
  part = block_sum_256(part, red);
  if (tid == 0) logits[b] = part + bfc[0];
}

// part [B][PW >= L + NO + 1]: per subgraph the addends of d w_fc (L), d bv (NO) and d b_fc (1); their column sums are the gradients
__global__ __launch_bounds__(HEAD_T) void k_head_post_bwd(const float* __restrict__ dlogits, const float* __restrict__ out, const float* __restrict__ asum,
                                                         const float* __restrict__ BDv, const float* __restrict__ bv, const float* __restrict__ sent,
                                                         const float* __restrict__ H, int64_t ldh, const float* __restrict__ w, int NH, int DP, int dv,
                                                         int n, int Ds, int d, int dh, int HP, float p1, float p2, uint64_t seed1, uint64_t seed2,
                                                         const unsigned long long* __restrict__ epoch, float* __restrict__ dz,
                                                         float* __restrict__ dattn, float* __restrict__ dout, float* __restrict__ dsent,
                                                         float* __restrict__ dZ, float* __restrict__ part, int PW) {
  __shared__ float douts[HEAD_T], red[4];
  const int b = blockIdx.x, tid = threadIdx.x, NO = NH * dv, L = NO + Ds + d;
  if (p1 > 0.f) seed1 = epoch_seed(seed1, epoch);
  if (p2 > 0.f) seed2 = epoch_seed(seed2, epoch);
  const float dl = dlogits[b];
  const float inv1 = p1 > 0.f ? 1.f / (1.f - p1) : 1.f, inv2 = p2 > 0.f ? 1.f / (1.f - p2) : 1.f;
  const float* Hb = H + (int64_t)b * ldh;
  float* pb = part + (int64_t)b * PW;
  for (int j = tid; j < 4 * HP; j += HEAD_T) dZ[(int64_t)b * 4 * HP + j] = 0.f;  // (the pads of the row stay zero)
  __syncthreads();
  for (int k = tid; k < L; k += HEAD_T) {
    const float keep2 = (p2 > 0.f && uniform01(seed2, (uint64_t)b * L + k) < p2) ? 0.f : inv2;
    const float dc = dl * w[k] * keep2;
    float v;
    if (k < NO) {
      const float keep1 = (p1 > 0.f && uniform01(seed1, (uint64_t)b * NO + k) < p1) ? 0.f : inv1;
      v = out[(int64_t)b * NO + k] * keep1;
      const float g = dc * keep1;
      douts[k] = g;
      dout[(int64_t)b * NO + k] = g;
      pb[L + k] = g * asum[b * NH + k / dv];
    } else if (k < NO + Ds) {
      v = sent[(int64_t)b * Ds + (k - NO)];
      if (dsent) dsent[(int64_t)b * Ds + (k - NO)] = dc;
    } else {
      const int pos = padded_pos(k - NO - Ds, dh, HP);
      v = Hb[pos];
      dZ[(int64_t)b * 4 * HP + pos] = dc;
    }
    pb[k] = dl * v * keep2;
  }
  if (tid == 0) pb[L + NO] = dl;
  if (tid >= 1 && L + NO + tid < PW) pb[L + NO + tid] = 0.f;  // (the pitch's padding columns)
  __syncthreads();
  for (int q = tid; q < NH * DP; q += HEAD_T) {
    const int h = q / DP;
    const float* row = BDv + (int64_t)q * NO + h * dv;
    float acc = 0.f;
    for (int i = 0; i < dv; ++i) acc = fmaf(row[i], douts[h * dv + i], acc);
    dz[(int64_t)b * NH * DP + q] = acc;
  }
  for (int h = 0; h < NH; ++h) {
    float s_ = 0.f;
    for (int i = tid; i < dv; i += HEAD_T) s_ = fmaf(douts[h * dv + i], bv[h * dv + i], s_);
    s_ = block_sum_256(s_, red);
    for (int l = tid; l < n; l += HEAD_T) dattn[((int64_t)b * NH + h) * n + l] = s_;
  }
}

// dK[b][0][:] += dZ[b][:] (the gradient of the subgraph's context row that the head reads directly)
__global__ void k_add_row0(float* __restrict__ dK, int64_t ldk_sub, const float* __restrict__ dZ, int Cc, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Cc) return;
  const int b = i / Cc, j = i - b * Cc;
  dK[(int64_t)b * ldk_sub + j] += dZ[i];
}

}  // namespace qagnn

using namespace qagnn;

static int pool_check(const char* who, int32_t B, int32_t n, int32_t NH, int32_t Cc, int32_t ldk, float p) {
  QAGNN_REQUIRE(B > 0 && n > 0 && n <= POOL_MAXN && NH >= 1 && NH <= POOL_MAXH, QAGNN_EINVAL, "%s: B=%d n=%d (<= %d) heads=%d (<= %d)", who, B,
                n, POOL_MAXN, NH, POOL_MAXH);
  QAGNN_REQUIRE(Cc > 0 && Cc % 4 == 0 && Cc <= 256 && ldk % 4 == 0 && ldk >= Cc, QAGNN_EINVAL, "%s: Cc=%d (multiple of 4, <= 256) ldk=%d", who,
                Cc, ldk);
  QAGNN_REQUIRE(p >= 0.f && p < 1.f, QAGNN_EINVAL, "%s: p=%f", who, p);
  return QAGNN_OK;
}

extern "C" int qagnn_pool_attn_fwd_f32(const float* u, const float* cvec, const float* K, int32_t ldk, const uint8_t* mask, int32_t B,
                                       int32_t n, int32_t NH, int32_t Cc, float inv_temp, float p, uint64_t seed, float* attn,
                                       float* attn_d, float* z, qagnn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  QAGNN_REQUIRE(u && cvec && K && mask && attn && attn_d && z, QAGNN_EINVAL, "pool_attn_fwd: null pointer");
  QAGNN_REQUIRE(aligned16(u) && aligned16(K) && aligned16(z), QAGNN_EINVAL, "pool_attn_fwd: u / K / z must be 16-byte aligned");
  if (int rc = pool_check("pool_attn_fwd", B, n, NH, Cc, ldk, p)) return rc;
  k_pool_fwd<<<B, 64 * POOL_W, 0, stream>>>(u, cvec, K, ldk, mask, n, NH, Cc, inv_temp, p, seed, seed_epoch_ptr(), attn, attn_d, z);
  QAGNN_LAUNCH_CHECK("k_pool_fwd");
  return QAGNN_OK;
}

extern "C" int qagnn_pool_attn_bwd_f32(const float* u, const float* K, int32_t ldk, int32_t B, int32_t n, int32_t NH, int32_t Cc,
                                       float inv_temp, float p, uint64_t seed, const float* attn, const float* attn_d, const float* dz,
                                       const float* dattn_d, float* dK, int32_t lddk, float* du, float* dc, qagnn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  QAGNN_REQUIRE(u && K && attn && attn_d && dz && dK && du && dc, QAGNN_EINVAL, "pool_attn_bwd: null pointer");
  QAGNN_REQUIRE(aligned16(u) && aligned16(K) && aligned16(dz) && aligned16(dK) && aligned16(du) && lddk % 4 == 0 && lddk >= Cc, QAGNN_EINVAL,
                "pool_attn_bwd: operands must be 16-byte aligned, lddk=%d", lddk);
  if (int rc = pool_check("pool_attn_bwd", B, n, NH, Cc, ldk, p)) return rc;
  k_pool_bwd<<<B, 64 * POOL_W, 0, stream>>>(u, K, ldk, n, NH, Cc, inv_temp, p, seed, seed_epoch_ptr(), attn, attn_d, dz, dattn_d, dK, lddk, du, dc);
  QAGNN_LAUNCH_CHECK("k_pool_bwd");
  return QAGNN_OK;
}

static int head_check(const char* who, int B, int NH, int DP, int dv, int n, int Ds, int d) {
  QAGNN_REQUIRE(B > 0 && NH >= 1 && NH <= POOL_MAXH && dv > 0 && NH * dv <= HEAD_T && DP > 0 && DP <= 256 && DP % 4 == 0 && n > 0 && Ds >= 0 && d > 0 &&
                    d % 4 == 0 && DP >= d && (DP / 4) >= (d / 4),
                QAGNN_EUNSUPPORTED, "%s: sizes outside what the head kernels take (B=%d NH=%d DP=%d dv=%d n=%d Ds=%d d=%d)", who, B, NH, DP, dv, n, Ds, d);
  return QAGNN_OK;
}

extern "C" int qagnn_head_post_fwd_f32(const float* z, const float* attn, const float* BDv, const float* bv, const float* sent, const float* H,
                                       int64_t ldh, const float* w_fc, const float* b_fc, int32_t B, int32_t NH, int32_t DP, int32_t dv, int32_t n,
                                       int32_t Ds, int32_t d, float p_pool, float p_fc, uint64_t seed_pool, uint64_t seed_fc, float* out, float* asum,
                                       float* logits, qagnn_stream_t stream_) {
  QAGNN_REQUIRE(z && attn && BDv && bv && (sent || Ds == 0) && H && w_fc && b_fc && out && asum && logits, QAGNN_EINVAL, "head_post_fwd: null pointer");
  QAGNN_REQUIRE(p_pool >= 0.f && p_pool < 1.f && p_fc >= 0.f && p_fc < 1.f, QAGNN_EINVAL, "head_post_fwd: dropout probabilities");
  int rc = head_check("head_post_fwd", B, NH, DP, dv, n, Ds, d);
  if (rc != QAGNN_OK) return rc;
  k_head_post_fwd<<<B, HEAD_T, 0, (hipStream_t)stream_>>>(z, attn, BDv, bv, sent, H, ldh, w_fc, b_fc, NH, DP, dv, n, Ds, d, d / 4, DP / 4, p_pool, p_fc,
                                                         seed_pool, seed_fc, seed_epoch_ptr(), out, asum, logits);
  QAGNN_LAUNCH_CHECK("k_head_post_fwd");
  return QAGNN_OK;
}

extern "C" int qagnn_head_post_bwd_f32(const float* dlogits, const float* out, const float* asum, const float* BDv, const float* bv, const float* sent,
                                       const float* H, int64_t ldh, const float* w_fc, int32_t B, int32_t NH, int32_t DP, int32_t dv, int32_t n,
                                       int32_t Ds, int32_t d, float p_pool, float p_fc, uint64_t seed_pool, uint64_t seed_fc, float* dz, float* dattn,
                                       float* dout, float* dsent, float* dZ, float* part, int32_t ldp, qagnn_stream_t stream_) {
  QAGNN_REQUIRE(dlogits && out && asum && BDv && bv && (sent || Ds == 0) && H && w_fc && dz && dattn && dout && dZ && part, QAGNN_EINVAL,
                "head_post_bwd: null pointer");
  int rc = head_check("head_post_bwd", B, NH, DP, dv, n, Ds, d);
  if (rc != QAGNN_OK) return rc;
  QAGNN_REQUIRE(ldp >= NH * dv + Ds + d + NH * dv + 1 && ldp <= NH * dv + Ds + d + NH * dv + 1 + 255, QAGNN_EINVAL, "head_post_bwd: pitch of part");
  k_head_post_bwd<<<B, HEAD_T, 0, (hipStream_t)stream_>>>(dlogits, out, asum, BDv, bv, sent, H, ldh, w_fc, NH, DP, dv, n, Ds, d, d / 4, DP / 4, p_pool, p_fc,
                                                         seed_pool, seed_fc, seed_epoch_ptr(), dz, dattn, dout, dsent, dZ, part, ldp);
  QAGNN_LAUNCH_CHECK("k_head_post_bwd");
  return QAGNN_OK;
}

extern "C" int qagnn_add_row0_f32(float* dK, int64_t ld_sub, const float* dZ, int32_t B, int32_t Cc, qagnn_stream_t stream_) {
  QAGNN_REQUIRE(dK && dZ && B > 0 && Cc > 0, QAGNN_EINVAL, "add_row0: bad arguments");
  k_add_row0<<<cdiv(B * Cc, 256), 256, 0, (hipStream_t)stream_>>>(dK, ld_sub, dZ, Cc, B);
  QAGNN_LAUNCH_CHECK("k_add_row0");
  return QAGNN_OK;
}
