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
