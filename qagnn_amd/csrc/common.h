// Shared helpers for the gfx950 kernels of libqagnn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/qagnn_hip.h"

namespace qagnn {

void set_error(const char* fmt, ...);

// weight-gradient products on the bf16 matrix cores (gemm_split.hip), used by qagnn_gemm_tn_f32's dispatch in gemm.hip
bool tn_split_ok(int R, int Ka, int No, int lda, int ldb, bool gather, bool affine);
int tn_split_chunk_rows(int R, int Ka, int No, int lo);
// amax (both launchers): nullptr, or {max|A1|, max|A2|, max|B|} device words -> the three-MFMA form (gemm_nn2.hip, header)
int launch_tn_split(const float* A, int lda, const float* B, int ldb, float* P, int R, int Ka, int No, const float* sc, const float* sh,
                    const int64_t* a_rowidx, int chunk_rows, hipStream_t stream, const uint32_t* const* amax = nullptr, int np = 2);
int tn_split2_chunk_rows(int R, int Ka1, int Ka2, int No, int lo);
int launch_tn_split2(const float* A1, int lda1, int Ka1, const float* A2, int lda2, int Ka2, const float* B, int ldb, float* P, int R, int No,
                     int chunk_rows, hipStream_t stream, const uint32_t* const* amax = nullptr, int np = 2);

// NN products, second kernel generation (gemm_nn2.hip): A fragments straight from global memory, B double-buffered in LDS
bool nn2_ok(const qagnn_gemm_nn_args& a, int ldn1, int ldn2);
int launch_nn2(int nt, const qagnn_gemm_nn_args& a, const float* B1n, int ldn1, const float* B2n, int ldn2, hipStream_t stream);
// np: the arithmetic form of a packed image -- 3 = exact 3 x bf16 split, 2 = scaled two-piece fp16 split (gemm_nn2.hip, header)
int64_t nn2_pack_bytes(int No, int K1, int K2, int np = 3);
bool nn2_packed_ok(const qagnn_gemm_nn_args& a, int64_t ws_bytes, int np = 3);
bool nn2_h2_ok(const qagnn_gemm_nn_args& a);
const void* nn2_prepack_lookup(const float* B1n, int ldn1, int K1, const float* B2n, int ldn2, int K2, int No, int np = 3);
int launch_nn2_prepacked(int nt, const qagnn_gemm_nn_args& a, const void* pk, hipStream_t stream, int np = 3);
int launch_nn2_packed(int nt, const qagnn_gemm_nn_args& a, const float* B1n, int ldn1, const float* B2n, int ldn2, void* ws, hipStream_t stream, int np = 3);

// producers that can leave max |output| behind for the three-MFMA GEMM form (elementwise.hip; amax = nullptr: plain launch)
int launch_gelu_dropout(const float* X, const float* dY, float* out, int64_t n, float p, uint64_t seed, uint32_t* amax, float* amax_part,
                        hipStream_t stream);
int64_t gelu_amax_scratch_elems(int64_t n);
int launch_bn_relu_bwd_colsum(const float* dR, const float* Hh, float* dH, int ld, int R, int Cc, const float* mean, const float* invstd,
                              const float* scale, const float* shift, const float* gamma, const float* sum_dy, const float* sum_dy_hhat,
                              float inv_rows, const float* roww, float* colsum, float* workspace, uint32_t* amax, hipStream_t stream);
int launch_bn_stats_finalize(const float* part, int n_tiles, int R, int Cc, const float* gamma, const float* beta, float eps, float* stats,
                             float* run_mean, float* run_var, int64_t* nbt, int d, float momentum, float unbias, int ones_col, uint32_t* amax_bound,
                             hipStream_t stream);

int launch_edge_attn_fwd(const qagnn_graph* g, const float* KMQ, int32_t ldk, const float* EkEm, int32_t lde, int32_t HP, float qscale,
                         float* score, float* a, float* alpha, float* aggr, int32_t lda, float* amax_part /* [N] or nullptr */, hipStream_t stream);
int launch_edge_attn_bwd(const qagnn_graph* g, const float* KMQ, int32_t ldk, const float* EkEm, int32_t lde, int32_t HP, float qscale,
                         const float* a, const float* alpha, const float* G, int32_t ldg, float* dKMQ, float* dEkEm, float* ga, float* rs,
                         float* cls_part, float* amax_part /* [3 N] or nullptr */, uint32_t* amax_slot, hipStream_t stream);
int launch_amax_reduce(const float* part, int64_t n, uint32_t* slot, hipStream_t stream);  // max of n non-negative floats -> *slot (elementwise.hip)

// launch timing (timing.hip): a scope at the top of an entry point brackets everything it launches on `s` while qagnn_timing_enable(1)
struct TimedScope {
  int kind;
  hipStream_t s;
  hipEvent_t e0, e1;
  bool counted;
  TimedScope(int kind, hipStream_t s);
  ~TimedScope();
  TimedScope(const TimedScope&) = delete;
  TimedScope& operator=(const TimedScope&) = delete;
};

#define QAGNN_REQUIRE(cond, code, ...) \
  do {                                 \
    if (!(cond)) {                     \
      qagnn::set_error(__VA_ARGS__);   \
      return (code);                   \
    }                                  \
  } while (0)

#define QAGNN_LAUNCH_CHECK(name)                                               \
  do {                                                                         \
    hipError_t e__ = hipGetLastError();                                        \
    if (e__ != hipSuccess) {                                                   \
      qagnn::set_error("%s launch failed: %s", name, hipGetErrorString(e__));  \
      return QAGNN_EHIP;                                                       \
    }                                                                          \
  } while (0)

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// MI355X: 8 XCDs, workgroup b is observed to run on XCD b % 8 (speed only, never correctness).
// Remap so that each XCD walks a contiguous range of logical work items: neighbouring node rows (one
// subgraph = n consecutive rows, all its edges stay inside) then share one L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int b, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7;
  const int xcd = b & 7, slot = b >> 3;
  const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

// The balanced form for the node-side edge kernels (graph_prep.hip: k_xcd_partition): XCD k walks the 4-node blocks
// [base[k], base[k + 1]); the grid is 8 x edge_xcd_cap(N) blocks and a block past its XCD's run has nothing to do (-1).
__host__ __device__ __forceinline__ int edge_xcd_cap(int N) {
  const int per = (((N + 3) >> 2) + 7) >> 3;
  return (per * 5 + 3) >> 2;  // 1.25 x an equal share
}
__device__ __forceinline__ int xcd_remap_balanced(int b, const int* __restrict__ base) {
  const int xcd = b & 7, lb = base[xcd] + (b >> 3);
  return lb < base[xcd + 1] ? lb : -1;
}

// Dropout seeds under hipGraph replay.  Every dropout launch takes its seed BY VALUE, which a captured graph would replay verbatim:
// the same keep masks in every training step.  The kernels therefore mix one device-resident word -- the seed EPOCH, one per device,
// 0 unless a caller advances it -- into the seed; a captured step ends with qagnn_seed_epoch_advance(), so each replay draws new
// masks while forward and backward of one replay still agree.  With the epoch at 0 (eager use) the seeds are used as passed.
const unsigned long long* seed_epoch_ptr();  // device address of the current device's epoch word (elementwise.hip)
__device__ __forceinline__ uint64_t epoch_seed(uint64_t seed, const unsigned long long* __restrict__ epoch) {
  return epoch ? seed + (uint64_t)epoch[0] * 0xD1B54A32D192ED03ull : seed;  // (null: the epoch word could not be resolved -- epoch 0)
}

// counter-based uniform in [0,1): splitmix64 finaliser over (seed, element index); same value in fwd and bwd
__device__ __forceinline__ float uniform01(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}


// Cross-lane reductions.  `__shfl_xor` compiles to ds_bpermute_b32 (an LDS-crossbar round trip, ~100 cycles, with an
// s_waitcnt lgkmcnt(0) behind every step); inside a 16-lane DPP row the same butterfly is four v_add_f32_dpp with
// row_ror:8/4/2/1 (a cyclic rotation inside the row), issued back to back.  After a rotation by r the partial sums are
// periodic with period r and the add is commutative, so all 16 lanes end up with the bit-identical total.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
constexpr int DPP_ROW_ROR = 0x120;  // + n: rotate right by n lanes inside each row of 16

// sum over the 16 lanes of a DPP row (lanes 16g..16g+15); every lane ends up with the total.  All 64 lanes must be active.
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f32<DPP_ROW_ROR + 8>(v);
  v += dpp_f32<DPP_ROW_ROR + 4>(v);
  v += dpp_f32<DPP_ROW_ROR + 2>(v);
  v += dpp_f32<DPP_ROW_ROR + 1>(v);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f32<DPP_ROW_ROR + 8>(v));
  v = fmaxf(v, dpp_f32<DPP_ROW_ROR + 4>(v));
  v = fmaxf(v, dpp_f32<DPP_ROW_ROR + 2>(v));
  v = fmaxf(v, dpp_f32<DPP_ROW_ROR + 1>(v));
  return v;
}

// sum over all 64 lanes of the wave; every lane ends up with the total (rows by DPP, the 4 row totals by two bpermutes)
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

typedef float f32x4s __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));

// ---- the two-piece fp16 GEMM form (NP = 2; gemm_nn2.hip, header): power-of-two operand scales from the bit pattern of max |x| (0 for an all-zero operand; inf / nan: scale 1, the result
// is inf / nan as it would be in fp32).  field = biased exponent of the scale that maps [2^e, 2^(e+1)) onto [2^14, 2^15), clamped so
// that its inverse 254 - field is a normal number too.
__host__ __device__ __forceinline__ uint32_t h2_scale_field(uint32_t amax_bits) {
  const int e = (int)((amax_bits >> 23) & 0xFFu);
  if (e == 0xFF) return 127u;
  int f = 268 - e;  // 2^(14 - (e - 127)) = 2^(f - 127)
  f = f < 1 ? 1 : (f > 253 ? 253 : f);
  return (uint32_t)f;
}
__device__ __forceinline__ float h2_field_to_scale(uint32_t f) { return __builtin_bit_cast(float, f << 23); }
// 1 / (s_a s_b) as a float: exponent fields add; clamped into the normal range (a product that leaves it would have left fp32 too)
__device__ __forceinline__ float h2_inv_scale(uint32_t fa, uint32_t fb) {
  int f = (254 - (int)fa) + (254 - (int)fb) - 127;
  f = f < 1 ? 1 : (f > 254 ? 254 : f);
  return __builtin_bit_cast(float, (uint32_t)f << 23);
}
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// two numbers -> the hi pieces alone (the one-MFMA reduced-precision form, NP = 1)
__device__ __forceinline__ void split1(float x, float y, float s, uint32_t& hi) {
  const f16x2 h = {(_Float16)(x * s), (_Float16)(y * s)};
  hi = __builtin_bit_cast(uint32_t, h);
}
// two numbers -> (hi, lo) pairs packed as fp16x2 dwords; the subtraction is exact (hi is x s rounded to 11 bits)
__device__ __forceinline__ void split2(float x, float y, float s, uint32_t& hi, uint32_t& lo) {
  const float xs = x * s, ys = y * s;
  const f16x2 h = {(_Float16)xs, (_Float16)ys};
  const f16x2 l = {(_Float16)(xs - (float)h[0]), (_Float16)(ys - (float)h[1])};
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}
// the partial products of one tile pair, small terms first; a, b: raw 16-byte fragments [piece] of the A and the B operand.  SWAP: the
// MFMA is issued with its operands exchanged (the transposed output tile: gemm_nn2.hip's direct-store epilogue) -- same products, same
// order, so both orientations give the same bits
#define QAGNN_BF(V) __builtin_bit_cast(bf16x8, V)
#define QAGNN_HF(V) __builtin_bit_cast(f16x8, V)
template <int NP, bool SWAP = false>
__device__ __forceinline__ f32x4s mfma_pieces(const u32x4s* __restrict__ a, const u32x4s* __restrict__ b, f32x4s c) {
#define QAGNN_MF3(I, J) c = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(QAGNN_BF(b[J]), QAGNN_BF(a[I]), c, 0, 0, 0) \
                                 : __builtin_amdgcn_mfma_f32_16x16x32_bf16(QAGNN_BF(a[I]), QAGNN_BF(b[J]), c, 0, 0, 0);
#define QAGNN_MF2(I, J) c = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_f16(QAGNN_HF(b[J]), QAGNN_HF(a[I]), c, 0, 0, 0) \
                                 : __builtin_amdgcn_mfma_f32_16x16x32_f16(QAGNN_HF(a[I]), QAGNN_HF(b[J]), c, 0, 0, 0);
  if constexpr (NP == 3) {
    QAGNN_MF3(2, 0) QAGNN_MF3(0, 2) QAGNN_MF3(1, 1) QAGNN_MF3(1, 0) QAGNN_MF3(0, 1) QAGNN_MF3(0, 0)
  } else if constexpr (NP == 2) {
    QAGNN_MF2(1, 0) QAGNN_MF2(0, 1) QAGNN_MF2(0, 0)
  } else {  // NP == 1: the reduced-precision form -- operands rounded to fp16 (11 significant bits) under the same scales, fp32 accumulation
    QAGNN_MF2(0, 0)
  }
#undef QAGNN_MF3
#undef QAGNN_MF2
  return c;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 fma4(float s, float4 a, float4 acc) {
  return make_float4(fmaf(s, a.x, acc.x), fmaf(s, a.y, acc.y), fmaf(s, a.z, acc.z), fmaf(s, a.w, acc.w));
}

// Operand maxima for the three-MFMA GEMM form (gemm_nn2.hip).  Agent-scope atomics are executed at the memory side on this chip (the
// eight L2s are not coherent with each other): ~4 ns EACH and serialised per address -- one atomic per wave of a 13 000-block elementwise
// launch (52 000 of them) doubled the whole training step (round 6, visit 2).  So: at most ~1 000 atomics per launch (fat blocks, one
// atomic per BLOCK), and the one-wave-per-node edge kernels store per-node maxima with plain stores that a 64-block kernel reduces.
// The wave reduction stays in the DPP network: rows by row_ror, then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3
// leave the total in LANE 63 (no LDS round trip).  All 64 lanes must be active; m >= 0.
__device__ __forceinline__ float wave_amax_lane63(float m) {
  m = row16_max(m);
  int b = __builtin_bit_cast(int, m);
  m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(b, b, 0x142, 0xA, 0xF, false)));
  b = __builtin_bit_cast(int, m);
  m = fmaxf(m, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(b, b, 0x143, 0xC, 0xF, false)));
  return m;
}
// max over the block's waves (<= 16), merged into *amax by ONE integer atomic max on the bit pattern (order-independent, hence
// deterministic).  Every thread of the block must call it (it holds a barrier); red: >= 16 floats of LDS.
__device__ __forceinline__ void block_amax_merge(float m, uint32_t* __restrict__ amax, float* __restrict__ red) {
  m = wave_amax_lane63(m);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if (lane == 63) red[w] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    if (t > 0.f) __hip_atomic_fetch_max(amax, __builtin_bit_cast(uint32_t, t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ __forceinline__ float absmax4(float4 v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }


}  // namespace qagnn
