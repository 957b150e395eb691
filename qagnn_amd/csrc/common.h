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
int launch_tn_split(const float* A, int lda, const float* B, int ldb, float* P, int R, int Ka, int No, const float* sc, const float* sh,
                    const int64_t* a_rowidx, int chunk_rows, hipStream_t stream);
int tn_split2_chunk_rows(int R, int Ka1, int Ka2, int No, int lo);
int launch_tn_split2(const float* A1, int lda1, int Ka1, const float* A2, int lda2, int Ka2, const float* B, int ldb, float* P, int R, int No,
                     int chunk_rows, hipStream_t stream);

// NN products, second kernel generation (gemm_nn2.hip): A fragments straight from global memory, B double-buffered in LDS
bool nn2_ok(const qagnn_gemm_nn_args& a, int ldn1, int ldn2);
int launch_nn2(int nt, const qagnn_gemm_nn_args& a, const float* B1n, int ldn1, const float* B2n, int ldn2, hipStream_t stream);
int64_t nn2_pack_bytes(int No, int K1, int K2);
bool nn2_packed_ok(const qagnn_gemm_nn_args& a, int64_t ws_bytes);
const void* nn2_prepack_lookup(const float* B1n, int ldn1, int K1, const float* B2n, int ldn2, int K2, int No);
int launch_nn2_prepacked(int nt, const qagnn_gemm_nn_args& a, const void* pk, hipStream_t stream);
int launch_nn2_packed(int nt, const qagnn_gemm_nn_args& a, const float* B1n, int ldn1, const float* B2n, int ldn2, void* ws, hipStream_t stream);

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

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 fma4(float s, float4 a, float4 acc) {
  return make_float4(fmaf(s, a.x, acc.x), fmaf(s, a.y, acc.y), fmaf(s, a.z, acc.z), fmaf(s, a.w, acc.w));
}

}  // namespace qagnn
