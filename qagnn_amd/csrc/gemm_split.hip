// fp32 GEMM on the bf16 matrix cores by exact operand splitting ("3 x bf16").
//
// The fp32-input MFMAs of csrc/gemm.hip run at the fp32 VECTOR rate (157 TFLOP/s, 1/16 of the bf16 MFMA rate), and the dense
// node-side GEMMs are 63 % of a training step.  A fp32 number is the EXACT sum of three bf16 numbers,
//     x = h1 + h2 + h3,   h1 = trunc8(x),  h2 = trunc8(x - h1),  h3 = x - h1 - h2        (24 = 8 + 8 + 8 significant bits),
// every subtraction being exact, so  a*b = sum_{p,q} a_p b_q  and each a_p*b_q is exact in fp32 (8 x 8 bits).  Keeping the six
// terms of order >= 2^-16 (11, 12, 21, 22, 13, 31) and accumulating them in the MFMA's fp32 accumulators drops only the terms
// 23, 32, 33: a relative error <= 2^-23 per product, the size of ONE fp32 rounding -- the same class of error the fp32 MFMA (or any
// other summation order) has.  Six bf16 MFMAs at 16x the fp32 rate = 2.67x the fp32-MFMA throughput; the operands are split once
// per tile on their way into LDS (5 integer/float VALU operations per element).  The result is NOT bit-identical to gemm.hip's
// (no two summation orders are); the parity tests hold this path to the same float64 yardstick.  QAGNN_GEMM_SPLIT=0 pins the
// fp32-MFMA kernels.
//
// Kernel shape: C[M][No] (+)= [A1|A2] * [B1;B2] + bias + rowtab[rowidx], as k_gemm_nn, but B comes in its [No][K] layout (k
// contiguous: the MFMA wants 8 consecutive k per lane for both operands; the stack holds every weight in both layouts).  Block =
// 4 waves, 128 rows x NT*16 columns, wave = 32 rows (2 row tiles) x NT column tiles of v_mfma_f32_16x16x32_bf16, k-tile = 32.
// LDS: three bf16 images of the A tile [128][32] and of the B tile [NT*16][32], 16-byte chunks XOR-swizzled by (row >> 2) & 3 so
// that the 16 lanes of a fragment read hit 16 different bank groups; single-buffered (the next tile's global loads are in flight,
// in registers, under the MFMAs), two blocks per CU.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace qagnn {


// 16-byte load through a buffer descriptor: per-lane byte offset + wave-uniform byte offset; out of range reads return zeros
__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff = 0) {
  return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0));
}

constexpr int SBK = 32, SBM = 128, SWAVES = 4, SRT = 2, STHR = SWAVES * 64;

__device__ __forceinline__ void split3(float x, uint32_t& h1, uint32_t& h2, uint32_t& h3) {
  const uint32_t u = __builtin_bit_cast(uint32_t, x);
  h1 = u & 0xFFFF0000u;
  const float r1 = x - __builtin_bit_cast(float, h1);  // exact
  h2 = __builtin_bit_cast(uint32_t, r1) & 0xFFFF0000u;
  const float r2 = r1 - __builtin_bit_cast(float, h2);  // exact, <= 8 significant bits
  h3 = __builtin_bit_cast(uint32_t, r2);  // <= 8 significant bits: the low half is already zero
}
// two fp32 bit patterns whose low halves are zero -> one dword of two bf16 (element 0 = lo, in the low half)
__device__ __forceinline__ uint32_t pack_hi(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }

// float4 (4 consecutive k of one row) -> the three bf16 images, 8 bytes each, at element offset `off` (already swizzled)
__device__ __forceinline__ void store_split(uint16_t* __restrict__ img, int img_elems, int off, float4 v) {
  uint32_t a1, a2, a3, b1, b2, b3, c1, c2, c3, d1, d2, d3;
  split3(v.x, a1, a2, a3);
  split3(v.y, b1, b2, b3);
  split3(v.z, c1, c2, c3);
  split3(v.w, d1, d2, d3);
  *reinterpret_cast<uint2*>(img + off) = make_uint2(pack_hi(a1, b1), pack_hi(c1, d1));
  *reinterpret_cast<uint2*>(img + img_elems + off) = make_uint2(pack_hi(a2, b2), pack_hi(c2, d2));
  *reinterpret_cast<uint2*>(img + 2 * img_elems + off) = make_uint2(pack_hi(a3, b3), pack_hi(c3, d3));
}
// timing ablations only (QAGNN_ABL_NO?SPLIT): the three stores without the split arithmetic
__device__ __forceinline__ void store_nosplit(uint16_t* __restrict__ img, int img_elems, int off, float4 v) {
  const uint2 h = make_uint2(pack_hi(__builtin_bit_cast(uint32_t, v.x), __builtin_bit_cast(uint32_t, v.y)),
                             pack_hi(__builtin_bit_cast(uint32_t, v.z), __builtin_bit_cast(uint32_t, v.w)));
  *reinterpret_cast<uint2*>(img + off) = h;
  *reinterpret_cast<uint2*>(img + img_elems + off) = h;
  *reinterpret_cast<uint2*>(img + 2 * img_elems + off) = h;
}
// element offset of (row, k) in a swizzled [rows][32] bf16 image
// The XOR pattern follows ds_read_b128's lane groups (MI355X_MICROARCH.md, LDS): a fragment read (lane -> row lane & 15, chunk lane >> 4)
// is serviced in four NON-contiguous 16-lane groups, e.g. {0-3, 12-15, 20-27} = rows 0-3, 12-15 of chunk c with rows 4-11 of chunk
// c+1.  Rows 64 B apart share a 16-bank quarter, so the four rows r, r+4, r+8, r+12 of a group must land in four different 16-byte
// slots: slot = chunk ^ G[(row >> 2) & 3] with G = {0, 2, 3, 1} does that for every group (with G = identity SQ_LDS_BANK_CONFLICT
// was 35 % of the LDS cycles).
__device__ __forceinline__ int swz(int row, int k) {
  const int g = (0x78 >> (((row >> 2) & 3) << 1)) & 3;
  return row * SBK + ((((k >> 3) ^ g) & 3) << 3) + (k & 7);
}

// QAGNN_NN_TRACE (tools/nn_trace.hip only): cycle stamps of one block's waves at the phase boundaries of every k-tile
#ifdef QAGNN_NN_TRACE
__device__ unsigned long long g_nn_trace[4][40][7];
#define NN_STAMP(kt, ph)                                                                                                   \
  do {                                                                                                                     \
    if (blockIdx.x == QAGNN_NN_TRACE && vb == (int)blockIdx.x && lane == 0 && (kt) < 40) g_nn_trace[w][kt][ph] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define NN_STAMP(kt, ph)
#endif

// FLAT: operands addressed with 64-bit pointers (row gather through a_rowidx, or an operand of 2 GB and more); otherwise every
// load goes through a buffer descriptor: per-thread 32-bit row offsets fixed per output tile, the k position as the instruction's
// scalar offset, rows past M / columns past No / k past K answered with zeros by the bounds check -- the 64-bit address arithmetic
// and the zeroing selects were 100 of the 468 VALU instructions per wave and k-tile (profiles/r2_run57_nn_trace.txt).
// STATS: the epilogue additionally leaves, per 128-row tile and output column, x0 = the tile's first value, S1 = sum (x - x0) and
// S2 = sum (x - x0)^2 over the tile's rows of C in a.colstat_part (BatchNorm batch statistics as a by-product of the GEMM that
// produces the BatchNorm input; the two stand-alone passes over h1 they replace cost 35 us + 4 launches per layer).
template <int NT, bool AFFINE, bool FLAT, bool STATS = false>
__global__ __launch_bounds__(STHR) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_gemm_nn_split(qagnn_gemm_nn_args a, const float* __restrict__ B1n,
                                                                                           int ldn1, const float* __restrict__ B2n,
                                                                                           int ldn2, int ntiles) {
  constexpr int BN = NT * 16;
  constexpr int A_EL = SBM * SBK, B_EL = BN * SBK;            // elements per bf16 image
  constexpr int A_IT = SBM * (SBK / 4) / STHR;                // float4 per thread and tile: 4
  constexpr int B_IT = (BN * (SBK / 4) + STHR - 1) / STHR;    // 7 at NT = 13
  constexpr int PS = BN + 4, SLAB_ROWS = 16;
  constexpr int KLOOP_B = (3 * A_EL + 3 * B_EL) * 2, STAGE_B = SWAVES * SLAB_ROWS * PS * 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[KLOOP_B > STAGE_B ? KLOOP_B : STAGE_B];
  uint16_t* const As = reinterpret_cast<uint16_t*>(smem_raw);
  uint16_t* const Bs = As + 3 * A_EL;
  __shared__ float cstat[STATS ? SWAVES : 1][2][STATS ? BN : 1];  // per-wave column sums of a tile, combined in wave order

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ncb = (a.No + BN - 1) / BN;
  const int nk1 = (a.K1 + SBK - 1) / SBK, nkt = nk1 + (a.K2 + SBK - 1) / SBK;
  const int lr = tid >> 3, kq = tid & 7;  // tile row (+ 32 per pass) and float4 column of this thread's loads

  for (int vb = blockIdx.x; vb < ntiles; vb += gridDim.x) {
    const int tile = a.xcd_remap ? xcd_remap(vb, ntiles) : vb;
    const int m0 = (tile / ncb) * SBM, n0 = (tile % ncb) * BN;
    f32x4s acc[SRT][NT];
#pragma unroll
    for (int i = 0; i < SRT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4s){0.f, 0.f, 0.f, 0.f};

    int64_t arow[A_IT];  // source row of A1 for this thread's tile rows (gathered or identity); -1 = zero row
#pragma unroll
    for (int p = 0; p < A_IT; ++p) {
      const int row = m0 + lr + p * 32;
      arow[p] = row < a.M ? (a.a_rowidx ? a.a_rowidx[row] : (int64_t)row) : -1;
    }
    float4 ra[A_IT], rb[B_IT];
    constexpr uint32_t OOB = 0x80000000u;
    auto gload_buf = [&](int kt) {
      const bool first = kt < nk1;
      const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(first ? a.A1 : a.A2), 0, a.M * (first ? a.lda1 : a.lda2) * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(first ? B1n : B2n), 0, a.No * (first ? ldn1 : ldn2) * 4, 0x00020000);
      const uint32_t lda4 = (uint32_t)(first ? a.lda1 : a.lda2) * 4u, ldn4 = (uint32_t)(first ? ldn1 : ldn2) * 4u;
      const int K = first ? a.K1 : a.K2, kb = (first ? kt : kt - nk1) * SBK;
      const bool kin = kb + kq * 4 < K;  // k past K: the whole float4 is outside (K % 4 == 0)
      const uint32_t soff = (uint32_t)kb * 4u;
      // ONE select per load: a slot outside in the row (column) direction, in k, or in both gets the out-of-range offset (two
      // separate OOB terms would add up to 0x80000000 + 0x80000000 = 0 and read row 0 instead of zeros)
#pragma unroll
      for (int p = 0; p < A_IT; ++p) {
        const int row = m0 + lr + p * 32;
        ra[p] = bload4(rA, (kin && row < a.M) ? (uint32_t)row * lda4 + (uint32_t)kq * 16u : OOB, soff);
      }
#pragma unroll
      for (int q = 0; q < B_IT; ++q) {
        const int col = n0 + lr + q * 32;
        rb[q] = bload4(rB, (kin && col < a.No) ? (uint32_t)col * ldn4 + (uint32_t)kq * 16u : OOB, soff);
      }
    };
    // FLAT: unconditional loads from clamped addresses (branches around loads make hipcc fall back to vmcnt(0)); what lies past the
    // operand -- rows >= M, columns >= No, k >= K -- is zeroed with selects when the tile goes to LDS
    auto gload_flat = [&](int kt) {
      const bool first = kt < nk1;
      const float* A = first ? a.A1 : a.A2;
      const float* Bn = first ? B1n : B2n;
      const int lda = first ? a.lda1 : a.lda2, ldn = first ? ldn1 : ldn2, K = first ? a.K1 : a.K2;
      const int k0 = (first ? kt : kt - nk1) * SBK + kq * 4, kc = min(k0, K - 4);
#pragma unroll
      for (int p = 0; p < A_IT; ++p) {
        const int64_t srow = first ? (arow[p] >= 0 ? arow[p] : 0) : (int64_t)min(m0 + lr + p * 32, a.M - 1);
        ra[p] = ld4(A + srow * lda + kc);
      }
#pragma unroll
      for (int q = 0; q < B_IT; ++q) rb[q] = ld4(Bn + (int64_t)min(n0 + lr + q * 32, a.No - 1) * ldn + kc);
    };
    auto gload = [&](int kt) {
      if constexpr (FLAT) gload_flat(kt);
      else gload_buf(kt);
    };
    auto lstore = [&](int kt) {
      const bool first = kt < nk1;
      const int K = first ? a.K1 : a.K2;
      const int kl = kq * 4, k0 = (first ? kt : kt - nk1) * SBK + kl;
      const bool kin = k0 < K;  // K is a multiple of 4: a float4 is inside or outside as a whole
#pragma unroll
      for (int p = 0; p < A_IT; ++p) {
        float4 v = ra[p];
        if (AFFINE && first) {
          const int kc = min(k0, K - 4);
          const float4 sc = ld4(a.a_scale + kc), sh = ld4(a.a_shift + kc);
          v.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f);
          v.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
          v.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f);
          v.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
        }
        if constexpr (FLAT) {
          const bool ok = kin && (!first || arow[p] >= 0) && (first || m0 + lr + p * 32 < a.M);
          v = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }  // (buffer loads: zeros already; under AFFINE an out-of-range k leaves relu(shift) here, multiplied by B's zeros)
#ifdef QAGNN_ABL_NOASPLIT  // timing ablation (tools/): no split arithmetic for A -- the high halves go to all three images
        store_nosplit(As, A_EL, swz(lr + p * 32, kl), v);
#else
        store_split(As, A_EL, swz(lr + p * 32, kl), v);
#endif
      }
#pragma unroll
      for (int q = 0; q < B_IT; ++q) {
        const int col = lr + q * 32;
        if (col < BN) {
          const float4 v = (!FLAT || (kin && n0 + col < a.No)) ? rb[q] : make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef QAGNN_ABL_NOBSPLIT  // timing ablation (tools/): what a weight operand that arrives pre-split would save
          store_nosplit(Bs, B_EL, swz(col, kl), v);
#else
          store_split(Bs, B_EL, swz(col, kl), v);
#endif
        }
      }
    };

    gload(0);
    for (int kt = 0; kt < nkt; ++kt) {
      NN_STAMP(kt, 0);
      __syncthreads();  // the previous tile's fragment reads (or the previous output tile's slab reads) are done
      NN_STAMP(kt, 1);
#ifdef QAGNN_NN_TRACE
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the tile's global loads have landed (separates their latency from the split)
      NN_STAMP(kt, 6);
#endif
      lstore(kt);
      NN_STAMP(kt, 2);
      __syncthreads();
      NN_STAMP(kt, 3);
      gload(min(kt + 1, nkt - 1));  // in flight under the MFMAs; past the last tile: a redundant reload, never stored
      NN_STAMP(kt, 4);
      // the wave in its MFMA phase goes first at the SIMD's issue port (over the co-resident block's wave, which is in its load / split
      // phase): 3-6 % on the products alone, 0.1-0.4 % on the step (profiles/r3_run36_setprio.txt)
      __builtin_amdgcn_s_setprio(1);
      bf16x8 af[SRT][3];
#pragma unroll
      for (int i = 0; i < SRT; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          af[i][p] = *reinterpret_cast<const bf16x8*>(As + p * A_EL + swz(w * 32 + i * 16 + (lane & 15), (lane >> 4) * 8));
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        bf16x8 bf[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) bf[p] = *reinterpret_cast<const bf16x8*>(Bs + p * B_EL + swz(j * 16 + (lane & 15), (lane >> 4) * 8));
#pragma unroll
        for (int i = 0; i < SRT; ++i) {  // small terms first
          f32x4s c = acc[i][j];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][2], bf[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][1], bf[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][1], bf[0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[0], c, 0, 0, 0);
          acc[i][j] = c;
        }
      }
      __builtin_amdgcn_s_setprio(0);
      NN_STAMP(kt, 5);
    }

    // epilogue: transpose 16 rows at a time through the wave's LDS slab, then whole-row 16-byte stores (as k_gemm_nn)
    float* const St = reinterpret_cast<float*>(smem_raw) + w * SLAB_ROWS * PS;
    constexpr int ROW_F4 = BN / 4, TILE_F4 = SLAB_ROWS * ROW_F4, ST_IT = (TILE_F4 + 63) / 64;
    constexpr int SC = (BN + 63) / 64;  // STATS: columns lane, lane + 64, ... of the tile per lane
    float x0r[SC], s1[SC], s2[SC];
#pragma unroll
    for (int cc = 0; cc < SC; ++cc) x0r[cc] = s1[cc] = s2[cc] = 0.f;
#pragma unroll
    for (int i = 0; i < SRT; ++i) {
      __syncthreads();  // k-loop reads (first pass) / the previous pass's slab reads are done before the slab is overwritten
      const int lr0 = (lane >> 4) * 4;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) St[(lr0 + r) * PS + j * 16 + (lane & 15)] = acc[i][j][r];
      __syncthreads();
      if constexpr (STATS) {
        // every wave's slab is complete: row 0 of wave 0's first slab is the tile's first row (the shift of this tile's sums)
        const float* const S0 = reinterpret_cast<const float*>(smem_raw);
#pragma unroll
        for (int cc = 0; cc < SC; ++cc) {
          const int c = lane + cc * 64;
          if (c < BN && n0 + c < a.No) {
            const float bc = a.bias ? a.bias[n0 + c] : 0.f;
            if (i == 0) x0r[cc] = S0[c] + bc;
            const int rows = a.M - (m0 + (w * SRT + i) * 16);  // rows of this slab inside the matrix (<= 0 past M)
            float v[SLAB_ROWS];
#pragma unroll
            for (int r = 0; r < SLAB_ROWS; ++r) v[r] = St[r * PS + c];  // 16 LDS reads in flight, then the arithmetic
#pragma unroll
            for (int r = 0; r < SLAB_ROWS; ++r) {
              const float dlt = r < rows ? (v[r] + bc) - x0r[cc] : 0.f;  // the value the store loop below writes, minus the shift
              s1[cc] += dlt;
              s2[cc] = fmaf(dlt, dlt, s2[cc]);
            }
          }
        }
      }
#pragma unroll
      for (int it = 0; it < ST_IT; ++it) {
        const int idx = lane + it * 64;
        if (idx >= TILE_F4) break;
        const int slr = idx / ROW_F4, c4 = idx % ROW_F4;
        const int row = m0 + (w * SRT + i) * 16 + slr, col = n0 + c4 * 4;
        if (row >= a.M || col >= a.No) continue;
        float4 v = ld4(St + slr * PS + c4 * 4);
        if (a.bias) v = add4(v, ld4(a.bias + col));
        if (a.rowtab) v = add4(v, ld4(a.rowtab + (int64_t)a.rowidx[row] * a.ldt + col));
        float* dst = a.C + (int64_t)row * a.ldc + col;
        if (a.accumulate) v = add4(v, ld4(dst));
        st4(dst, v);
      }
    }
    if constexpr (STATS) {
#pragma unroll
      for (int cc = 0; cc < SC; ++cc) {
        const int c = lane + cc * 64;
        if (c < BN) { cstat[w][0][c] = s1[cc]; cstat[w][1][c] = s2[cc]; }
      }
      __syncthreads();
      if (tid < BN && n0 + tid < a.No) {  // thread (w, lane) = column w * 64 + lane: its own x0r[w] is that column's shift
        float x0c = x0r[0];
#pragma unroll
        for (int cc = 1; cc < SC; ++cc) x0c = (w == cc) ? x0r[cc] : x0c;
        float* const pt = a.colstat_part + (int64_t)(tile / ncb) * 3 * a.No + n0 + tid;
        pt[0] = x0c;
        pt[a.No] = (cstat[0][0][tid] + cstat[1][0][tid]) + (cstat[2][0][tid] + cstat[3][0][tid]);
        pt[2 * a.No] = (cstat[0][1][tid] + cstat[1][1][tid]) + (cstat[2][1][tid] + cstat[3][1][tid]);
      }
      // (cstat is rewritten only after the next tile's k-loop and slab barriers)
    }
  }
}

static int split_num_cus() {
  static int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    return v;
  }();
  return n;
}

template <int NT>
static int launch_split(const qagnn_gemm_nn_args& a, const float* B1n, int ldn1, const float* B2n, int ldn2, hipStream_t stream) {
  qagnn_gemm_nn_args b = a;
  b.xcd_remap = 1;
  const int ntiles = cdiv(a.No, NT * 16) * cdiv(a.M, SBM);
  const int cap = (split_num_cus() * 2) & ~7;
  const int grid = ntiles < cap ? ntiles : cap;
  const int64_t lim = (int64_t)0x7FFFFFFF;
  const bool flat = a.a_rowidx || (int64_t)a.M * a.lda1 * 4 >= lim || (int64_t)a.M * a.lda2 * 4 >= lim ||
                    (int64_t)a.No * ldn1 * 4 >= lim || (int64_t)a.No * ldn2 * 4 >= lim;
  if constexpr (NT == 13 || NT == 7 || NT == 4 || NT == 2) {
    if (a.colstat_part) {  // (validated by the entry point: bias-only epilogue, no gather, 32-bit offsets)
      k_gemm_nn_split<NT, false, false, true><<<grid, STHR, 0, stream>>>(b, B1n, ldn1, B2n, ldn2, ntiles);
      QAGNN_LAUNCH_CHECK("k_gemm_nn_split<stats>");
      return QAGNN_OK;
    }
  }
  if (flat) {
    if (a.a_scale) k_gemm_nn_split<NT, true, true><<<grid, STHR, 0, stream>>>(b, B1n, ldn1, B2n, ldn2, ntiles);
    else k_gemm_nn_split<NT, false, true><<<grid, STHR, 0, stream>>>(b, B1n, ldn1, B2n, ldn2, ntiles);
  } else {
    if (a.a_scale) k_gemm_nn_split<NT, true, false><<<grid, STHR, 0, stream>>>(b, B1n, ldn1, B2n, ldn2, ntiles);
    else k_gemm_nn_split<NT, false, false><<<grid, STHR, 0, stream>>>(b, B1n, ldn1, B2n, ldn2, ntiles);
  }
  QAGNN_LAUNCH_CHECK("k_gemm_nn_split");
  return QAGNN_OK;
}


// ------------------------------------------------------------------------------------------------------------
// TN (weight gradients): P[chunk][Ka][No] = sum_{r in chunk} A[r][ka] * B[r][no], the same exact 3 x bf16 products.
//
// Both operands arrive with the reduction index r as the SLOW index, while the MFMA wants 8 consecutive r per lane, so the tiles are
// transposed on their way into LDS: a task = 8 rows x one float4 column (buffer loads: 16-lane row segments of 256 B, rows past R
// and columns past the operand come back as zeros from the descriptor's bounds check -- no clamps, no selects); the 32 numbers are
// split and written, per column, as one 16-byte chunk per piece.  LDS holds [piece][column][32 rows] bf16 with a column pitch of
// 80 B (32 rows + 8 pad) and the 16-byte slot of row group g stored at g ^ s(column), s = bit 2 ^ bit 3 of the column: with that,
// the 8-lane groups of ds_write_b128 (2 float4 columns x 4 row groups) and the four 16-lane groups of the ds_read_b128 fragment
// reads (MI355X_MICROARCH.md, LDS) each touch every bank once (searched exhaustively; pitch 80 B alone left 34 % of the LDS
// cycles as conflicts).
//
// Block = 4 waves, one chunk of rows, KT*16 x NT*16 outputs with (KT, NT) = (13, 7) or (7, 13): 77 KB of LDS and <= 256 VGPRs, so
// TWO blocks per CU -- one block's split/transpose (VALU) runs under the other's MFMAs, which a single block with its
// barrier-separated phases cannot do (v1: 208 x 208 per block, one block per CU, 26 % MFMA-busy).  The 320 tasks of a k-tile are
// six task-waves: the four of the wide operand stay with their wave, the two of the narrow one rotate over the waves tile by tile.
// Wave w owns the 16-row output strips w, w+4, ... of the first 4*(KT/4) strips with all NT column tiles (A fragments stay in
// registers across the column loop); the tiles of the KT%4 leftover strips are dealt out one by one (tile q -> wave q % 4).
// ------------------------------------------------------------------------------------------------------------
constexpr int TKR = 32, TCP = 40, TTHR = 256;

// 8 rows of one column -> NP 16-byte chunks (NP == 2: the scaled fp16 pieces of x s, common.h / gemm_nn2.hip's header)
template <int NP>
__device__ __forceinline__ void store_col8(uint16_t* __restrict__ dst, int img_elems, const float (&x)[8], float s) {
  if constexpr (NP == 2) {
    uint32_t ph[4], pl[4];
#pragma unroll
    for (int e = 0; e < 8; e += 2) split2(x[e], x[e + 1], s, ph[e >> 1], pl[e >> 1]);
    *reinterpret_cast<uint4*>(dst) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
    *reinterpret_cast<uint4*>(dst + img_elems) = make_uint4(pl[0], pl[1], pl[2], pl[3]);
    return;
  }
  if constexpr (NP == 1) {  // (the one-MFMA reduced-precision form: x s rounded to fp16)
    uint32_t ph[4];
#pragma unroll
    for (int e = 0; e < 8; e += 2) split1(x[e], x[e + 1], s, ph[e >> 1]);
    *reinterpret_cast<uint4*>(dst) = make_uint4(ph[0], ph[1], ph[2], ph[3]);
    return;
  }
  uint32_t h1[8], h2[8], h3[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) split3(x[i], h1[i], h2[i], h3[i]);
  *reinterpret_cast<uint4*>(dst) = make_uint4(pack_hi(h1[0], h1[1]), pack_hi(h1[2], h1[3]), pack_hi(h1[4], h1[5]), pack_hi(h1[6], h1[7]));
  *reinterpret_cast<uint4*>(dst + img_elems) = make_uint4(pack_hi(h2[0], h2[1]), pack_hi(h2[2], h2[3]), pack_hi(h2[4], h2[5]), pack_hi(h2[6], h2[7]));
  *reinterpret_cast<uint4*>(dst + 2 * img_elems) = make_uint4(pack_hi(h3[0], h3[1]), pack_hi(h3[2], h3[3]), pack_hi(h3[4], h3[5]), pack_hi(h3[6], h3[7]));
}

// a task's 8 x 4 numbers -> LDS (4 columns); AFF: relu(x * sc + sh) per column first
template <bool AFF, int NP>
__device__ __forceinline__ void store_task(uint16_t* __restrict__ img, int img_elems, int off, const float4 (&r)[8], float4 sc, float4 sh, float s) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v = j == 0 ? r[i].x : j == 1 ? r[i].y : j == 2 ? r[i].z : r[i].w;
      if (AFF) {
        const float s = j == 0 ? sc.x : j == 1 ? sc.y : j == 2 ? sc.z : sc.w, h = j == 0 ? sh.x : j == 1 ? sh.y : j == 2 ? sh.z : sh.w;
        v = fmaxf(fmaf(v, s, h), 0.f);
      }
      x[i] = v;
    }
    store_col8<NP>(img + off + j * TCP, img_elems, x, s);
  }
}

// GATHER: row r of A is A[a_rowidx[r]] (negative = zero row): the frozen entity table under cpt_transform's weight gradient.  The
// table can exceed 2 GB, so these rows come through 64-bit flat loads; their indices are fetched one tile ahead, behind the stores.
// NP == 2: the three-MFMA form; amax[0..2] = the words holding max |A|, max |A2|, max |B| (bit patterns)
struct TnAmax { const uint32_t* a; const uint32_t* a2; const uint32_t* b; };
template <int KT, int NT, bool AFFINE, bool GATHER, int NP = 3>
__global__ __launch_bounds__(TTHR) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_gemm_tn_split(
    const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ P, int R, int Ka, int No,
    const float* __restrict__ a_scale, const float* __restrict__ a_shift, int chunk_rows, const int64_t* __restrict__ a_rowidx,
    const float* __restrict__ A2, int lda2, int Ka2, TnAmax amax) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_tn[];
  constexpr int AC = KT * 16, BC = NT * 16, A_EL = AC * TCP, B_EL = BC * TCP;
  constexpr int MT = KT / 4, REM = KT % 4, RS = (REM * NT + 3) / 4;  // full strips per wave; leftover strips, shared tile by tile
  constexpr int A_TW = (AC + 63) / 64, B_TW = (BC + 63) / 64;
  constexpr bool A_MAJOR = A_TW >= B_TW;                              // the wide operand: one task-wave per wave, every tile
  constexpr int MAJ_TW = A_MAJOR ? A_TW : B_TW, MIN_TW = A_MAJOR ? B_TW : A_TW;
  constexpr int MAJ_C = A_MAJOR ? AC : BC, MIN_C = A_MAJOR ? BC : AC, MAJ_EL = A_MAJOR ? A_EL : B_EL, MIN_EL = A_MAJOR ? B_EL : A_EL;
  static_assert(MAJ_TW <= 4 && MIN_TW <= 4, "task-waves per operand");
  static_assert(NP == 3 || !GATHER, "the gathered table goes through the exact split");
  uint16_t* const As = reinterpret_cast<uint16_t*>(smem_tn);
  uint16_t* const Bs = As + NP * A_EL;
  uint16_t* const Maj = A_MAJOR ? As : Bs;
  uint16_t* const Min = A_MAJOR ? Bs : As;
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // Two A operands (A2 != nullptr): the product [A | A2]^T B, rows [0, Ka) of the output from A, rows [Ka, Ka + Ka2) from A2 -- the
  // row tiles of A2 follow those of A in blockIdx.y, and such a block simply works on the other operand (one launch and one chunk sum
  // for the two weight gradients that share dK|dM|dQ: X^T dKMQ and S^T dKMQ)
  const int ka_total = Ka + (A2 ? Ka2 : 0);
  // XCD-aware block order (chunk_rows < 0: off; see k_gemm_tn_ws): the blocks of one chunk on one XCD
  int bx = blockIdx.x, by = blockIdx.y, chunk = blockIdx.z, row_shift = 0;
  if (chunk_rows < 0) {
    chunk_rows = -chunk_rows;
  } else {
    const int lin = bx + (int)gridDim.x * (by + (int)gridDim.y * chunk);
    const int v = xcd_remap(lin, (int)(gridDim.x * gridDim.y * gridDim.z));
    bx = v % (int)gridDim.x;
    by = (v / (int)gridDim.x) % (int)gridDim.y;
    chunk = v / (int)(gridDim.x * gridDim.y);
  }
  bool second = false;
  if (A2 != nullptr) {
    const int n1 = (Ka + AC - 1) / AC;
    if (by >= n1) { by -= n1; row_shift = Ka; A = A2; lda = lda2; Ka = Ka2; second = true; }
  }
  uint32_t fa = 127u, fb = 127u;
  float sa = 1.f, sb = 1.f;
  if constexpr (NP <= 2) {
    fa = __builtin_amdgcn_readfirstlane(h2_scale_field(second ? amax.a2[0] : amax.a[0]));
    fb = __builtin_amdgcn_readfirstlane(h2_scale_field(amax.b[0]));
    sa = h2_field_to_scale(fa);
    sb = h2_field_to_scale(fb);
  }
  const int n0 = bx * BC, m0 = by * AC;
  const int r_beg = chunk * chunk_rows, r_end = min(R, r_beg + chunk_rows);
  const int ntile = (r_end - r_beg + TKR - 1) / TKR;  // chunk_rows is a multiple of TKR: only the last chunk has a ragged tile, past R

  static_assert(!GATHER || A_MAJOR, "the gathered operand is handled as the wide one");
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, GATHER ? 0 : R * lda * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, R * ldb * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsMaj = A_MAJOR ? rsA : rsB, rsMin = A_MAJOR ? rsB : rsA;
  const int maj0 = A_MAJOR ? m0 : n0, min0 = A_MAJOR ? n0 : m0, majDim = A_MAJOR ? Ka : No, minDim = A_MAJOR ? No : Ka;
  const uint32_t ldMaj4 = (uint32_t)(A_MAJOR ? lda : ldb) * 4u, ldMin4 = (uint32_t)(A_MAJOR ? ldb : lda) * 4u;
  constexpr uint32_t OOB = 0x80000000u;  // beyond any operand this kernel is launched on: the load returns zeros

  const int g = lane & 3;                                   // row group (8 rows) of this thread's tasks
  const int c4j = w * 16 + (lane >> 2);                     // float4 column of the major task
  const bool maj_act = w < MAJ_TW && c4j * 4 < MAJ_C;
  const bool maj_in = maj_act && maj0 + c4j * 4 < majDim;
  uint32_t maj_voff = maj_in ? ((uint32_t)(r_beg + g * 8) * ldMaj4 + (uint32_t)(maj0 + c4j * 4) * 4u) : OOB;
  const int maj_off = c4j * 4 * TCP + ((g ^ ((c4j ^ (c4j >> 1)) & 1)) << 3);
  float4 scj = make_float4(1.f, 1.f, 1.f, 1.f), shj = make_float4(0.f, 0.f, 0.f, 0.f);
  if (AFFINE && A_MAJOR) {
    const int cc = min(maj0 + c4j * 4, Ka - 4);
    scj = ld4(a_scale + cc);
    shj = ld4(a_shift + cc);
  }

  f32x4s acc[MT > 0 ? MT : 1][NT], accr[RS > 0 ? RS : 1];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4s){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < RS; ++q) accr[q] = (f32x4s){0.f, 0.f, 0.f, 0.f};

  float4 rj[8], rn[8];
  int gi[GATHER ? 8 : 1];  // table rows of this thread's 8 rows of the NEXT tile to load (-1: zero row / past R / idle lane)
  const float* const Acol = A + min(maj0 + c4j * 4, Ka - 4);
  auto gidx = [&](int t) {
    if constexpr (GATHER) {
      const int r0 = r_beg + t * TKR + g * 8;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t v = a_rowidx[min(r0 + i, R - 1)];
        gi[i] = (maj_in && r0 + i < R && v >= 0) ? (int)v : -1;
      }
    }
  };
  auto gload = [&](int t) {
    if constexpr (GATHER) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 v = ld4(Acol + (int64_t)max(gi[i], 0) * lda);
        rj[i] = gi[i] >= 0 ? v : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) rj[i] = bload4(rsMaj, maj_voff + (uint32_t)i * ldMaj4);
      maj_voff += maj_in ? (uint32_t)TKR * ldMaj4 : 0u;
    }
    const int mk = (w - t) & 3;  // this tile's minor task-wave of this wave (wave-uniform)
    if (mk < MIN_TW) {
      const int c4 = mk * 16 + (lane >> 2);
      const bool in = c4 * 4 < MIN_C && min0 + c4 * 4 < minDim;
      const uint32_t voff = in ? ((uint32_t)(r_beg + t * TKR + g * 8) * ldMin4 + (uint32_t)(min0 + c4 * 4) * 4u) : OOB;
#pragma unroll
      for (int i = 0; i < 8; ++i) rn[i] = bload4(rsMin, voff + (uint32_t)i * ldMin4);
    }
  };
  auto lstore = [&](int t) {
    if (maj_act) store_task<AFFINE && A_MAJOR, NP>(Maj, MAJ_EL, maj_off, rj, scj, shj, A_MAJOR ? sa : sb);
    const int mk = (w - t) & 3;
    if (mk < MIN_TW) {
      const int c4 = mk * 16 + (lane >> 2);
      if (c4 * 4 < MIN_C) {
        const int off = c4 * 4 * TCP + ((g ^ ((c4 ^ (c4 >> 1)) & 1)) << 3);
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (AFFINE && !A_MAJOR) {
          const int cc = min(min0 + c4 * 4, Ka - 4);
          sc = ld4(a_scale + cc);
          sh = ld4(a_shift + cc);
        }
        store_task<AFFINE && !A_MAJOR, NP>(Min, MIN_EL, off, rn, sc, sh, A_MAJOR ? sb : sa);
      }
    }
  };

  const int rd_off = (lane & 15) * TCP + (((lane >> 4) ^ ((((lane & 15) >> 2) ^ ((lane & 15) >> 3)) & 1)) << 3);
#define QAGNN_SIX(C, AF, BF) C = mfma_pieces<NP>(AF, BF, C);

  gidx(0);
  gload(0);
  for (int t = 0; t < ntile; ++t) {
    __syncthreads();  // the previous tile's fragment reads are done
    lstore(t);
    gidx(t + 1);
    __syncthreads();
    gload(t + 1);  // in flight under the MFMAs; past the last tile: rows of the next chunk (or zeros), never stored
    __builtin_amdgcn_s_setprio(1);  // (as in k_gemm_nn_split: the MFMA phase goes first)
    u32x4s af[MT > 0 ? MT : 1][NP];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int p = 0; p < NP; ++p) af[i][p] = *reinterpret_cast<const u32x4s*>(As + p * A_EL + (w + 4 * i) * 16 * TCP + rd_off);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      u32x4s bf[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) bf[p] = *reinterpret_cast<const u32x4s*>(Bs + p * B_EL + j * 16 * TCP + rd_off);
#pragma unroll
      for (int i = 0; i < MT; ++i) {  // small terms first
        f32x4s c = acc[i][j];
        QAGNN_SIX(c, af[i], bf)
        acc[i][j] = c;
      }
    }
#pragma unroll
    for (int s = 0; s < RS; ++s) {
      const int q = w + 4 * s;  // wave-uniform
      if (q < REM * NT) {
        const int ir = q / NT, jr = q % NT;
        u32x4s ar[NP], br[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          ar[p] = *reinterpret_cast<const u32x4s*>(As + p * A_EL + (4 * MT + ir) * 16 * TCP + rd_off);
          br[p] = *reinterpret_cast<const u32x4s*>(Bs + p * B_EL + jr * 16 * TCP + rd_off);
        }
        f32x4s c = accr[s];
        QAGNN_SIX(c, ar, br)
        accr[s] = c;
      }
    }
    __builtin_amdgcn_s_setprio(0);
  }
#undef QAGNN_SIX

  float* const Pc = P + (int64_t)chunk * ka_total * No;
  const float inv = NP <= 2 ? h2_inv_scale(fa, fb) : 1.f;  // (NP == 2: the operand scales come out again, exactly)
  auto put = [&](int strip, int j, const f32x4s& c) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + strip * 16 + (lane >> 4) * 4 + r, col = n0 + j * 16 + (lane & 15);
      if (row < Ka && col < No) Pc[(int64_t)(row + row_shift) * No + col] = NP <= 2 ? c[r] * inv : c[r];
    }
  };
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) put(w + 4 * i, j, acc[i][j]);
#pragma unroll
  for (int s = 0; s < RS; ++s) {
    const int q = w + 4 * s;
    if (q < REM * NT) put(4 * MT + q / NT, q % NT, accr[s]);
  }
}

// ------------------------------------------------------------------------------------------------------------
// k_gemm_tn_ws (round 4): the same product with the work of a CU divided by KIND instead of by output.  Where k_gemm_tn_split runs at
// 44 % of the matrix pipe (187 us for the [X | S]^T dKMQ product of the 320-subgraph batch): per k-tile a block spends ~2 200 cycles of
// MFMAs, ~2 000 cycles of VALU work (split + transpose of 320 x 32 numbers) and ~1 900 cycles of LDS traffic, in barrier-separated
// phases, and the two blocks of a CU run in lockstep (same work, fair arbitration), so the three resources are used one after the other.
// Here ONE block of 8 waves owns the CU, one COMPUTE wave and one PRODUCER wave per SIMD:
//   * the four compute waves own the KT x NT output tiles exactly as the four waves of k_gemm_tn_split do (strips of the 13-wide
//     operand stay in registers; the (7, 13) shape mirrored, so that it reads 42 fragments per k-tile like the (13, 7) one instead of
//     100), and do nothing but fragment reads (issued two column tiles ahead, counted waits) and MFMAs;
//   * the four producer waves load the NEXT k-tile's rows of both operands (one unified space of float4 columns [A | B]: 80 columns =
//     5 task-waves of 64 tasks, four fixed and the fifth rotating over the producers), split and transpose them into the OTHER of two
//     LDS images (2 x 77 KB), and issue the loads of the tile after that;
//   * one barrier per k-tile.  The roles are taken by arrival order on each SIMD (HW_ID), so that every SIMD holds one wave of each
//     kind: the matrix pipe sees an MFMA stream, the vector ALU the split arithmetic, at the same time.
// ------------------------------------------------------------------------------------------------------------
static bool tn_xcd() { return true; }  // the blocks of one split-K chunk on one XCD (-2..5 % per kernel: profiles/r4_run17_tn_ws.txt)
constexpr int WTHR = 512;
// QAGNN_TNW_ABL (tools/tn_ablate.hip only; numerically wrong, timing only): bit 0 the producers do not split / store, bit 1 the producers
// do not load, bit 2 the compute waves issue no MFMAs, bit 3 no fragment reads either, bit 4 the producers wait for their loads and drop them
#ifndef QAGNN_TNW_ABL
#define QAGNN_TNW_ABL 0
#endif

// store_task with a per-lane floor (AFF: relu for the lanes of A, -inf = pass-through for the lanes of B)
template <bool AFF, int NP>
__device__ __forceinline__ void store_task_lo(uint16_t* __restrict__ img, int img_elems, int off, const float4 (&r)[8], float4 sc, float4 sh, float lo,
                                              float s) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v = j == 0 ? r[i].x : j == 1 ? r[i].y : j == 2 ? r[i].z : r[i].w;
      if (AFF) {
        const float s_ = j == 0 ? sc.x : j == 1 ? sc.y : j == 2 ? sc.z : sc.w, h = j == 0 ? sh.x : j == 1 ? sh.y : j == 2 ? sh.z : sh.w;
        v = fmaxf(fmaf(v, s_, h), lo);
      }
      x[i] = v;
    }
    store_col8<NP>(img + off + j * TCP, img_elems, x, s);
  }
}

template <int KT, int NT, bool AFFINE, int NP = 3>
__global__ __launch_bounds__(WTHR) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_gemm_tn_ws(
    const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ P, int R, int Ka, int No,
    const float* __restrict__ a_scale, const float* __restrict__ a_shift, int chunk_rows, const float* __restrict__ A2, int lda2, int Ka2,
    TnAmax amax) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_tw[];
  constexpr int AC = KT * 16, BC = NT * 16, A_EL = AC * TCP, B_EL = BC * TCP;
  constexpr int IMG_EL = NP * (A_EL + B_EL), IMG_B = IMG_EL * 2;  // one k-tile's image: [A: piece][column][32 rows + pad] | [B: ...]
  constexpr bool MAJ_A = KT >= NT;                                // the operand whose strips a compute wave keeps in registers
  constexpr int MAJT = MAJ_A ? KT : NT, MINT = MAJ_A ? NT : KT;
  constexpr int MT = MAJT / 4, REM = MAJT % 4, RS = (REM * MINT + 3) / 4;
  constexpr int MAJ_PIECE = (MAJ_A ? A_EL : B_EL) * 2, MIN_PIECE = (MAJ_A ? B_EL : A_EL) * 2;  // bytes between the pieces of an operand
  constexpr int STRIP_B = 16 * TCP * 2;                                                        // bytes of one 16-column strip
  constexpr int NC4 = (AC + BC) / 4, NTW = (NC4 + 15) / 16;                                    // float4 columns of [A | B], task-waves
  static_assert(MT >= 1 && NTW >= 4 && NTW <= 8, "shapes: (13, 7) and (7, 13)");
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int ka_total = Ka + (A2 ? Ka2 : 0);
  // XCD-aware block order (chunk_rows < 0: off): the blocks of one chunk read the same rows of A and B, so they belong on ONE XCD,
  // next to each other in time -- in launch order they are dealt out over all eight L2s and every operand row crosses the fabric once
  // per block that uses it (744 MB instead of 242 MB for the two-operand product: the kernel ran at the speed of those re-reads)
  int bx = blockIdx.x, by = blockIdx.y, chunk = blockIdx.z, row_shift = 0;
  if (chunk_rows < 0) {
    chunk_rows = -chunk_rows;
  } else {
    const int lin = bx + (int)gridDim.x * (by + (int)gridDim.y * chunk);
    const int v = xcd_remap(lin, (int)(gridDim.x * gridDim.y * gridDim.z));
    bx = v % (int)gridDim.x;
    by = (v / (int)gridDim.x) % (int)gridDim.y;
    chunk = v / (int)(gridDim.x * gridDim.y);
  }
  bool second = false;
  if (A2 != nullptr) {
    const int n1 = (Ka + AC - 1) / AC;
    if (by >= n1) { by -= n1; row_shift = Ka; A = A2; lda = lda2; Ka = Ka2; second = true; }
  }
  uint32_t fa = 127u, fb = 127u;
  float sa = 1.f, sb = 1.f;
  if constexpr (NP <= 2) {
    fa = __builtin_amdgcn_readfirstlane(h2_scale_field(second ? amax.a2[0] : amax.a[0]));
    fb = __builtin_amdgcn_readfirstlane(h2_scale_field(amax.b[0]));
    sa = h2_field_to_scale(fa);
    sb = h2_field_to_scale(fb);
  }
  const int n0 = bx * BC, m0 = by * AC;
  const int r_beg = chunk * chunk_rows, r_end = min(R, r_beg + chunk_rows);
  const int ntile = (r_end - r_beg + TKR - 1) / TKR;  // chunk_rows is a multiple of TKR: only the last chunk has a ragged tile, past R

  // roles: the first wave to arrive on a SIMD computes, the second produces; should the hardware ever place the waves otherwise
  // (not 4 + 4), waves 0-3 compute and 4-7 produce -- correct either way, only the overlap is lost
  int* const ctl = reinterpret_cast<int*>(smem_tw + 2 * IMG_B);
  if (tid < 8) ctl[tid] = 0;
  __syncthreads();
  int role, idx;
  {
    const int simd = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3;  // HW_ID[5:4] = SIMD_ID
    int v = 0;
    if (lane == 0) v = atomicAdd(&ctl[simd], 1);
    role = __builtin_amdgcn_readfirstlane(v) & 1;
    if (lane == 0) v = atomicAdd(&ctl[4 + role], 1);
    idx = __builtin_amdgcn_readfirstlane(v);
    __syncthreads();
    if (ctl[4] != 4 || ctl[5] != 4) { role = w >> 2; idx = w & 3; }
  }
#define QAGNN_TNW_BAR                                                  \
  {                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                 \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    \
    __builtin_amdgcn_sched_barrier(0);                                 \
  }
  __syncthreads();  // (ctl is read; the images may be written)

  const int ntile4 = (ntile + 3) & ~3;  // both roles run whole groups of four k-tiles (the producers' loop is unrolled by four)
  if (role == 1) {
    // ---------------------------------------------------------------- producer
    // Everything about a producer's k-tile is a compile-time constant of (its index, the tile's position in a group of four): which
    // task-waves it holds (its own; the rotating fifth on tile T iff (idx - T) % 4 == 0), which operand(s) their columns lie in, how
    // many loads that makes.  hipcc then counts its vmcnt waits exactly; with the same facts behind uniform branches it waited for
    // vmcnt(0) in front of every split -- i.e. for the loads it had just issued for the tile after (3.1 us per k-tile instead of 1.2).
    // Loads and stores run past the chunk's last tile unconditionally (rows past R read zeros, rows of the next chunk land in an
    // image nobody reads).
    auto produce = [&](auto IDXC) {
      constexpr int IDX = decltype(IDXC)::value;
      const int g = lane & 3;
      const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, R * lda * 4, 0x00020000);
      const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, R * ldb * 4, 0x00020000);
      constexpr uint32_t OOB = 0x80000000u;  // (+ any offset of this kernel stays out of range: the load answers with zeros)
      const uint32_t ldA4 = (uint32_t)lda * 4u, ldB4 = (uint32_t)ldb * 4u;
      constexpr int TW[2] = {IDX, 4};
      constexpr bool HAS_A[2] = {TW[0] * 16 < AC / 4, TW[1] * 16 < AC / 4};
      constexpr bool HAS_B[2] = {TW[0] * 16 + 15 >= AC / 4, TW[1] * 16 + 15 >= AC / 4};
      uint32_t voA[2], voB[2];
      int off[2], iel[2];
      float4 sc[2], sh[2];
      float lo[2], ssc[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int c4u = TW[k] * 16 + (lane >> 2);
        const bool act = c4u < NC4 && (k == 0 || NTW > 4);
        const bool isA = c4u < AC / 4;
        const int c4 = isA ? c4u : c4u - AC / 4;
        voA[k] = act && isA && m0 + c4 * 4 < Ka ? (uint32_t)(r_beg + g * 8) * ldA4 + (uint32_t)(m0 + c4 * 4) * 4u : OOB;
        voB[k] = act && !isA && n0 + c4 * 4 < No ? (uint32_t)(r_beg + g * 8) * ldB4 + (uint32_t)(n0 + c4 * 4) * 4u : OOB;
        iel[k] = isA ? A_EL : B_EL;
        off[k] = act ? (isA ? 0 : NP * A_EL) + c4 * 4 * TCP + ((g ^ ((c4 ^ (c4 >> 1)) & 1)) << 3) : -1;
        ssc[k] = isA ? sa : sb;
        sc[k] = make_float4(1.f, 1.f, 1.f, 1.f);
        sh[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        lo[k] = -INFINITY;
        if (AFFINE && isA && act) {
          const int cc = min(m0 + c4 * 4, Ka - 4);
          sc[k] = ld4(a_scale + cc);
          sh[k] = ld4(a_shift + cc);
          lo[k] = 0.f;
        }
      }
      // two register sets per task slot: the loads of tile t + 2 are issued BEFORE the split work of tile t + 1 and have a whole
      // k-tile to land (even tiles in set a, odd tiles in set b)
      float4 r1a[8], r2a[8], r1b[8], r2b[8];
      if constexpr ((QAGNN_TNW_ABL & 2) != 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) r1a[i] = r2a[i] = r1b[i] = r2b[i] = make_float4(1.f + lane, 2.f, 3.f, 4.f);
      }
      // the rows of tile T of task slot K -> the registers RR (a task-wave that straddles the end of A asks both operands and ORs)
#define QAGNN_TNW_LOAD1(T, K, RR)                                                                          \
      {                                                                                                    \
        const uint32_t a_ = voA[K] + (uint32_t)(T) * (TKR * ldA4), b_ = voB[K] + (uint32_t)(T) * (TKR * ldB4); \
        if constexpr (HAS_A[K] && HAS_B[K]) {                                                              \
          _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                  \
            const u32x4s x_ = __builtin_bit_cast(u32x4s, bload4(rsA, a_ + (uint32_t)i * ldA4)) |           \
                              __builtin_bit_cast(u32x4s, bload4(rsB, b_ + (uint32_t)i * ldB4));            \
            RR[i] = __builtin_bit_cast(float4, x_);                                                        \
          }                                                                                                \
        } else if constexpr (HAS_A[K]) {                                                                   \
          _Pragma("unroll") for (int i = 0; i < 8; ++i) RR[i] = bload4(rsA, a_ + (uint32_t)i * ldA4);      \
        } else {                                                                                           \
          _Pragma("unroll") for (int i = 0; i < 8; ++i) RR[i] = bload4(rsB, b_ + (uint32_t)i * ldB4);      \
        }                                                                                                  \
      }
      // TP: the tile's position modulo 4 (t itself is a multiple of 4)
#define QAGNN_TNW_LOAD(T, TP, R1, R2)                                                                      \
      if constexpr (!(QAGNN_TNW_ABL & 2)) {                                                                \
        QAGNN_TNW_LOAD1(T, 0, R1)                                                                          \
        if constexpr (NTW > 4 && ((IDX - (TP)) & 3) == 0) QAGNN_TNW_LOAD1(T, 1, R2)                        \
      }
#define QAGNN_TNW_STORE(T, TP, R1, R2)                                                                     \
      if constexpr ((QAGNN_TNW_ABL & 16) != 0) { /* wait for the tile's loads, do nothing with them */     \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(R1[i].x), "v"(R1[i].w));       \
        if constexpr (NTW > 4 && ((IDX - (TP)) & 3) == 0) {                                                \
          _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile("" ::"v"(R2[i].x), "v"(R2[i].w));     \
        }                                                                                                  \
      } else if constexpr (!(QAGNN_TNW_ABL & 1)) {                                                         \
        uint16_t* const img_ = reinterpret_cast<uint16_t*>(smem_tw) + ((TP)&1) * IMG_EL;                   \
        if (off[0] >= 0) store_task_lo<AFFINE, NP>(img_, iel[0], off[0], R1, sc[0], sh[0], lo[0], ssc[0]); \
        if constexpr (NTW > 4 && ((IDX - (TP)) & 3) == 0) {                                                \
          if (off[1] >= 0) store_task_lo<AFFINE, NP>(img_, iel[1], off[1], R2, sc[1], sh[1], lo[1], ssc[1]); \
        }                                                                                                  \
      }
      QAGNN_TNW_LOAD(0, 0, r1a, r2a)
      QAGNN_TNW_LOAD(1, 1, r1b, r2b)
      QAGNN_TNW_STORE(0, 0, r1a, r2a)
      QAGNN_TNW_BAR
      for (int t = 0; t < ntile4; t += 4) {
        QAGNN_TNW_LOAD(t + 2, 2, r1a, r2a)   // compute: tile t
        QAGNN_TNW_STORE(t + 1, 1, r1b, r2b)  // (the compute waves read the other image)
        QAGNN_TNW_BAR
        QAGNN_TNW_LOAD(t + 3, 3, r1b, r2b)   // compute: tile t + 1
        QAGNN_TNW_STORE(t + 2, 2, r1a, r2a)
        QAGNN_TNW_BAR
        QAGNN_TNW_LOAD(t + 4, 0, r1a, r2a)   // compute: tile t + 2
        QAGNN_TNW_STORE(t + 3, 3, r1b, r2b)
        QAGNN_TNW_BAR
        QAGNN_TNW_LOAD(t + 5, 1, r1b, r2b)   // compute: tile t + 3
        QAGNN_TNW_STORE(t + 4, 0, r1a, r2a)
        QAGNN_TNW_BAR
      }
#undef QAGNN_TNW_LOAD
#undef QAGNN_TNW_LOAD1
#undef QAGNN_TNW_STORE
    };
    switch (idx) {
      case 0: produce(std::integral_constant<int, 0>{}); break;
      case 1: produce(std::integral_constant<int, 1>{}); break;
      case 2: produce(std::integral_constant<int, 2>{}); break;
      default: produce(std::integral_constant<int, 3>{}); break;
    }
    return;
  }

  // ------------------------------------------------------------------ compute
  const int ci = idx;
  f32x4s acc[MT][MINT], accr[RS > 0 ? RS : 1];
#pragma unroll
  for (int s_ = 0; s_ < MT; ++s_)
#pragma unroll
    for (int n = 0; n < MINT; ++n) acc[s_][n] = (f32x4s){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < RS; ++q) accr[q] = (f32x4s){0.f, 0.f, 0.f, 0.f};

  const int rd_off = (lane & 15) * TCP + (((lane >> 4) ^ ((((lane & 15) >> 2) ^ ((lane & 15) >> 3)) & 1)) << 3);
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem_tw + (uint32_t)rd_off * 2u;
  const uint32_t maj0 = lds0 + (MAJ_A ? 0u : (uint32_t)(NP * A_EL * 2)) + (uint32_t)(ci * STRIP_B);  // this wave's first strip
  const uint32_t min0 = lds0 + (MAJ_A ? (uint32_t)(NP * A_EL * 2) : 0u);
  // the leftover strips' tiles, dealt out one by one: tile q = ci + 4 s -> (strip 4 MT + q / MINT, minor tile q % MINT)
  uint32_t rem_maj[RS > 0 ? RS : 1], rem_min[RS > 0 ? RS : 1];
#pragma unroll
  for (int s_ = 0; s_ < RS; ++s_) {
    const int q = ci + 4 * s_;
    rem_maj[s_] = lds0 + (MAJ_A ? 0u : (uint32_t)(NP * A_EL * 2)) + (uint32_t)((4 * MT + q / MINT) * STRIP_B);
    rem_min[s_] = min0 + (uint32_t)((q % MINT) * STRIP_B);
  }
#define QAGNN_TNW_FRAG(DST, ADDR, OFF, PIECE)                                                                            \
  {                                                                                                                      \
    _Pragma("unroll") for (int p = 0; p < NP; ++p)                                                                       \
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST[p]) : "v"(ADDR), "i"((OFF) + p * (PIECE)));              \
  }
  // N: the number of FRAGMENT SETS (NP reads each) that may stay in flight behind F
#define QAGNN_TNW_WAIT(F, N)                                                                                             \
  {                                                                                                                      \
    if constexpr (NP == 3) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]) : "i"((N) * 3));    \
    else if constexpr (NP == 2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(F[0]), "+v"(F[1]) : "i"((N) * 2));           \
    else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(F[0]) : "i"((N) * 1));                                              \
  }
#define QAGNN_TNW_TOUCH(F)                                                               \
  {                                                                                      \
    if constexpr (NP == 3) asm volatile("" : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]));        \
    else if constexpr (NP == 2) asm volatile("" : "+v"(F[0]), "+v"(F[1]));               \
    else asm volatile("" : "+v"(F[0]));                                                  \
  }
#define QAGNN_TNW_SIX(C, AF, BF)                                          \
  if constexpr ((QAGNN_TNW_ABL & 4) != 0) { asm volatile("" ::"v"(AF[0]), "v"(AF[NP > 1 ? 1 : 0]), "v"(BF[0]), "v"(BF[NP > 1 ? 1 : 0])); } else { C = mfma_pieces<NP>(AF, BF, C); }

  QAGNN_TNW_BAR  // image 0 is complete
  for (int t = 0; t < ntile4; ++t) {
    if (t >= ntile) {  // (the producers' group of four is not over)
      QAGNN_TNW_BAR
      continue;
    }
    if constexpr ((QAGNN_TNW_ABL & 8) != 0) {
      QAGNN_TNW_BAR
      continue;
    }
    const uint32_t cur = (uint32_t)((t & 1) * IMG_B);
    const uint32_t amaj = maj0 + cur, amin = min0 + cur;
    u32x4s mf[MT][NP], nf[3][NP];
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s_ = 0; s_ < MT; ++s_) QAGNN_TNW_FRAG(mf[s_], amaj, s_ * 4 * STRIP_B, MAJ_PIECE)
    QAGNN_TNW_FRAG(nf[0], amin, 0, MIN_PIECE)
    if constexpr (MINT > 1) QAGNN_TNW_FRAG(nf[1], amin, STRIP_B, MIN_PIECE)
#pragma unroll
    for (int n = 0; n < MINT; ++n) {
      u32x4s(&bfn)[NP] = nf[n % 3];
      if (n + 2 < MINT) {
        QAGNN_TNW_FRAG(nf[(n + 2) % 3], amin, (n + 2) * STRIP_B, MIN_PIECE)
        QAGNN_TNW_WAIT(bfn, 2);
      } else if (n + 1 < MINT) {
        QAGNN_TNW_WAIT(bfn, 1);
      } else {
        QAGNN_TNW_WAIT(bfn, 0);
      }
      if (n == 0) {  // (the strips' fragments are older than every minor fragment: landed with the first wait)
#pragma unroll
        for (int s_ = 0; s_ < MT; ++s_) QAGNN_TNW_TOUCH(mf[s_])
      }
#pragma unroll
      for (int s_ = 0; s_ < MT; ++s_) {  // small terms first
        f32x4s c = acc[s_][n];
        if constexpr (MAJ_A) { QAGNN_TNW_SIX(c, mf[s_], bfn) } else { QAGNN_TNW_SIX(c, bfn, mf[s_]) }
        acc[s_][n] = c;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int s_ = 0; s_ < RS; ++s_) {
      if (ci + 4 * s_ < REM * MINT) {  // wave-uniform
        u32x4s ar[NP], br[NP];
        QAGNN_TNW_FRAG(ar, rem_maj[s_] + cur, 0, MAJ_PIECE)
        QAGNN_TNW_FRAG(br, rem_min[s_] + cur, 0, MIN_PIECE)
        QAGNN_TNW_WAIT(ar, 0);
        QAGNN_TNW_TOUCH(br)
        f32x4s c = accr[s_];
        if constexpr (MAJ_A) { QAGNN_TNW_SIX(c, ar, br) } else { QAGNN_TNW_SIX(c, br, ar) }
        accr[s_] = c;
      }
    }
    __builtin_amdgcn_s_setprio(0);
    QAGNN_TNW_BAR
  }
#undef QAGNN_TNW_FRAG
#undef QAGNN_TNW_WAIT
#undef QAGNN_TNW_TOUCH
#undef QAGNN_TNW_SIX
#undef QAGNN_TNW_BAR

  float* const Pc = P + (int64_t)chunk * ka_total * No;
  const float inv = NP <= 2 ? h2_inv_scale(fa, fb) : 1.f;
  auto put = [&](int maj_strip, int min_tile, const f32x4s& c) {
    const int rstrip = MAJ_A ? maj_strip : min_tile, ctile = MAJ_A ? min_tile : maj_strip;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m0 + rstrip * 16 + (lane >> 4) * 4 + r, col = n0 + ctile * 16 + (lane & 15);
      if (row < Ka && col < No) Pc[(int64_t)(row + row_shift) * No + col] = NP <= 2 ? c[r] * inv : c[r];
    }
  };
#pragma unroll
  for (int s_ = 0; s_ < MT; ++s_)
#pragma unroll
    for (int n = 0; n < MINT; ++n) put(ci + 4 * s_, n, acc[s_][n]);
#pragma unroll
  for (int s_ = 0; s_ < RS; ++s_) {
    const int q = ci + 4 * s_;
    if (q < REM * MINT) put(4 * MT + q / MINT, q % MINT, accr[s_]);
  }
}

template <int KT, int NT, bool AFFINE, int NP = 3>
static int launch_tn_ws_i(dim3 grid, hipStream_t stream, const float* A, int lda, const float* B, int ldb, float* P, int R, int Ka, int No,
                          const float* sc, const float* sh, int chunk_rows, const float* A2 = nullptr, int lda2 = 0, int Ka2 = 0,
                          TnAmax amax = TnAmax{nullptr, nullptr, nullptr}) {
  constexpr size_t lds = (size_t)2 * NP * (KT * 16 + NT * 16) * TCP * sizeof(uint16_t) + 64;
  static bool raised[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (!raised[dev & 63]) {
    hipError_t e = hipFuncSetAttribute((const void*)k_gemm_tn_ws<KT, NT, AFFINE, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("gemm_tn_ws: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e)); return QAGNN_EHIP; }
    raised[dev & 63] = true;
  }
  k_gemm_tn_ws<KT, NT, AFFINE, NP><<<grid, WTHR, lds, stream>>>(A, lda, B, ldb, P, R, Ka, No, sc, sh, tn_xcd() ? chunk_rows : -chunk_rows, A2, lda2, Ka2, amax);
  QAGNN_LAUNCH_CHECK("k_gemm_tn_ws");
  return QAGNN_OK;
}

template <int KT, int NT, bool AFFINE, bool GATHER = false, int NP = 3>
static int launch_tn_split_i(dim3 grid, hipStream_t stream, const float* A, int lda, const float* B, int ldb, float* P, int R, int Ka, int No,
                             const float* sc, const float* sh, int chunk_rows, const int64_t* ridx = nullptr, const float* A2 = nullptr,
                             int lda2 = 0, int Ka2 = 0, TnAmax amax = TnAmax{nullptr, nullptr, nullptr}) {
  constexpr size_t lds = (size_t)NP * (KT * 16 + NT * 16) * TCP * sizeof(uint16_t);
  static bool raised[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (lds > 64 * 1024 && !raised[dev & 63]) {
    hipError_t e = hipFuncSetAttribute((const void*)k_gemm_tn_split<KT, NT, AFFINE, GATHER, NP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("gemm_tn_split: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e)); return QAGNN_EHIP; }
    raised[dev & 63] = true;
  }
  k_gemm_tn_split<KT, NT, AFFINE, GATHER, NP><<<grid, TTHR, lds, stream>>>(A, lda, B, ldb, P, R, Ka, No, sc, sh, tn_xcd() ? chunk_rows : -chunk_rows, ridx,
                                                                           A2, lda2, Ka2, amax);
  QAGNN_LAUNCH_CHECK("k_gemm_tn_split");
  return QAGNN_OK;
}

// QAGNN_GEMM_SPLIT=0 pins the fp32-MFMA kernels of gemm.hip: the one numerically distinct fallback (the module mirror reads the same
// variable for the NN products: qagnn_amd/_lib.py)
static int tn_split_mode() {
  static const int v = getenv("QAGNN_GEMM_SPLIT") ? atoi(getenv("QAGNN_GEMM_SPLIT")) : 1;
  return v;
}
bool tn_split_ok(int R, int Ka, int No, int lda, int ldb, bool gather, bool affine) {
  const int v = tn_split_mode();
  const int64_t big = (int64_t)R * ((lda > ldb && !gather) ? lda : ldb) * 4;  // (a gathered A goes through flat loads)
  if (gather && (Ka <= 112 || affine)) return false;
  return v != 0 && Ka >= 64 && No >= 104 && R >= 1024 && big < (int64_t)0x7FFFFFFF;  // 32-bit buffer offsets
}
static bool tn_split_wide_b(int Ka) { return Ka <= 112; }  // (KT, NT) = (7, 13), else (13, 7)
constexpr int TN_WS_MIN_TILES_C = 28;  // (= TN_WS_MIN_TILES below: chunks of that many k-tiles and more run k_gemm_tn_ws)
// rows per split-K chunk: two blocks per CU, a multiple of the 32-row k-tile, never below `lo` (the workspace's sizing)
int tn_split_chunk_rows(int R, int Ka, int No, int lo) {
  const int ac = tn_split_wide_b(Ka) ? 112 : 208, bc = tn_split_wide_b(Ka) ? 208 : 112;
  const int blocks_per_chunk = cdiv(No, bc) * cdiv(Ka, ac);
  const int target = 2 * split_num_cus() / blocks_per_chunk;
  const int rows = (cdiv(R, target > 0 ? target : 1) + TKR - 1) / TKR * TKR;
  const int lo32 = (lo + TKR - 1) / TKR * TKR;
  return rows > lo32 ? rows : lo32;
}
// [A1 | A2]^T B in one launch: 112-row output tiles (the (7, 13) shape) over A1's columns, then over A2's
int tn_split2_chunk_rows(int R, int Ka1, int Ka2, int No, int lo) {
  const int blocks_per_chunk = cdiv(No, 208) * (cdiv(Ka1, 112) + cdiv(Ka2, 112));
  const int lo32 = (lo + TKR - 1) / TKR * TKR;
#ifndef QAGNN_TN_WS_TWO_ROUNDS
  // Long products run k_gemm_tn_ws, ONE 8-wave block per CU: chunks sized for one block per CU (28 chunks of 72 k-tiles at 64 000 rows
  // instead of 56 of 36) halve the partial sums that are written and summed again and the uncovered first loads / last stores per CU
  {
    const int target1 = split_num_cus() / blocks_per_chunk;
    const int rows1 = (cdiv(R, target1 > 0 ? target1 : 1) + TKR - 1) / TKR * TKR;
    if (rows1 >= 2 * TN_WS_MIN_TILES_C * TKR) return rows1 > lo32 ? rows1 : lo32;
  }
#endif
  const int target = 2 * split_num_cus() / blocks_per_chunk;
  const int rows = (cdiv(R, target > 0 ? target : 1) + TKR - 1) / TKR * TKR;
  return rows > lo32 ? rows : lo32;
}
// k_gemm_tn_ws serves the two-operand product [X | S]^T dKMQ where its chunks are long (36 k-tiles per block at 64 000 rows: 201 -> 171 us).
// Measured with 8 - 24 k-tiles per block (tools/tn_ablate.hip, profiles/r4_run17_tn_ws.txt): 208 x 624 115 -> 126 us, 208 x 208 43 -> 57,
// 112 x 624 72 -> 74 -- one block per CU leaves a block's first loads and its partial-sum stores uncovered, which only a long chunk
// amortises; at the 2-tile chunks of a 10-subgraph batch it took 29 us against the 4-wave kernel's ~14 (profiles/r5_run5_tn_ws_min_tiles_ab_b10.txt:
// the step 2.29 -> 2.22 ms)
constexpr int TN_WS_MIN_TILES = TN_WS_MIN_TILES_C;
int launch_tn_split2(const float* A1, int lda1, int Ka1, const float* A2, int lda2, int Ka2, const float* B, int ldb, float* P, int R, int No,
                     int chunk_rows, hipStream_t stream, const uint32_t* const* amax, int np) {
  dim3 grid(cdiv(No, 208), cdiv(Ka1, 112) + cdiv(Ka2, 112), cdiv(R, chunk_rows));
  if (amax) {  // the scaled fp16 forms (amax = {max|A1|, max|A2|, max|B|}; np = 2: three MFMAs, np = 1: one)
    const TnAmax am{amax[0], amax[1], amax[2]};
    if (np == 1) {
      if (chunk_rows >= TN_WS_MIN_TILES * 32)
        return launch_tn_ws_i<7, 13, false, 1>(grid, stream, A1, lda1, B, ldb, P, R, Ka1, No, nullptr, nullptr, chunk_rows, A2, lda2, Ka2, am);
      return launch_tn_split_i<7, 13, false, false, 1>(grid, stream, A1, lda1, B, ldb, P, R, Ka1, No, nullptr, nullptr, chunk_rows, nullptr, A2, lda2, Ka2, am);
    }
    if (chunk_rows >= TN_WS_MIN_TILES * 32)
      return launch_tn_ws_i<7, 13, false, 2>(grid, stream, A1, lda1, B, ldb, P, R, Ka1, No, nullptr, nullptr, chunk_rows, A2, lda2, Ka2, am);
    return launch_tn_split_i<7, 13, false, false, 2>(grid, stream, A1, lda1, B, ldb, P, R, Ka1, No, nullptr, nullptr, chunk_rows, nullptr, A2, lda2, Ka2, am);
  }
  if (chunk_rows >= TN_WS_MIN_TILES * 32)
    return launch_tn_ws_i<7, 13, false>(grid, stream, A1, lda1, B, ldb, P, R, Ka1, No, nullptr, nullptr, chunk_rows, A2, lda2, Ka2);
  return launch_tn_split_i<7, 13, false>(grid, stream, A1, lda1, B, ldb, P, R, Ka1, No, nullptr, nullptr, chunk_rows, nullptr, A2, lda2, Ka2);
}
int launch_tn_split(const float* A, int lda, const float* B, int ldb, float* P, int R, int Ka, int No, const float* sc, const float* sh,
                    const int64_t* ridx, int chunk_rows, hipStream_t stream, const uint32_t* const* amax, int np) {
  if (amax && !ridx && np == 1) {  // the one-MFMA reduced-precision form
    const TnAmax am{amax[0], nullptr, amax[2]};
    if (tn_split_wide_b(Ka)) {
      dim3 grid(cdiv(No, 208), cdiv(Ka, 112), cdiv(R, chunk_rows));
      return sc ? launch_tn_split_i<7, 13, true, false, 1>(grid, stream, A, lda, B, ldb, P, R, Ka, No, sc, sh, chunk_rows, nullptr, nullptr, 0, 0, am)
                : launch_tn_split_i<7, 13, false, false, 1>(grid, stream, A, lda, B, ldb, P, R, Ka, No, sc, sh, chunk_rows, nullptr, nullptr, 0, 0, am);
    }
    dim3 grid(cdiv(No, 112), cdiv(Ka, 208), cdiv(R, chunk_rows));
    return sc ? launch_tn_split_i<13, 7, true, false, 1>(grid, stream, A, lda, B, ldb, P, R, Ka, No, sc, sh, chunk_rows, nullptr, nullptr, 0, 0, am)
              : launch_tn_split_i<13, 7, false, false, 1>(grid, stream, A, lda, B, ldb, P, R, Ka, No, sc, sh, chunk_rows, nullptr, nullptr, 0, 0, am);
  }
  if (amax && !ridx) {  // the three-MFMA form (amax = {max|A|, -, max|B|})
    const TnAmax am{amax[0], nullptr, amax[2]};
    if (tn_split_wide_b(Ka)) {
      dim3 grid(cdiv(No, 208), cdiv(Ka, 112), cdiv(R, chunk_rows));
      return sc ? launch_tn_split_i<7, 13, true, false, 2>(grid, stream, A, lda, B, ldb, P, R, Ka, No, sc, sh, chunk_rows, nullptr, nullptr, 0, 0, am)
                : launch_tn_split_i<7, 13, false, false, 2>(grid, stream, A, lda, B, ldb, P, R, Ka, No, sc, sh, chunk_rows, nullptr, nullptr, 0, 0, am);
    }
    dim3 grid(cdiv(No, 112), cdiv(Ka, 208), cdiv(R, chunk_rows));
    return sc ? launch_tn_split_i<13, 7, true, false, 2>(grid, stream, A, lda, B, ldb, P, R, Ka, No, sc, sh, chunk_rows, nullptr, nullptr, 0, 0, am)
              : launch_tn_split_i<13, 7, false, false, 2>(grid, stream, A, lda, B, ldb, P, R, Ka, No, sc, sh, chunk_rows, nullptr, nullptr, 0, 0, am);
  }
  if (ridx) {
    dim3 grid(cdiv(No, 112), cdiv(Ka, 208), cdiv(R, chunk_rows));
    return launch_tn_split_i<13, 7, false, true>(grid, stream, A, lda, B, ldb, P, R, Ka, No, sc, sh, chunk_rows, ridx);
  }
  if (tn_split_wide_b(Ka)) {
    dim3 grid(cdiv(No, 208), cdiv(Ka, 112), cdiv(R, chunk_rows));
    return sc ? launch_tn_split_i<7, 13, true>(grid, stream, A, lda, B, ldb, P, R, Ka, No, sc, sh, chunk_rows)
              : launch_tn_split_i<7, 13, false>(grid, stream, A, lda, B, ldb, P, R, Ka, No, sc, sh, chunk_rows);
  }
  dim3 grid(cdiv(No, 112), cdiv(Ka, 208), cdiv(R, chunk_rows));
  return sc ? launch_tn_split_i<13, 7, true>(grid, stream, A, lda, B, ldb, P, R, Ka, No, sc, sh, chunk_rows)
            : launch_tn_split_i<13, 7, false>(grid, stream, A, lda, B, ldb, P, R, Ka, No, sc, sh, chunk_rows);
}

}  // namespace qagnn

using namespace qagnn;

extern "C" int64_t qagnn_gemm_nn_pack_bytes(int32_t No, int32_t K1, int32_t K2) { return nn2_pack_bytes(No, K1, K2, 3) + 256; }  // (>= the two-piece image + its scale words)

extern "C" int64_t qagnn_gemm_nn_ws_bytes(const qagnn_gemm_nn_args* a, const float* B1n, int32_t ldn1, const float* B2n, int32_t ldn2) {
  if (!a || !B1n || !nn2_ok(*a, ldn1, ldn2)) return 0;                                      // not a product of the packed kernels
  const int np = nn2_h2_ok(*a) ? (a->pieces == 1 ? 1 : 2) : 3;
  if (nn2_prepack_lookup(B1n, ldn1, a->K1, B2n, ldn2, a->K2, a->No, np)) return 0;  // B is registered: nothing to pack per call
  const int64_t need = nn2_pack_bytes(a->No, a->K1, a->K2, np);
  return nn2_packed_ok(*a, need, np) ? need : 0;                                              // (too few rows: the in-kernel split)
}

extern "C" int qagnn_gemm_nn_split_f32(const qagnn_gemm_nn_args* a, const float* B1n, int32_t ldn1, const float* B2n, int32_t ldn2,
                                       qagnn_stream_t stream_) {
  return qagnn_gemm_nn_split_ws_f32(a, B1n, ldn1, B2n, ldn2, nullptr, 0, stream_);
}

extern "C" int qagnn_gemm_nn_split_ws_f32(const qagnn_gemm_nn_args* a, const float* B1n, int32_t ldn1, const float* B2n, int32_t ldn2,
                                          void* ws, int64_t ws_bytes, qagnn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  TimedScope timed(0, stream);
  QAGNN_REQUIRE(a && a->A1 && B1n && a->C, QAGNN_EINVAL, "gemm_nn_split: null pointer");
  QAGNN_REQUIRE(a->M > 0 && a->No > 0 && a->K1 >= 4, QAGNN_EINVAL, "gemm_nn_split: bad sizes M=%d No=%d K1=%d", a->M, a->No, a->K1);
  QAGNN_REQUIRE(a->K1 % 4 == 0 && a->K2 % 4 == 0 && a->K2 >= 0 && (a->K2 == 0 || a->K2 >= 4), QAGNN_EINVAL,
                "gemm_nn_split: K1=%d K2=%d must be multiples of 4", a->K1, a->K2);
  QAGNN_REQUIRE(a->No % 4 == 0, QAGNN_EINVAL, "gemm_nn_split: No=%d must be a multiple of 4", a->No);
  QAGNN_REQUIRE(a->lda1 % 4 == 0 && ldn1 % 4 == 0 && ldn1 >= a->K1 && aligned16(a->A1) && aligned16(B1n), QAGNN_EINVAL,
                "gemm_nn_split: operand 1 must be 16-byte aligned with pitches multiple of 4");
  QAGNN_REQUIRE(a->K2 == 0 || (a->A2 && B2n && a->lda2 % 4 == 0 && ldn2 % 4 == 0 && ldn2 >= a->K2 && aligned16(a->A2) && aligned16(B2n)),
                QAGNN_EINVAL, "gemm_nn_split: operand 2 must be 16-byte aligned with pitches multiple of 4");
  QAGNN_REQUIRE(!a->rowtab || a->rowidx, QAGNN_EINVAL, "gemm_nn_split: rowtab without rowidx");
  QAGNN_REQUIRE(a->ldc % 4 == 0 && aligned16(a->C) && (!a->bias || aligned16(a->bias)) && (!a->rowtab || (aligned16(a->rowtab) && a->ldt % 4 == 0)),
                QAGNN_EINVAL, "gemm_nn_split: C / bias / rowtab must be 16-byte aligned with pitches multiple of 4");
  QAGNN_REQUIRE(!a->a_scale || (a->a_shift && aligned16(a->a_scale) && aligned16(a->a_shift)), QAGNN_EINVAL,
                "gemm_nn_split: a_scale/a_shift must both be given and 16-byte aligned");
  const int nt16 = cdiv(a->No, 16);
  if (a->colstat_part) {
    const int64_t lim = (int64_t)0x7FFFFFFF;
    QAGNN_REQUIRE(nt16 == 13 && !a->a_scale && !a->a_rowidx && !a->rowtab && !a->accumulate && a->K2 == 0, QAGNN_EUNSUPPORTED,
                  "gemm_nn_split: column statistics need 193..208 output columns and a bias-only epilogue (No=%d)", a->No);
    QAGNN_REQUIRE((int64_t)a->M * a->lda1 * 4 < lim && (int64_t)a->No * ldn1 * 4 < lim, QAGNN_EUNSUPPORTED,
                  "gemm_nn_split: column statistics with operands of 2 GB and more");
  }
  int nt = nt16 >= 13 ? 13 : nt16 >= 8 ? 8 : nt16 >= 7 ? 7 : nt16 >= 4 ? 4 : 2;
  // Few row tiles (the host-bound configurations: 10 subgraphs are 16 row tiles, a 64-subgraph MedQA shard 100, on 256 CUs): a
  // block's time is its own serial k-loop, which scales with the column tiles it carries, and the other CUs idle -- so the column
  // tile narrows until there are about 1.5 blocks per CU (or it is 32 columns wide).  Measured, rocprofv3 kernel durations
  // (profiles/r3_run5_nn_small_m.txt): 2 000 x 208 x 208 25.3 -> 9.4 us at 32 columns; 12 800 rows 27.3 -> 16.3 us at 64 columns
  // (18.6 at 32); 624 -> 208 at 12 800 rows 63 -> 37 us.  The arithmetic per output element does not depend on the tile shape:
  // results are bit-identical.
  {
    const int row_tiles = cdiv(a->M, SBM), want = split_num_cus() * 3 / 2;
    const int cands[3] = {7, 4, 2};
    for (int ci = 0; ci < 3 && row_tiles * cdiv(a->No, nt * 16) < want; ++ci)
      if (cands[ci] < nt) nt = cands[ci];
  }
  if (nn2_ok(*a, ldn1, ldn2)) {
    // The three-MFMA form where the operand maxima are known (a_amax1 / a_amax2) and B's two-piece image exists or can be made
    if (nn2_h2_ok(*a)) {
      const int np = a->pieces == 1 ? 1 : 2;  // (1: the one-MFMA reduced-precision form, on request only)
      if (const void* pk = nn2_prepack_lookup(B1n, ldn1, a->K1, B2n, ldn2, a->K2, a->No, np)) return launch_nn2_prepacked(nt, *a, pk, stream, np);
      if (ws && aligned16(ws) && nn2_packed_ok(*a, ws_bytes, np)) return launch_nn2_packed(nt, *a, B1n, ldn1, B2n, ldn2, ws, stream, np);
    }
    // B pre-packed by the caller (qagnn_gemm_nn_prepack_f32: one launch for all weights of a step)?
    if (const void* pk = nn2_prepack_lookup(B1n, ldn1, a->K1, B2n, ldn2, a->K2, a->No)) return launch_nn2_prepacked(nt, *a, pk, stream);
    if (ws && nn2_packed_ok(*a, ws_bytes)) {
      QAGNN_REQUIRE(aligned16(ws), QAGNN_EINVAL, "gemm_nn_split: the pack workspace must be 16-byte aligned");
      return launch_nn2_packed(nt, *a, B1n, ldn1, B2n, ldn2, ws, stream);
    }
    return launch_nn2(nt, *a, B1n, ldn1, B2n, ldn2, stream);
  }
  switch (nt) {
    case 13: return launch_split<13>(*a, B1n, ldn1, B2n, ldn2, stream);
    case 8: return launch_split<8>(*a, B1n, ldn1, B2n, ldn2, stream);
    case 7: return launch_split<7>(*a, B1n, ldn1, B2n, ldn2, stream);
    case 4: return launch_split<4>(*a, B1n, ldn1, B2n, ldn2, stream);
    default: return launch_split<2>(*a, B1n, ldn1, B2n, ldn2, stream);
  }
}
