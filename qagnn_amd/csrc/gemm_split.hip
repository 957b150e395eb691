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

#include "common.h"

namespace qagnn {

typedef float f32x4s __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int SBK = 32, SBM = 128, SWAVES = 4, SRT = 2, STHR = SWAVES * 64;

__device__ __forceinline__ void split3(float x, uint32_t& h1, uint32_t& h2, uint32_t& h3) {
  const uint32_t u = __builtin_bit_cast(uint32_t, x);
  h1 = u & 0xFFFF0000u;
  const float r1 = x - __builtin_bit_cast(float, h1);  // exact
  h2 = __builtin_bit_cast(uint32_t, r1) & 0xFFFF0000u;
  const float r2 = r1 - __builtin_bit_cast(float, h2);  // exact, <= 8 significant bits
  h3 = __builtin_bit_cast(uint32_t, r2) & 0xFFFF0000u;
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

template <int NT, bool AFFINE>
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
    // unconditional loads from clamped addresses (branches around loads make hipcc fall back to vmcnt(0)); what lies past the
    // operand -- rows >= M, columns >= No, k >= K -- is zeroed with selects when the tile goes to LDS
    auto gload = [&](int kt) {
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
        const bool ok = kin && (!first || arow[p] >= 0) && (first || m0 + lr + p * 32 < a.M);
        v = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        store_split(As, A_EL, swz(lr + p * 32, kl), v);
      }
#pragma unroll
      for (int q = 0; q < B_IT; ++q) {
        const int col = lr + q * 32;
        if (col < BN) {
          const float4 v = (kin && n0 + col < a.No) ? rb[q] : make_float4(0.f, 0.f, 0.f, 0.f);
          store_split(Bs, B_EL, swz(col, kl), v);
        }
      }
    };

    gload(0);
    for (int kt = 0; kt < nkt; ++kt) {
      __syncthreads();  // the previous tile's fragment reads (or the previous output tile's slab reads) are done
      lstore(kt);
      __syncthreads();
      gload(min(kt + 1, nkt - 1));  // in flight under the MFMAs; past the last tile: a redundant reload, never stored
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
    }

    // epilogue: transpose 16 rows at a time through the wave's LDS slab, then whole-row 16-byte stores (as k_gemm_nn)
    float* const St = reinterpret_cast<float*>(smem_raw) + w * SLAB_ROWS * PS;
    constexpr int ROW_F4 = BN / 4, TILE_F4 = SLAB_ROWS * ROW_F4, ST_IT = (TILE_F4 + 63) / 64;
#pragma unroll
    for (int i = 0; i < SRT; ++i) {
      __syncthreads();  // k-loop reads (first pass) / the previous pass's slab reads are done before the slab is overwritten
      const int lr0 = (lane >> 4) * 4;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) St[(lr0 + r) * PS + j * 16 + (lane & 15)] = acc[i][j][r];
      __syncthreads();
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
  if (a.a_scale) k_gemm_nn_split<NT, true><<<grid, STHR, 0, stream>>>(b, B1n, ldn1, B2n, ldn2, ntiles);
  else k_gemm_nn_split<NT, false><<<grid, STHR, 0, stream>>>(b, B1n, ldn1, B2n, ldn2, ntiles);
  QAGNN_LAUNCH_CHECK("k_gemm_nn_split");
  return QAGNN_OK;
}

}  // namespace qagnn

using namespace qagnn;

extern "C" int qagnn_gemm_nn_split_f32(const qagnn_gemm_nn_args* a, const float* B1n, int32_t ldn1, const float* B2n, int32_t ldn2,
                                       qagnn_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
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
  if (nt16 >= 13) return launch_split<13>(*a, B1n, ldn1, B2n, ldn2, stream);
  if (nt16 >= 8) return launch_split<8>(*a, B1n, ldn1, B2n, ldn2, stream);
  if (nt16 >= 7) return launch_split<7>(*a, B1n, ldn1, B2n, ldn2, stream);
  if (nt16 >= 4) return launch_split<4>(*a, B1n, ldn1, B2n, ldn2, stream);
  return launch_split<2>(*a, B1n, ldn1, B2n, ldn2, stream);
}
