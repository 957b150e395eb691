// NN products of the stack on the bf16 matrix cores by exact operand splitting (see gemm_split.hip for the arithmetic), second
// kernel generation (round 4).  What changed against k_gemm_nn_split, and why (profiles/r3_run20_nn_sq_counters.txt: the matrix pipe
// was 54 % busy, a wave spent 23 % parked on barriers and 41 % stalled at issue; every k-tile cost two barriers, an LDS round trip of
// BOTH operands and ~370 VALU instructions per wave):
//
//  * The A operand (activations, streamed from HBM) never touches LDS.  A wave owns 32 rows of the output tile and nobody else
//    reads them, so its lanes load the MFMA A fragments straight from global memory -- lane l: row l & 15, the 8 consecutive k of
//    chunk l >> 4 = two 16-byte loads -- and split them in registers into the three bf16 fragments.  LDS traffic and the
//    split / store work of the A tile (16 KB per k-tile and block) are gone, and so is the second barrier.
//  * The B operand (weights, L2-resident) goes through LDS as before (all four waves need it), but DOUBLE-buffered and stored in
//    FRAGMENT order: the image of (column tile j, piece p) is one 1-KB block holding each lane's 16 bytes at a lane-linear slot
//    (XOR-permuted, below), so a fragment read is ONE conflict-free ds_read_b128 at base + immediate and the writes of the next
//    tile can be issued between the MFMAs of the current one.  One barrier per k-tile.
//  * The two K segments ([X | S]: K1 = 208, K2 = 112) are walked as ONE virtual K of 320 = 10 k-tiles (was 7 + 4 zero-padded tiles):
//    the single tile that straddles the segments is loaded first, synchronously, in the prologue (two loads per lane, one of them
//    answered with zeros by the bounds check, OR-ed); every tile of the steady-state loop lies in one segment and picks its buffer
//    descriptor with scalar selects.
//
// LDS slot permutation.  Fragment lane l = (x = l & 15, c = l >> 4) keeps its 16 bytes at slot (x ^ 2c) + 16c of the block.
// ds_read_b128 is serviced in four non-contiguous 16-lane groups (MI355X_MICROARCH.md, LDS), and every group then covers 16
// different slots mod 16 (checked by enumeration in tests/test_host_logic_emu.py::test_nn2_lds_slots).  The writes are 8-byte halves
// of a slot, ds_write_b64 is serviced in contiguous 16-lane groups with banks mod 32 (a 128-byte window): loader lanes 16g..16g+15
// hold two weight rows (x = 2r, 2r + 1) times eight float4 (c = kq >> 1, half = kq & 1), i.e. slots {x ^ 2c} = 8 different values
// mod 8, both halves each = every bank once.
//
//  * (the straddling tile, as built: one 64-bit flat load per destination with the segment chosen per lane -- see QAGNN_NN2_FIRST_LOADS)
//
// Shapes: a wave = 32 rows x NT column tiles of v_mfma_f32_16x16x32_bf16 (2 x NT x 4 accumulator registers), k-tile 32.
//   WV = 4: block = 128 rows x NT*16 columns, LDS 2 x NT x 3 KB (78 KB at NT = 13) => two blocks per CU.  They do NOT drift apart as
//     hoped: blocks b and b + 256 share a CU and run in lockstep (tools/cu_census.hip, profiles/r4_run16_nn2_stagger.txt), so the parts
//     of a k-tile add up instead of overlapping -- which is what the WV = 8 form below is for.
//   WV = 8 (PACKED, >= 10 k-tiles, >= 8 column tiles, >= one 256-row tile per CU): the staggered block, see the comment at the kernel.
//   PACKED: B arrives pre-split in the kernel's LDS image order (k_pack_b / k_pack_b_multi: once per product, or once per training step
//     for all weights through the registry of qagnn_gemm_nn_prepack_f32) and goes to LDS by DMA.
// NP (round 6): the arithmetic form.  NP = 3 is the exact 3 x bf16 split above (six MFMAs per product).  NP = 2 is the error-corrected
// TWO-piece fp16 split (Ootomo & Yokota 2022): x s = hi + lo, hi = fp16(x s), lo = fp16(x s - hi), 22 significant bits, and
// a b = [a_lo b_hi + a_hi b_lo + a_hi b_hi] / (s_a s_b) with the three products accumulated in fp32 -- THREE v_mfma_f32_16x16x32_f16 per
// tile pair, dropped term a_lo b_lo <= 2^-22 |a b|, i.e. a relative error of ~2^-21 per product where the bf16 form has 2^-23; both are below
// the sqrt(K) 2^-24 accumulation noise of any fp32 dot product at K >= 208.  fp16 has 5 exponent bits, so each operand carries an exact
// power-of-two scale s that puts its largest magnitude into [2^14, 2^15): A's comes from the caller (qagnn_gemm_nn_args.a_amax1 / a_amax2: device
// words holding the bit pattern of max |A|, produced by the kernel that wrote A; one common scale for [A1 | A2]), B's is computed per
// column tile by the packer and travels behind the image.  Elements below 2^-24 of the tensor's maximum lose relative (never absolute)
// accuracy: the error of every output is <= 2^-21 sum_k |a_k| |b_k| + 2^-38 K max|A| max|B_col|.  PACKED, WV = 4 only; a product whose A
// has no known maximum takes the NP = 3 kernels.
// Epilogues: PACKED without column statistics stores straight from the accumulators (the MFMAs are issued with their operands swapped,
// so that a lane holds four consecutive columns of a row); the others go through 16-row LDS slabs as in k_gemm_nn_split (whole-row
// 16-byte stores; bias / row table / accumulate / BatchNorm column statistics).
#include <stdlib.h>

#include "common.h"

namespace qagnn {


namespace nn2 {

// QAGNN_NN2_ABL (tools/nn2_ablate.hip only; numerically wrong, timing only): bit 0 no loads in the steady loop, bit 1 no B split / store,
// bit 2 no A split, bit 3 no barrier, bit 4 no fragment reads, bit 5 no MFMAs, bit 6 no stores of the result
#ifndef QAGNN_NN2_ABL
#define QAGNN_NN2_ABL 0
#endif

constexpr int BK = 32;
constexpr uint32_t OOB = 0x80000000u;  // beyond any operand this kernel is launched on: the buffer load answers with zeros

__device__ __forceinline__ u32x4s bload(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(u32x4s, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, (int)soff, 0));
}

// x = h1 + h2 + h3 exactly, each h a bf16 number carried in the high half of a dword (see gemm_split.hip)
__device__ __forceinline__ void split3(uint32_t u, uint32_t& h1, uint32_t& h2, uint32_t& h3) {
  h1 = u & 0xFFFF0000u;
  const float r1 = __builtin_bit_cast(float, u) - __builtin_bit_cast(float, h1);
  h2 = __builtin_bit_cast(uint32_t, r1) & 0xFFFF0000u;
  const float r2 = r1 - __builtin_bit_cast(float, h2);
  h3 = __builtin_bit_cast(uint32_t, r2);
}
__device__ __forceinline__ uint32_t pack_hi(uint32_t lo, uint32_t hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }

template <bool AFFINE, int NP = 2>
__device__ __forceinline__ void split_frag2(const u32x4s (&r)[2], u32x4s (&f)[NP], const float* __restrict__ sc, const float* __restrict__ sh, float lo,
                                            float s) {
  uint32_t ph[4], pl[4];
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    // (through scalars: __builtin_bit_cast applied to a vector ELEMENT reads the vector's first four bytes whatever the index -- hipcc 7.2)
    const uint32_t ux = r[e >> 2][e & 3], uy = r[(e + 1) >> 2][(e + 1) & 3];
    float x = __builtin_bit_cast(float, ux), y = __builtin_bit_cast(float, uy);
    if constexpr (AFFINE) {
      x = fmaxf(fmaf(x, sc[e], sh[e]), lo);
      y = fmaxf(fmaf(y, sc[e + 1], sh[e + 1]), lo);
    }
    if constexpr (NP == 2) split2(x, y, s, ph[e >> 1], pl[e >> 1]);
    else split1(x, y, s, ph[e >> 1]);
  }
  f[0] = (u32x4s){ph[0], ph[1], ph[2], ph[3]};
  if constexpr (NP == 2) f[1] = (u32x4s){pl[0], pl[1], pl[2], pl[3]};
}

// 8 consecutive k of one row (two 16-byte loads) -> the three bf16x8 fragments
template <bool AFFINE>
__device__ __forceinline__ void split_frag(const u32x4s (&r)[2], u32x4s (&f)[3], const float* __restrict__ sc, const float* __restrict__ sh, float lo) {
  uint32_t h1[8], h2[8], h3[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    uint32_t u = r[e >> 2][e & 3];
    if constexpr (AFFINE) u = __builtin_bit_cast(uint32_t, fmaxf(fmaf(__builtin_bit_cast(float, u), sc[e], sh[e]), lo));
    split3(u, h1[e], h2[e], h3[e]);
  }
  const u32x4s p1 = {pack_hi(h1[0], h1[1]), pack_hi(h1[2], h1[3]), pack_hi(h1[4], h1[5]), pack_hi(h1[6], h1[7])};
  const u32x4s p2 = {pack_hi(h2[0], h2[1]), pack_hi(h2[2], h2[3]), pack_hi(h2[4], h2[5]), pack_hi(h2[6], h2[7])};
  const u32x4s p3 = {pack_hi(h3[0], h3[1]), pack_hi(h3[2], h3[3]), pack_hi(h3[4], h3[5]), pack_hi(h3[6], h3[7])};
  f[0] = p1;
  f[1] = p2;
  f[2] = p3;
}

// one float4 of a weight row (4 consecutive k) -> 8 bytes per piece at `dst` (+ 1024 per piece)
__device__ __forceinline__ void store_b4(unsigned char* __restrict__ dst, u32x4s v) {
  uint32_t a1, a2, a3, b1, b2, b3, c1, c2, c3, d1, d2, d3;
  split3(v[0], a1, a2, a3);
  split3(v[1], b1, b2, b3);
  split3(v[2], c1, c2, c3);
  split3(v[3], d1, d2, d3);
  *reinterpret_cast<uint2*>(dst) = make_uint2(pack_hi(a1, b1), pack_hi(c1, d1));
  *reinterpret_cast<uint2*>(dst + 1024) = make_uint2(pack_hi(a2, b2), pack_hi(c2, d2));
  *reinterpret_cast<uint2*>(dst + 2048) = make_uint2(pack_hi(a3, b3), pack_hi(c3, d3));
}

#define QAGNN_NN2_SIX_T(C, AF, BF) C = mfma_pieces<NP, true>(AF, BF, C);
#define QAGNN_NN2_SIX(C, AF, BF) C = mfma_pieces<NP>(AF, BF, C);

// Per column tile 12 MFMAs (hipcc interleaves the two row tiles' accumulate chains by itself), then the split work of a B load round as
// one clump (ORDER is a vestigial template argument, always 0: the pinned MFMA / 2 VALU interleave it selected measured no faster).
// PACKED: B arrives pre-split in the kernel's own LDS image order (k_pack_b below: [k-tile of the walk][column tile][piece][lane] x 16 B,
// zero-filled past K and No), B1n = that buffer, ldn1 = column tiles per k-tile.  A k-tile of B is then NT * 3 contiguous KB that go
// to LDS by DMA (global_load_lds, 1 KB per wave-instruction): no registers, no split arithmetic, no ds_write for B in the k-loop
// (tools/nn2_ablate.hip: the in-kernel split of B costs 19 % of the projection, and it is repeated by all 500 row tiles).
//
// WV = 8 (PACKED only): the STAGGERED block.  Two 4-wave blocks that share a CU run in lockstep (same work, fair arbitration; neither a
// static priority for one of them nor an initial skew of up to 6 000 cycles changes the kernel time, tools/cu_census.hip and
// profiles/r4_run16_nn2_stagger.txt), so their MFMA phases coincide and so do their load / split phases: the matrix pipe idles while
// both split.  Here ONE block of 8 waves owns the CU (256 rows), two waves per SIMD, one of each SIMD's pair in phase group 0 and the
// other in group 1, and every k-tile is two barrier-separated phases per wave:
//     N: split the A fragments of tile t out of the raw registers, issue the loads of tile t + 1 (A) and this wave's share of the DMA
//        of B tile t + 1 + g, read the first two column tiles' B fragments;      M: the 12 NT MFMAs of tile t, fragment reads only.
// Group 1 passes one extra barrier before its first phase and group 0 one after its last, so at every moment one wave of a SIMD is in
// M and the other in N: the matrix pipe sees one uninterrupted MFMA stream, the split / load work runs under it.  B is shared by all
// 8 waves (half the DMA instructions per wave) in a ring of THREE images: tile t is read in the global phases 2t .. 2t + 2 (group 0:
// end of N_t and M_t, group 1 one phase later), group 0 issues its share of tile t's DMA in its N_(t-1) (phase 2t - 2) and waits for it
// at the end of M_(t-1); group 1 would be too late there and issues in ITS N_(t-2) (phase 2t - 3; the image's previous tenant, tile t - 3,
// was last read in phase 2t - 4).  The loop's back edge sits right after the barrier that ends M: every load issued in the previous N has
// had a whole M phase to land, so nothing is in flight across it.
template <int NT, bool STATS, int WV, int NP = 3>
constexpr int nn2_lds_bytes(int aff_bytes) {
  const int ring = (WV == 8 ? 3 : 2) * NT * NP * 1024 + aff_bytes;
  const int epi = WV * 16 * (NT * 16 + 4) * 4 + (STATS ? WV * 2 * NT * 16 * 4 : 0);
  return ring > epi ? ring : epi;
}

template <int NT, bool AFFINE, bool STATS, int NP, bool PACKED, int WV = 4>
__global__ __launch_bounds__(WV * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_gemm_nn2(qagnn_gemm_nn_args a, const float* __restrict__ B1n,
                                                                                          int ldn1, const float* __restrict__ B2n, int ldn2,
                                                                                          int ntiles) {
  static_assert(WV == 4 || (WV == 8 && PACKED), "the staggered block takes B by DMA only");
  static_assert(NP == 3 || (NP <= 2 && PACKED && WV == 4), "the scaled fp16 forms (two pieces: three MFMAs; one piece: one MFMA): packed B, 4-wave blocks");
  static_assert(!(AFFINE && STATS), "no product needs both");
  constexpr bool STAG = WV == 8;
  constexpr bool DIRECT = PACKED && !STATS;  // (see below)
  // DIRECT: the MFMAs are issued with their operands swapped (the transposed 16 x 16 tile: lane (x, c) then holds FOUR CONSECUTIVE
  // COLUMNS 4c .. 4c + 3 of row x), so the epilogue stores its accumulators straight from the registers -- 16 rows x 64 bytes per
  // instruction, column tiles j and j + 1 completing each other's 128-byte lines -- without the LDS transpose and its four barriers.
  // That leaves LDS free at the end of a tile: the first loads of the NEXT tile are issued in front of the stores, and the stores
  // (address-predicated buffer stores, a fixed 2 NT per wave, so that s_waitcnt vmcnt(2 NT) means "everything older has landed") drain
  // under the next tile's k-loop.  With one block per CU nothing else would run under either.  (The 4-wave packed blocks store the same
  // way, without the early loads.)
  constexpr int THR = WV * 64, BM = WV * 32;
  constexpr int BN = NT * 16;
  constexpr int IMG = NT * NP * 1024;  // bytes of one B image set: [column tile][piece][64 slots x 16 B]
  constexpr int RING_B = (STAG ? 3 : 2) * IMG;
  constexpr int BR = (NT + 1) / 2;     // load rounds of the B tile: 32 weight rows (output columns) x 8 float4 per round
  constexpr int PS = BN + 4, SLAB_ROWS = 16;
  constexpr int SLAB_B = WV * SLAB_ROWS * PS * 4;
  static_assert(DIRECT || SLAB_B + (STATS ? WV * 2 * BN * 4 : 0) <= (AFFINE ? RING_B : nn2_lds_bytes<NT, STATS, WV, NP>(0)), "the epilogue reuses the k-loop's LDS");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* const aff = reinterpret_cast<float*>(smem + RING_B);  // AFFINE: scale[KA] | shift[KA], KA = K1 rounded up to 32, zero-filled

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = a.M, No = a.No, K1 = a.K1, K2 = a.K2;
  const int ncb = (No + BN - 1) / BN;
  // the k-tile walk: [the tile that straddles the segments], the whole tiles of segment 1, the tiles of segment 2 from s2 on
  const int r1 = K2 > 0 ? (K1 & 31) : 0;
  const int mixi = r1 != 0 ? 1 : 0;
  const int n1 = mixi ? (K1 >> 5) : ((K1 + 31) >> 5);
  const int s2 = mixi ? 32 - r1 : 0;
  const int n2 = K2 > s2 ? ((K2 - s2 + 31) >> 5) : 0;
  const int nkt = mixi + n1 + n2;
  const int KA = (K1 + 31) & ~31;

  // all four operands through buffer descriptors (built once, wave-uniform); a null second segment gets an empty range
  const __amdgpu_buffer_rsrc_t rA1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A1), 0, (a.a_rowidx ? (int)a.a_rows : M) * a.lda1 * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.A2), 0, K2 > 0 ? M * a.lda2 * 4 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B1n), 0, No * ldn1 * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B2n), 0, K2 > 0 ? No * ldn2 * 4 : 0, 0x00020000);
  const uint32_t bstep1 = 128u * (uint32_t)ldn1, bstep2 = 128u * (uint32_t)ldn2;  // bytes between the weight rows of two B load rounds
  const unsigned char* const pk = reinterpret_cast<const unsigned char*>(B1n);   // PACKED

  if constexpr (AFFINE) {  // once per block: the BatchNorm scale / shift of A1's columns
    for (int i = tid; i < 2 * KA; i += THR) {
      const int k = i < KA ? i : i - KA;
      aff[i] = k < K1 ? (i < KA ? a.a_scale[k] : a.a_shift[k]) : 0.f;
    }
  }

  // NP == 2: the operands' power-of-two scales (see the header).  A: one common scale from max(max|A1|, max|A2|); B: per column tile,
  // written behind the image (and its 13 tiles of slack) by the packer
  uint32_t fa = 127u;
  float sa = 1.f;
  if constexpr (NP <= 2) {
    uint32_t mb = a.a_amax1[0];
    if (K2 > 0 && a.a_amax2) mb = max(mb, a.a_amax2[0]);
    fa = __builtin_amdgcn_readfirstlane(h2_scale_field(mb));
    sa = h2_field_to_scale(fa);
  }
  const int ac = lane >> 4;                 // this lane's k chunk (8 k) of a tile, A side
  const int nl = tid >> 3, kq = tid & 7;    // B loader: weight row of a round, float4 index inside the k-tile
  // LDS byte offset of the loader's (row, float4) inside a round's two column tiles, and of the fragment reads (see the header)
  const int bx = nl & 15, bc = kq >> 1;
  const int wr_off = ((nl >> 4) * 3 * 64 + ((bx ^ (2 * bc)) + 16 * bc)) * 16 + (kq & 1) * 8;
  const int rd_off = (((lane & 15) ^ (2 * ac)) + 16 * ac) * 16;
  // the last load round of an odd NT covers one column tile only: the loaders of its second half (waves 2, 3) sit it out
  const bool last_round_on = (NT & 1) == 0 || w < 2;

  // STAG: the phase group of this wave.  256 registers per wave = two waves per SIMD; the first wave to arrive on a SIMD joins group 0,
  // the second group 1 (any split is correct -- the groups only decide who is in which phase -- an even one is what overlaps)
  int g = 0;
  if constexpr (STAG) {
    int* const cnt = reinterpret_cast<int*>(smem);
    if (tid < 4) cnt[tid] = 0;
    __syncthreads();
    const int simd = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4) & 3;  // HW_ID[5:4] = SIMD_ID
    int slot = 0;
    if (lane == 0) slot = atomicAdd(&cnt[simd], 1);
    g = __builtin_amdgcn_readfirstlane(slot) & 1;
    __syncthreads();
  }

  int tile = 0, m0 = 0, n0 = 0;
  uint32_t arow1[2], arow2[2];  // per-lane byte offsets of the operand rows (the k position comes in as the scalar offset); rows outside: OOB
#define QAGNN_NN2_SET_TILE(VB)                                                                                           \
  {                                                                                                                      \
    tile = a.xcd_remap ? xcd_remap((VB), ntiles) : (VB);                                                                 \
    m0 = (tile / ncb) * BM;                                                                                              \
    n0 = (tile % ncb) * BN;                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                      \
      const int row = m0 + w * 32 + i * 16 + (lane & 15);                                                                \
      /* (a_rowidx: row m of A1 is table row a_rowidx[m], negative = a zero row -- nn2_ok: the table lies below 2 GB) */   \
      const int64_t grow = (a.a_rowidx && row < M) ? a.a_rowidx[row] : (int64_t)row;                                     \
      arow1[i] = (row < M && grow >= 0) ? (uint32_t)grow * (uint32_t)a.lda1 * 4u + (uint32_t)ac * 32u : OOB;             \
      arow2[i] = row < M ? (uint32_t)row * (uint32_t)a.lda2 * 4u + (uint32_t)ac * 32u : OOB;                             \
    }                                                                                                                    \
  }
  bool loaded = false;  // DIRECT: this tile's first loads were issued in front of the previous tile's stores
  u32x4s ra[2][2];
  for (int vb = blockIdx.x; vb < ntiles; vb += gridDim.x) {
    if (!loaded) QAGNN_NN2_SET_TILE(vb)
    const int nvalid = min(BN, No - n0) - nl;  // load round q is inside the matrix for this loader iff 32 q < nvalid

    f32x4s acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4s){0.f, 0.f, 0.f, 0.f};

    const uint32_t bbase1 = (uint32_t)(n0 + nl) * (uint32_t)ldn1 * 4u + (uint32_t)kq * 16u;
    const uint32_t bbase2 = (uint32_t)(n0 + nl) * (uint32_t)ldn2 * 4u + (uint32_t)kq * 16u;

    u32x4s rb[PACKED ? 1 : BR];
    u32x4s af[2][NP];

    // loads of a tile that lies in one segment (steady state); `it` past the last tile: everything out of range, zeros
#define QAGNN_NN2_GLOAD(IT)                                                                                              \
    {                                                                                                                    \
      const int u_ = (IT)-mixi;                                                                                          \
      if (u_ < n1) {                                                                                                     \
        const int kb_ = u_ * 32;                                                                                         \
        if constexpr (!PACKED) {                                                                                         \
          const uint32_t bo_ = kb_ + kq * 4 < K1 ? bbase1 : OOB;                                                         \
          _Pragma("unroll") for (int q = 0; q < BR; ++q)                                                                 \
              rb[q] = bload(rB1, q * 32 < nvalid ? bo_ : OOB, (uint32_t)kb_ * 4u + (uint32_t)q * bstep1);                \
        }                                                                                                                \
        const bool ain_ = kb_ + ac * 8 < K1;                                                                             \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                  \
          const uint32_t off_ = ain_ ? arow1[i] : OOB;                                                                   \
          ra[i][0] = bload(rA1, off_, (uint32_t)kb_ * 4u);                                                               \
          ra[i][1] = bload(rA1, off_, (uint32_t)kb_ * 4u + 16u);                                                         \
        }                                                                                                                \
      } else {                                                                                                           \
        const int kb_ = s2 + (u_ - n1) * 32;                                                                             \
        if constexpr (!PACKED) {                                                                                         \
          const uint32_t bo_ = kb_ + kq * 4 < K2 ? bbase2 : OOB;                                                         \
          _Pragma("unroll") for (int q = 0; q < BR; ++q)                                                                 \
              rb[q] = bload(rB2, q * 32 < nvalid ? bo_ : OOB, (uint32_t)kb_ * 4u + (uint32_t)q * bstep2);                \
        }                                                                                                                \
        const bool ain_ = kb_ + ac * 8 < K2;                                                                             \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                  \
          const uint32_t off_ = ain_ ? arow2[i] : OOB;                                                                   \
          ra[i][0] = bload(rA2, off_, (uint32_t)kb_ * 4u);                                                               \
          ra[i][1] = bload(rA2, off_, (uint32_t)kb_ * 4u + 16u);                                                         \
        }                                                                                                                \
      }                                                                                                                  \
    }
    // PACKED: the NT * 3 KB of k-tile IT -> the image BUF by DMA, the 1-KB blocks dealt out over the waves
#define QAGNN_NN2_GLDS(IT, BUF)                                                                                          \
    {                                                                                                                    \
      const unsigned char* src_ = pk + ((int64_t)(IT) * ldn1 + n0 / 16) * (NP * 1024) + lane * 16;                       \
      _Pragma("unroll") for (int b = 0; b < (NT * NP + WV - 1) / WV; ++b) {                                              \
        const int blk_ = w + b * WV;                                                                                     \
        if (blk_ < NT * NP)                                                                                              \
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_ + blk_ * 1024),         \
                                           (__attribute__((address_space(3))) void*)((BUF) + blk_ * 1024), 16, 0, 0);    \
      }                                                                                                                  \
    }
    // one load round of B -> the image `BUF` (uniform condition: see last_round_on)
#define QAGNN_NN2_STORE_B(Q, BUF)                                                       \
    {                                                                                   \
      if constexpr (!PACKED) {                                                          \
        if ((Q) + 1 < BR || last_round_on) store_b4((BUF) + wr_off + (Q) * (2 * 3 * 1024), rb[Q]); \
      }                                                                                 \
    }
    // IT: the tile the fragments belong to.  AFFINE applies to segment 1 only: a segment-2 chunk gets scale 1, shift 0 and the floor
    // -inf instead of 0, i.e. passes through unchanged; k past K1 reads the zero fill (relu(0 * x + 0) = 0)
#define QAGNN_NN2_SPLIT_A(IT)                                                                                            \
    {                                                                                                                    \
      if constexpr (AFFINE) {                                                                                            \
        const int u_ = (IT)-mixi;                                                                                        \
        const bool mixt_ = mixi && (IT) == 0;                                                                            \
        const bool s1_ = mixt_ ? (ac * 8 < r1) : (u_ < n1);                                                              \
        const int k_ = min(mixt_ ? K1 - r1 + ac * 8 : u_ * 32 + ac * 8, KA - 8);                                         \
        const float4 c0 = ld4(aff + k_), c1 = ld4(aff + k_ + 4), h0 = ld4(aff + KA + k_), h1 = ld4(aff + KA + k_ + 4);   \
        const float sc[8] = {s1_ ? c0.x : 1.f, s1_ ? c0.y : 1.f, s1_ ? c0.z : 1.f, s1_ ? c0.w : 1.f,                     \
                             s1_ ? c1.x : 1.f, s1_ ? c1.y : 1.f, s1_ ? c1.z : 1.f, s1_ ? c1.w : 1.f};                    \
        const float sh[8] = {s1_ ? h0.x : 0.f, s1_ ? h0.y : 0.f, s1_ ? h0.z : 0.f, s1_ ? h0.w : 0.f,                     \
                             s1_ ? h1.x : 0.f, s1_ ? h1.y : 0.f, s1_ ? h1.z : 0.f, s1_ ? h1.w : 0.f};                    \
        const float lo = s1_ ? 0.f : -INFINITY;                                                                          \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                  \
          if constexpr (NP <= 2) split_frag2<true, NP>(ra[i], af[i], sc, sh, lo, sa);                                    \
          else split_frag<true>(ra[i], af[i], sc, sh, lo);                                                               \
        }                                                                                                                \
      } else {                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                  \
          if constexpr (NP <= 2) split_frag2<false, NP>(ra[i], af[i], nullptr, nullptr, 0.f, sa);                        \
          else split_frag<false>(ra[i], af[i], nullptr, nullptr, 0.f);                                                   \
        }                                                                                                                \
      }                                                                                                                  \
    }
    // the MFMAs of one k-tile out of the image CUR; STORE: the next tile's B rows (loaded at the top of this iteration) go to the
    // image NXT behind the LAST BR column tiles, one load round each.  Fragments are read one column tile ahead, nothing else moves
    // across column tiles.
#define QAGNN_NN2_MFMA_TILE(CUR, NXT, STORE)                                                                             \
    {                                                                                                                    \
      __builtin_amdgcn_s_setprio(1);                                                                                     \
      u32x4s bfa[NP], bfb[NP];                                                                                           \
      _Pragma("unroll") for (int p = 0; p < NP; ++p) bfa[p] = *reinterpret_cast<const u32x4s*>((CUR) + p * 1024 + rd_off); \
      _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                                   \
        u32x4s(&bf)[NP] = (j & 1) ? bfb : bfa;                                                                           \
        u32x4s(&bn)[NP] = (j & 1) ? bfa : bfb;                                                                           \
        if (j + 1 < NT && !(QAGNN_NN2_ABL & 16)) {                                                                       \
          _Pragma("unroll") for (int p = 0; p < NP; ++p)                                                                 \
              bn[p] = *reinterpret_cast<const u32x4s*>((CUR) + ((j + 1) * NP + p) * 1024 + rd_off);                      \
        }                                                                                                                \
        if constexpr (!(QAGNN_NN2_ABL & 32)) {                                                                           \
          _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                \
            f32x4s c = acc[i][j];                                                                                        \
            if constexpr (DIRECT) { QAGNN_NN2_SIX_T(c, af[i], bf) } else { QAGNN_NN2_SIX(c, af[i], bf) }                  \
            acc[i][j] = c;                                                                                               \
          }                                                                                                              \
        } else {                                                                                                         \
          _Pragma("unroll") for (int p = 0; p < NP; ++p) asm volatile("" ::"v"(bf[p]));                                  \
        }                                                                                                                \
        if ((STORE) && j >= NT - BR) QAGNN_NN2_STORE_B(j - (NT - BR), NXT)                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
      }                                                                                                                  \
      __builtin_amdgcn_s_setprio(0);                                                                                     \
    }

    // (Round 6, visit 18, measured and removed: the PACKED 4-wave blocks with their fragment reads as inline assembly two column tiles
    // ahead and counted lgkmcnt waits, as in the staggered block below.  hipcc issues the reads of column tile j + 1 behind four of tile
    // j's six MFMAs and waits lgkmcnt(0) two MFMAs later; with the reads hoisted the ISA is six back-to-back MFMAs per column tile behind
    // lgkmcnt(4) -- and the kernels take the same time: projection 95.3 -> 100.8 us alone, NN products 2.12 ms per step either way
    // (profiles/r6_run18_nn2_asm_frags.txt).  The LDS round trip is not what the MFMA phases wait for.)
    // STAG: the B fragments of column tile J of the image at LDS address ADDR (+ this lane's slot); the MFMAs of one k-tile with the
    // fragments read TWO column tiles ahead (a ring of three register sets; sets 0 and 1 arrive in flight from the N phase), so that the
    // one wave of the SIMD that is in its M phase never waits for LDS.  Reads and waits are inline assembly: hipcc's own counter waits for
    // lgkmcnt(0) every third column tile -- right behind three fresh reads -- where lgkmcnt(6) is what the data flow needs.  The raw
    // barriers: the compiler must not add its vmcnt(0), the loads issued in N land under M.
#define QAGNN_NN2_FRAG(DST, ADDR, J)                                                                                     \
    {                                                                                                                    \
      _Pragma("unroll") for (int p = 0; p < 3; ++p)                                                                      \
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST[p]) : "v"(ADDR), "i"(((J) * 3 + p) * 1024));           \
    }
    // the fragments of column tile j are the oldest reads in flight; behind them: tiles j + 1 and j + 2 (LDS returns in order)
#define QAGNN_NN2_FRAG_WAIT(F, N) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(F[0]), "+v"(F[1]), "+v"(F[2]))
#define QAGNN_NN2_MFMA_TILE3(ADDR)                                                                                       \
    {                                                                                                                    \
      __builtin_amdgcn_s_setprio(1);                                                                                     \
      _Pragma("unroll") for (int j = 0; j < NT; ++j) {                                                                   \
        u32x4s(&bf)[3] = bfr[j % 3];                                                                                     \
        if (j + 2 < NT) {                                                                                                \
          QAGNN_NN2_FRAG(bfr[(j + 2) % 3], ADDR, j + 2)                                                                  \
          QAGNN_NN2_FRAG_WAIT(bf, 6);                                                                                    \
        } else if (j + 1 < NT) {                                                                                         \
          QAGNN_NN2_FRAG_WAIT(bf, 3);                                                                                    \
        } else {                                                                                                         \
          QAGNN_NN2_FRAG_WAIT(bf, 0);                                                                                    \
        }                                                                                                                \
        if constexpr (!(QAGNN_NN2_ABL & 32)) {                                                                           \
          _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                \
            f32x4s c = acc[i][j];                                                                                        \
            if constexpr (DIRECT) { QAGNN_NN2_SIX_T(c, af[i], bf) } else { QAGNN_NN2_SIX(c, af[i], bf) }                  \
            acc[i][j] = c;                                                                                               \
          }                                                                                                              \
        }                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
      }                                                                                                                  \
      __builtin_amdgcn_s_setprio(0);                                                                                     \
    }
    // (no lgkmcnt wait in front of the N phase's barrier: the fragment reads stay in flight across it and the compiler's own counter
    // stays exact; sched_barrier: the split arithmetic must not sink behind the barrier into the M phase)
#define QAGNN_NN2_BAR                            \
    {                                            \
      __builtin_amdgcn_sched_barrier(0);         \
      asm volatile("s_barrier" ::: "memory");    \
      __builtin_amdgcn_sched_barrier(0);         \
    }
#define QAGNN_NN2_BAR_VMN /* everything but the last 2 NT vector-memory operations (the previous tile's stores) has landed */ \
    {                                                                                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                                                   \
      asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(2 * NT) : "memory");                                         \
      __builtin_amdgcn_sched_barrier(0);                                                                                   \
    }
#define QAGNN_NN2_BAR_VM                                         \
    {                                                            \
      __builtin_amdgcn_sched_barrier(0);                         \
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory"); \
      __builtin_amdgcn_sched_barrier(0);                         \
    }

    // the first loads of a tile: A of k-tile 0 into the raw registers (B too without PACKED); STAG: B tile 0 (and group 1's share of
    // tile 1) by DMA.  The straddling tile: k in [K1 - r1, K1) of segment 1, then [0, 32 - r1) of segment 2 (nn2_ok: K2 >= 32 - r1).  One
    // 64-bit flat load per destination, the segment chosen per lane; rows past M and columns past No read the last valid row instead
    // (their products are never stored), so nothing needs zeroing and no second set of registers is in flight
#define QAGNN_NN2_FIRST_LOADS                                                                                            \
    {                                                                                                                    \
      if (mixi) {                                                                                                        \
        const int kbq = kq * 4, kaq = ac * 8;                                                                            \
        const bool b1 = kbq < r1, a1 = kaq < r1;                                                                         \
        if constexpr (!PACKED) {                                                                                         \
          _Pragma("unroll") for (int q = 0; q < BR; ++q) {                                                               \
            const int64_t col = min(n0 + nl + q * 32, No - 1);                                                           \
            const float* p = b1 ? B1n + col * ldn1 + (K1 - r1 + kbq) : B2n + col * ldn2 + (kbq - r1);                    \
            rb[q] = *reinterpret_cast<const u32x4s*>(p);                                                                 \
          }                                                                                                              \
        }                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                  \
          const int64_t row = min(m0 + w * 32 + i * 16 + (lane & 15), M - 1);                                            \
          const float* p = a1 ? a.A1 + row * a.lda1 + (K1 - r1 + kaq) : a.A2 + row * a.lda2 + (kaq - r1);                \
          ra[i][0] = *reinterpret_cast<const u32x4s*>(p);                                                                \
          ra[i][1] = *reinterpret_cast<const u32x4s*>(p + 4);                                                            \
        }                                                                                                                \
      } else {                                                                                                           \
        QAGNN_NN2_GLOAD(0)                                                                                               \
      }                                                                                                                  \
      if constexpr (STAG) {                                                                                              \
        QAGNN_NN2_GLDS(0, smem)                                                                                          \
        if (g == 1 && nkt > 1) QAGNN_NN2_GLDS(1, smem + IMG)                                                             \
      }                                                                                                                  \
    }

    if constexpr (!(DIRECT && STAG)) __syncthreads();  // the previous output tile's slab / last-image reads (and the scale / shift fill) are done
    // ---- prologue: tile 0 into image 0 / the fragment registers, tile 1 in flight
    if (!loaded) QAGNN_NN2_FIRST_LOADS
    if constexpr (STAG) {
      if (loaded) {
        QAGNN_NN2_BAR_VMN           // (the previous tile's stores stay in flight)
      } else {
        QAGNN_NN2_BAR_VM            // tile 0 (and group 1's share of tile 1) has landed
      }
      if (g == 1) QAGNN_NN2_BAR     // group 1 runs one phase behind from here on
      u32x4s bfr[3][3];
      const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + (uint32_t)rd_off;
      for (int it = 0; it < nkt; ++it) {
        const uint32_t cur = lds0 + (uint32_t)((it % 3) * IMG);
        // N: fragments of tile it, loads of tile it + 1 (A) / it + 1 + g (this wave's share of B)
        if ((QAGNN_NN2_ABL & 4) == 0 || it == 0) QAGNN_NN2_SPLIT_A(it)
        QAGNN_NN2_FRAG(bfr[0], cur, 0)
        if constexpr (NT > 1) QAGNN_NN2_FRAG(bfr[1], cur, 1)
        if constexpr (!(QAGNN_NN2_ABL & 1)) {
          if (it + 1 < nkt) QAGNN_NN2_GLOAD(it + 1)
        }
        if constexpr (!(QAGNN_NN2_ABL & 2)) {
          const int tb = it + 1 + g;
          if (tb < nkt) QAGNN_NN2_GLDS(tb, smem + (tb % 3) * IMG)
        }
        QAGNN_NN2_BAR
        // M
        QAGNN_NN2_MFMA_TILE3(cur)
        QAGNN_NN2_BAR_VM  // this wave's loads of the N phase have landed; the other group moves on to its M
      }
      if (g == 0) QAGNN_NN2_BAR
    } else {
    if constexpr (PACKED) {
      QAGNN_NN2_GLDS(0, smem)
    } else {
#pragma unroll
      for (int q = 0; q < BR; ++q) QAGNN_NN2_STORE_B(q, smem)
    }
    QAGNN_NN2_SPLIT_A(0)
    __syncthreads();
    // Steady state.  The loads of tile it + 1 are issued at the TOP of iteration it and consumed inside it (B behind the last column
    // tiles, A after the MFMAs): no load is in flight across the loop's back edge -- with the loads issued at the bottom, hipcc put
    // copies of their destination registers (and with them an s_waitcnt vmcnt(0)) right behind the barrier of every k-tile.
    for (int it = 0; it + 1 < nkt; ++it) {
      unsigned char* const cur = smem + (it & 1) * IMG;
      unsigned char* const nxt = smem + ((it & 1) ^ 1) * IMG;
      if constexpr (PACKED && !(QAGNN_NN2_ABL & 2)) QAGNN_NN2_GLDS(it + 1, nxt)
      if constexpr (!(QAGNN_NN2_ABL & 1)) QAGNN_NN2_GLOAD(it + 1)
      QAGNN_NN2_MFMA_TILE(cur, nxt, !PACKED && !(QAGNN_NN2_ABL & 2))   // + tile it + 1's B rows -> nxt
      if constexpr (!(QAGNN_NN2_ABL & 4)) QAGNN_NN2_SPLIT_A(it + 1)  // tile it + 1's A fragments (the MFMAs above were the last readers of af)
      if constexpr (!(QAGNN_NN2_ABL & 8)) __syncthreads();  // nxt is complete; everybody is done reading cur
    }
    {
      unsigned char* const cur = smem + ((nkt - 1) & 1) * IMG;
      QAGNN_NN2_MFMA_TILE(cur, cur, false)
    }
    }
    if constexpr (NP <= 2) {  // undo the operand scales: one exact power of two per column tile
      const uint32_t* const bfield = reinterpret_cast<const uint32_t*>(pk + ((int64_t)nkt * ldn1 + 13) * (NP * 1024)) + n0 / 16;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float inv = h2_inv_scale(fa, bfield[j]);
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][j] *= inv;
      }
    }
    if constexpr (DIRECT) {
      // ---- STAG: the next tile's first loads (every wave is past its last fragment read: the ring is free); then this tile's stores
      const int em0 = m0, en0 = n0;
      if constexpr (STAG) {
        loaded = vb + (int)gridDim.x < ntiles;
        if (loaded) {
          QAGNN_NN2_SET_TILE(vb + (int)gridDim.x)
          QAGNN_NN2_FIRST_LOADS
        }
      }
      const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(a.C, 0, M * a.ldc * 4, 0x00020000);
      const int c4 = 4 * (lane >> 4);
      // product + bias + table row + old value, in the slab epilogue's order; every kind of addend is fetched for all column tiles
      // at once (a column past No reads the last four columns instead and is not stored: No % 4 == 0), and ALL addends are in
      // before the first store is issued -- waiting for a load behind a store would wait for the store's acknowledgement too
      uint32_t crow[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = em0 + w * 32 + i * 16 + (lane & 15);
        const bool rok = row < M;
        crow[i] = rok ? (uint32_t)row * (uint32_t)a.ldc * 4u : OOB;
        f32x4s t[NT];
        if (a.bias) {
#pragma unroll
          for (int j = 0; j < NT; ++j) t[j] = __builtin_bit_cast(f32x4s, ld4(a.bias + min(en0 + j * 16 + c4, No - 4)));
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] += t[j];
        }
        if (a.rowtab) {
          const float* const trow = a.rowtab + (int64_t)a.rowidx[rok ? row : 0] * a.ldt;
#pragma unroll
          for (int j = 0; j < NT; ++j) t[j] = __builtin_bit_cast(f32x4s, ld4(trow + min(en0 + j * 16 + c4, No - 4)));
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] += t[j];
        }
        if (a.accumulate) {
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int col = en0 + j * 16 + c4;
            t[j] = __builtin_bit_cast(f32x4s, bload(rC, rok && col < No ? crow[i] + (uint32_t)col * 4u : OOB, 0u));
          }
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[i][j] += t[j];
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int col = en0 + j * 16 + c4;
          const uint32_t off = crow[i] != OOB && col < No ? crow[i] + (uint32_t)col * 4u : OOB;
          if constexpr (QAGNN_NN2_ABL & 64) {
            if (acc[i][j][0] == 1.2345e-30f) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, acc[i][j]), rC, (int)off, 0, 0);
          } else {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, acc[i][j]), rC, (int)off, 0, 0);
          }
        }
      continue;
    }
#undef QAGNN_NN2_FRAG
#undef QAGNN_NN2_FRAG_WAIT
#undef QAGNN_NN2_MFMA_TILE3
#undef QAGNN_NN2_BAR
#undef QAGNN_NN2_BAR_VM
#undef QAGNN_NN2_BAR_VMN
#undef QAGNN_NN2_FIRST_LOADS
#undef QAGNN_NN2_GLOAD
#undef QAGNN_NN2_SET_TILE
#undef QAGNN_NN2_GLDS
#undef QAGNN_NN2_STORE_B
#undef QAGNN_NN2_SPLIT_A
#undef QAGNN_NN2_MFMA_TILE

    // ---- epilogue: transpose 16 rows at a time through the wave's LDS slab, then whole-row 16-byte stores
    float* const St = reinterpret_cast<float*>(smem) + w * SLAB_ROWS * PS;
    float (*const cstat)[2][BN] = reinterpret_cast<float (*)[2][BN]>(smem + SLAB_B);
    constexpr int ROW_F4 = BN / 4, TILE_F4 = SLAB_ROWS * ROW_F4, ST_IT = (TILE_F4 + 63) / 64;
    constexpr int SC = (BN + 63) / 64;
    float x0r[SC], s1[SC], s2v[SC];
#pragma unroll
    for (int cc = 0; cc < SC; ++cc) x0r[cc] = s1[cc] = s2v[cc] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      __syncthreads();  // k-loop reads (first pass) / the previous pass's slab reads are done before the slab is overwritten
      const int lr0 = (lane >> 4) * 4;
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) St[(lr0 + r) * PS + j * 16 + (lane & 15)] = acc[i][j][r];
      __syncthreads();
      if constexpr (STATS) {
        // row 0 of the first slab of this wave's 128-row statistics tile (wave 0's, or wave 4's in a 256-row block) = the tile's first row
        const float* const S0 = reinterpret_cast<const float*>(smem) + (w & ~3) * SLAB_ROWS * PS;
#pragma unroll
        for (int cc = 0; cc < SC; ++cc) {
          const int c = lane + cc * 64;
          if (c < BN && n0 + c < No) {
            const float bcv = a.bias ? a.bias[n0 + c] : 0.f;
            if (i == 0) x0r[cc] = S0[c] + bcv;
            const int rows = M - (m0 + (w * 2 + i) * 16);
            float v[SLAB_ROWS];
#pragma unroll
            for (int r = 0; r < SLAB_ROWS; ++r) v[r] = St[r * PS + c];
#pragma unroll
            for (int r = 0; r < SLAB_ROWS; ++r) {
              const float dlt = r < rows ? (v[r] + bcv) - x0r[cc] : 0.f;
              s1[cc] += dlt;
              s2v[cc] = fmaf(dlt, dlt, s2v[cc]);
            }
          }
        }
      }
#pragma unroll
      for (int itx = 0; itx < ST_IT; ++itx) {
        const int idx = lane + itx * 64;
        if (idx >= TILE_F4) break;
        const int slr = idx / ROW_F4, c4 = idx % ROW_F4;
        const int row = m0 + (w * 2 + i) * 16 + slr, col = n0 + c4 * 4;
        if (row >= M || col >= No) continue;
        float4 v = ld4(St + slr * PS + c4 * 4);
        if (a.bias) v = add4(v, ld4(a.bias + col));
        if (a.rowtab) v = add4(v, ld4(a.rowtab + (int64_t)a.rowidx[row] * a.ldt + col));
        float* dst = a.C + (int64_t)row * a.ldc + col;
        if (a.accumulate) v = add4(v, ld4(dst));
        if constexpr (QAGNN_NN2_ABL & 64) {
          if (v.x == 1.2345e-30f) st4(dst, v);
        } else {
          st4(dst, v);
        }
      }
    }
    if constexpr (STATS) {
#pragma unroll
      for (int cc = 0; cc < SC; ++cc) {
        const int c = lane + cc * 64;
        if (c < BN) { cstat[w][0][c] = s1[cc]; cstat[w][1][c] = s2v[cc]; }
      }
      __syncthreads();
      // one partial per 128 rows (the contract of colstat_part): thread (hw, lane) of half `hf` = column hw * 64 + lane of that half,
      // its own x0r[hw] is that column's shift
      const int hw = w & 3, hf = w >> 2, col = hw * 64 + lane;
      if (col < BN && n0 + col < No && m0 + hf * 128 < M) {
        float x0c = x0r[0];
#pragma unroll
        for (int cc = 1; cc < SC; ++cc) x0c = (hw == cc) ? x0r[cc] : x0c;
        float* const pt = a.colstat_part + ((int64_t)(tile / ncb) * (WV / 4) + hf) * 3 * No + n0 + col;
        const int w0 = hf * 4;
        pt[0] = x0c;
        pt[No] = (cstat[w0][0][col] + cstat[w0 + 1][0][col]) + (cstat[w0 + 2][0][col] + cstat[w0 + 3][0][col]);
        pt[2 * No] = (cstat[w0][1][col] + cstat[w0 + 1][1][col]) + (cstat[w0 + 2][1][col] + cstat[w0 + 3][1][col]);
      }
    }
  }
}
#undef QAGNN_NN2_SIX
#undef QAGNN_NN2_SIX_T

static int num_cus() {
  static int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    return v;
  }();
  return n;
}

template <int NT, bool AFFINE, bool STATS, int NP, bool PACKED, int WV>
static int launch_i(const qagnn_gemm_nn_args& b, const float* B1n, int ldn1, const float* B2n, int ldn2, int grid, int ntiles, hipStream_t stream) {
  const size_t lds = nn2_lds_bytes<NT, STATS, WV, NP>(AFFINE ? 2 * ((b.K1 + 31) & ~31) * 4 : 0);
  constexpr int lds_max = nn2_lds_bytes<NT, STATS, WV, NP>(AFFINE ? 2 * 256 * 4 : 0);  // (nn2_ok: K1 <= 256 with a scale / shift)
  static bool raised[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (lds_max > 64 * 1024 && !raised[dev & 63]) {
    hipError_t e = hipFuncSetAttribute((const void*)k_gemm_nn2<NT, AFFINE, STATS, NP, PACKED, WV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    if (e != hipSuccess) { set_error("gemm_nn2: cannot raise the dynamic LDS limit: %s", hipGetErrorString(e)); return QAGNN_EHIP; }
    raised[dev & 63] = true;
  }
  k_gemm_nn2<NT, AFFINE, STATS, NP, PACKED, WV><<<grid, WV * 64, lds, stream>>>(b, B1n, ldn1, B2n, ldn2, ntiles);
  QAGNN_LAUNCH_CHECK("k_gemm_nn2");
  return QAGNN_OK;
}

// PACKED: B1n = the packed buffer, ldn1 = its column tiles per k-tile.  WV = 8: the staggered 256-row block, one per CU.
template <int NT, int NP, bool PACKED = false, int WV = 4>
static int launch_nt(const qagnn_gemm_nn_args& a, const float* B1n, int ldn1, const float* B2n, int ldn2, hipStream_t stream) {
  qagnn_gemm_nn_args b = a;
  b.xcd_remap = 1;
  const int ntiles = cdiv(a.No, NT * 16) * cdiv(a.M, WV * 32);
  const int cap = (num_cus() * (WV == 8 ? 1 : 2)) & ~7;
  const int grid = ntiles < cap ? ntiles : cap;
  if constexpr (NT == 13 || NT == 7 || NT == 4 || NT == 2) {
    if (a.colstat_part) return launch_i<NT, false, true, NP, PACKED, WV>(b, B1n, ldn1, B2n, ldn2, grid, ntiles, stream);
  }
  if (a.a_scale) return launch_i<NT, true, false, NP, PACKED, WV>(b, B1n, ldn1, B2n, ldn2, grid, ntiles, stream);
  return launch_i<NT, false, false, NP, PACKED, WV>(b, B1n, ldn1, B2n, ldn2, grid, ntiles, stream);
}

// B [No][K1] | [No][K2] (k contiguous) -> the packed image of PACKED kernels: block ((it * NJ + j) * 3 + p) of 1 KB holds, for lane
// l = (x = l & 15, c = l >> 4), at slot (x ^ 2c) + 16c, the piece-p halves of B[k = position(it, 8 c .. 8 c + 7)][n = 16 j + x]; `it` walks the k-tiles in the
// kernel's order (the straddling tile first).  One wave per block of NP pieces; zeros past K and past No.
// NP == 2: blocks of 2 KB (hi | lo fp16 pieces of B s_j), s_j = the power of two that puts the largest magnitude of column tile j -- over
// ALL of its k, both segments -- into [2^14, 2^15); every wave of a column tile derives it again from the weights (they are L2-resident and
// tiny), the wave of k-tile 0 leaves its exponent field at word j behind the image and its 13 tiles of slack (k_gemm_nn2's epilogue).
template <int NP>
__device__ __forceinline__ void pack_b_wave(const float* __restrict__ B1n, int ldn1, int K1, const float* __restrict__ B2n, int ldn2, int K2, int No,
                                            int NJ, int nkt, int it, int j, unsigned char* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  if (j >= NJ) return;
  float sb = 1.f;
  if constexpr (NP <= 2) {
    const int nn = j * 16 + (lane & 15);
    float m = 0.f;
    if (nn < No) {
      for (int k = (lane >> 4) * 4; k < K1; k += 16) {
        const float4 v = ld4(B1n + (int64_t)nn * ldn1 + k);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      }
      for (int k = (lane >> 4) * 4; k < K2; k += 16) {
        const float4 v = ld4(B2n + (int64_t)nn * ldn2 + k);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      }
    }
    // (fmaxf drops a NaN operand: a NaN weight is not seen by the scale, and reaches the output through its own products)
    const uint32_t fb = h2_scale_field(__builtin_bit_cast(uint32_t, wave_max(m)));
    sb = h2_field_to_scale(fb);
    if (it == 0 && lane == 0) reinterpret_cast<uint32_t*>(out + ((int64_t)nkt * NJ + 13) * (NP * 1024))[j] = fb;
  }
  const int r1 = K2 > 0 ? (K1 & 31) : 0;
  const int mixi = r1 != 0 ? 1 : 0;
  const int n1 = mixi ? (K1 >> 5) : ((K1 + 31) >> 5);
  const int s2 = mixi ? 32 - r1 : 0;
  const int kk = (lane >> 4) * 8, n = j * 16 + (lane & 15);
  const float* src = nullptr;  // the 8 consecutive k of this lane, or nullptr = zeros
  if (n < No) {
    if (mixi && it == 0) {
      if (kk < r1) src = B1n + (int64_t)n * ldn1 + (K1 - r1 + kk);
      else if (kk - r1 < K2) src = B2n + (int64_t)n * ldn2 + (kk - r1);
    } else {
      const int u = it - mixi;
      if (u < n1) {
        if (u * 32 + kk < K1) src = B1n + (int64_t)n * ldn1 + (u * 32 + kk);
      } else {
        const int k2 = s2 + (u - n1) * 32 + kk;
        if (k2 < K2) src = B2n + (int64_t)n * ldn2 + k2;
      }
    }
  }
  u32x4s r[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
  if (src) {
    r[0] = *reinterpret_cast<const u32x4s*>(src);
    r[1] = *reinterpret_cast<const u32x4s*>(src + 4);
  }
  u32x4s f[NP];
  if constexpr (NP <= 2) split_frag2<false, NP>(r, f, nullptr, nullptr, 0.f, sb);
  else split_frag<false>(r, f, nullptr, nullptr, 0.f);
  // (the slot permutation of the in-kernel loader, so that both kinds of image are read with the same fragment offsets)
  unsigned char* dst = out + ((int64_t)it * NJ + j) * (NP * 1024) + (((lane & 15) ^ (2 * (lane >> 4))) + 16 * (lane >> 4)) * 16;
#pragma unroll
  for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4s*>(dst + p * 1024) = f[p];
}

template <int NP>
__global__ __launch_bounds__(256) void k_pack_b(const float* __restrict__ B1n, int ldn1, int K1, const float* __restrict__ B2n, int ldn2, int K2,
                                               int No, int NJ, int nkt, unsigned char* __restrict__ out) {
  pack_b_wave<NP>(B1n, ldn1, K1, B2n, ldn2, K2, No, NJ, nkt, blockIdx.y, blockIdx.x * 4 + (threadIdx.x >> 6), out);
}

// The same for up to PACK_MAX weights in ONE launch (qagnn_gemm_nn_prepack_f32: all B operands of a training step's large NN products,
// packed right behind the operand-packing gather; 39 pack launches per step of the 320-subgraph batch become one).  The descriptors
// travel as kernel arguments (3.4 KB): nothing to upload, and a captured graph replays them as they are.
constexpr int PACK_MAX = 60;
struct PackOne { const float* B1n; const float* B2n; long long out_off; int ldn1, K1, ldn2, K2, No, NJ, jb, blk0, np, nkt; };  // jb = ceil(NJ / 4)
struct PackArgs { int n; int nblk; PackOne d[PACK_MAX]; };
__global__ __launch_bounds__(256) void k_pack_b_multi(PackArgs a, unsigned char* __restrict__ out) {
  int i = 0;
  while (i + 1 < a.n && (int)blockIdx.x >= a.d[i + 1].blk0) ++i;  // wave-uniform walk over <= 60 entries
  const PackOne& d = a.d[i];
  const int lb = (int)blockIdx.x - d.blk0;
  const int it = lb / d.jb, j = (lb % d.jb) * 4 + (threadIdx.x >> 6);
  if (d.np == 2) pack_b_wave<2>(d.B1n, d.ldn1, d.K1, d.B2n, d.ldn2, d.K2, d.No, d.NJ, d.nkt, it, j, out + d.out_off);
  else if (d.np == 1) pack_b_wave<1>(d.B1n, d.ldn1, d.K1, d.B2n, d.ldn2, d.K2, d.No, d.NJ, d.nkt, it, j, out + d.out_off);
  else pack_b_wave<3>(d.B1n, d.ldn1, d.K1, d.B2n, d.ldn2, d.K2, d.No, d.NJ, d.nkt, it, j, out + d.out_off);
}

static int walk_tiles(int K1, int K2) {
  const int r1 = K2 > 0 ? (K1 & 31) : 0;
  const int mixi = r1 != 0 ? 1 : 0;
  const int n1 = mixi ? (K1 >> 5) : ((K1 + 31) >> 5);
  const int s2 = mixi ? 32 - r1 : 0;
  const int n2 = K2 > s2 ? ((K2 - s2 + 31) >> 5) : 0;
  return mixi + n1 + n2;
}

}  // namespace nn2

// Which form a product takes (the A/B runs of the forms against each other: profiles/r4_run16_nn2_stagger.txt, r4_run28_round4_switches_ab.txt):
// B packed once per product (k_pack_b) wherever the caller hands over a workspace and the product has at least NN2_PACK_MIN_M rows, the
// in-kernel split otherwise; the staggered 8-wave block wherever a packed product has at least one 256-row tile per CU.
constexpr int NN2_PACK_MIN_M = 8192;
bool nn2_packed_ok(const qagnn_gemm_nn_args& a, int64_t ws_bytes, int np) { return a.M >= NN2_PACK_MIN_M && ws_bytes >= nn2_pack_bytes(a.No, a.K1, a.K2, np); }
// Measured at M = 64 000 (tools/nn2_ablate.hip, profiles/r4_run16_nn2_stagger.txt), 4-wave blocks -> staggered block:
// [208|112] -> 624 141 -> 122 us, 624 -> 208 96..101 -> 79, but 208 -> 208 38 -> 39 and 624 -> 112 (NT = 7) 53 -> 54: with one block per
// CU nothing runs under a tile's first loads, and the last tiles' stores are a tail at HBM speed, which 10 k-tiles of 13 column tiles
// amortise and 7 k-tiles or 7 column tiles do not.
static bool nn2_staggered(int nt, const qagnn_gemm_nn_args& a) {
  return nt >= 8 && nn2::walk_tiles(a.K1, a.K2) >= 10 && (int64_t)cdiv(a.No, nt * 16) * cdiv(a.M, 256) * 10 >= nn2::num_cus() * 9;
}
// the PACKED kernel on an image `p` of B (NJ column tiles per k-tile)
static int launch_nn2_image(int nt, const qagnn_gemm_nn_args& a, const float* p, int NJ, hipStream_t stream, int np = 3) {
  if (np == 1) {  // the one-MFMA reduced-precision form (qagnn_gemm_nn_args.pieces = 1)
    switch (nt) {
      case 13: return nn2::launch_nt<13, 1, true>(a, p, NJ, nullptr, 0, stream);
      case 8: return nn2::launch_nt<8, 1, true>(a, p, NJ, nullptr, 0, stream);
      case 7: return nn2::launch_nt<7, 1, true>(a, p, NJ, nullptr, 0, stream);
      case 4: return nn2::launch_nt<4, 1, true>(a, p, NJ, nullptr, 0, stream);
      default: return nn2::launch_nt<2, 1, true>(a, p, NJ, nullptr, 0, stream);
    }
  }
  if (np == 2) {  // the two-piece fp16 form: 4-wave blocks at every shape (tools/nn2_ablate.hip, profiles/r6_run1_three_product_ablation.txt)
    switch (nt) {
      case 13: return nn2::launch_nt<13, 2, true>(a, p, NJ, nullptr, 0, stream);
      case 8: return nn2::launch_nt<8, 2, true>(a, p, NJ, nullptr, 0, stream);
      case 7: return nn2::launch_nt<7, 2, true>(a, p, NJ, nullptr, 0, stream);
      case 4: return nn2::launch_nt<4, 2, true>(a, p, NJ, nullptr, 0, stream);
      default: return nn2::launch_nt<2, 2, true>(a, p, NJ, nullptr, 0, stream);
    }
  }
  if (nn2_staggered(nt, a)) {
    if (nt == 13) return nn2::launch_nt<13, 3, true, 8>(a, p, NJ, nullptr, 0, stream);
    return nn2::launch_nt<8, 3, true, 8>(a, p, NJ, nullptr, 0, stream);
  }
  switch (nt) {
    case 13: return nn2::launch_nt<13, 3, true>(a, p, NJ, nullptr, 0, stream);
    case 8: return nn2::launch_nt<8, 3, true>(a, p, NJ, nullptr, 0, stream);
    case 7: return nn2::launch_nt<7, 3, true>(a, p, NJ, nullptr, 0, stream);
    case 4: return nn2::launch_nt<4, 3, true>(a, p, NJ, nullptr, 0, stream);
    default: return nn2::launch_nt<2, 3, true>(a, p, NJ, nullptr, 0, stream);
  }
}

// what the second-generation kernel takes: no fused row gather, 32-bit operand offsets, segments that are multiples of 8
bool nn2_ok(const qagnn_gemm_nn_args& a, int ldn1, int ldn2) {
  const int64_t lim = (int64_t)0x7FFFFFFF;
  // (a gathered A1: the table's extent must be known and addressable with the kernels' 32-bit offsets; one segment)
  if (a.a_rowidx && !(a.a_rows > 0 && a.a_rows * (int64_t)a.lda1 * 4 < lim && a.K2 == 0)) return false;
  if (a.K1 % 8 != 0 || a.K2 % 8 != 0) return false;
  if ((int64_t)a.M * a.lda1 * 4 >= lim || (int64_t)a.No * ldn1 * 4 >= lim || (int64_t)a.M * a.ldc * 4 >= lim) return false;
  if (a.K2 > 0 && ((int64_t)a.M * a.lda2 * 4 >= lim || (int64_t)a.No * ldn2 * 4 >= lim)) return false;
  if (a.K2 > 0 && (a.K1 & 31) != 0 && a.K2 < 32 - (a.K1 & 31)) return false;  // (the straddling tile must lie inside segment 2)
  if (a.a_scale && a.K1 > 256) return false;  // (the scale / shift vectors live in LDS next to the two B images)
  return true;
}

int launch_nn2(int nt, const qagnn_gemm_nn_args& a, const float* B1n, int ldn1, const float* B2n, int ldn2, hipStream_t stream) {
  switch (nt) {
    case 13: return nn2::launch_nt<13, 3>(a, B1n, ldn1, B2n, ldn2, stream);
    case 8: return nn2::launch_nt<8, 3>(a, B1n, ldn1, B2n, ldn2, stream);
    case 7: return nn2::launch_nt<7, 3>(a, B1n, ldn1, B2n, ldn2, stream);
    case 4: return nn2::launch_nt<4, 3>(a, B1n, ldn1, B2n, ldn2, stream);
    default: return nn2::launch_nt<2, 3>(a, B1n, ldn1, B2n, ldn2, stream);
  }
}

// bytes of the packed image of B for one product (13 column tiles of slack: the last column block of a k-tile reads its full width);
// np == 2: + one exponent word per column tile (13 more of slack for the same reason), rounded up to 16 bytes
int64_t nn2_pack_bytes(int No, int K1, int K2, int np) {
  const int64_t img = ((int64_t)nn2::walk_tiles(K1, K2) * cdiv(No, 16) + 13) * (np * 1024);
  return np <= 2 ? img + (((int64_t)cdiv(No, 16) + 13) * 4 + 15) / 16 * 16 : img;
}
// (h2_ok: what the two-piece form additionally asks of a product -- a known maximum of A and a finite-size image)
bool nn2_h2_ok(const qagnn_gemm_nn_args& a) { return a.a_amax1 != nullptr && (a.K2 == 0 || a.a_amax2 != nullptr) && a.M >= NN2_PACK_MIN_M; }

// pack B into `ws` (>= nn2_pack_bytes), then the PACKED kernel: two launches
int launch_nn2_packed(int nt, const qagnn_gemm_nn_args& a, const float* B1n, int ldn1, const float* B2n, int ldn2, void* ws, hipStream_t stream, int np) {
  const int NJ = cdiv(a.No, 16), nkt = nn2::walk_tiles(a.K1, a.K2);
  if (np == 2) nn2::k_pack_b<2><<<dim3(cdiv(NJ, 4), nkt), 256, 0, stream>>>(B1n, ldn1, a.K1, B2n, ldn2, a.K2, a.No, NJ, nkt, reinterpret_cast<unsigned char*>(ws));
  else if (np == 1) nn2::k_pack_b<1><<<dim3(cdiv(NJ, 4), nkt), 256, 0, stream>>>(B1n, ldn1, a.K1, B2n, ldn2, a.K2, a.No, NJ, nkt, reinterpret_cast<unsigned char*>(ws));
  else nn2::k_pack_b<3><<<dim3(cdiv(NJ, 4), nkt), 256, 0, stream>>>(B1n, ldn1, a.K1, B2n, ldn2, a.K2, a.No, NJ, nkt, reinterpret_cast<unsigned char*>(ws));
  QAGNN_LAUNCH_CHECK("k_pack_b");
  return launch_nn2_image(nt, a, reinterpret_cast<const float*>(ws), NJ, stream, np);
}

// ------------------------------------------------------------------------------------------------------------
// Registry of pre-packed B images (qagnn_gemm_nn_prepack_f32): (B1n, B2n, K1, K2, No, pitches) -> packed image.  The caller keeps the
// fp32 weights and the packed buffer alive -- and unchanged -- until it clears or replaces the tag; qagnn_gemm_nn_split*_f32 look a
// product's B operand up here first.  Host-side only (a handful of entries, one mutex).
// ------------------------------------------------------------------------------------------------------------
}  // namespace qagnn
#include <mutex>
#include <vector>
namespace qagnn {
struct PrepackEntry { long long tag; const float* B1n; const float* B2n; int ldn1, K1, ldn2, K2, No, np; const void* pk; };
static std::mutex g_prepack_mu;
static std::vector<PrepackEntry> g_prepack;

const void* nn2_prepack_lookup(const float* B1n, int ldn1, int K1, const float* B2n, int ldn2, int K2, int No, int np) {
  std::lock_guard<std::mutex> lk(g_prepack_mu);
  // newest registration first: should an owner ever leave stale entries behind, a live one for the same address wins
  for (size_t i = g_prepack.size(); i-- > 0;) {
    const PrepackEntry& e = g_prepack[i];
    if (e.np == np && e.B1n == B1n && e.K1 == K1 && e.No == No && e.ldn1 == ldn1 && e.K2 == K2 && (K2 == 0 || (e.B2n == B2n && e.ldn2 == ldn2))) return e.pk;
  }
  return nullptr;
}

static bool prepack_takes(const qagnn_pack_desc& d) {
  if (!d.B1n || d.No <= 0 || d.K1 < 8 || d.K1 % 8 != 0 || d.K2 < 0 || d.K2 % 8 != 0 || d.ldn1 < d.K1 || d.ldn1 % 4 != 0 || !aligned16(d.B1n)) return false;
  if (d.K2 > 0 && (!d.B2n || d.ldn2 < d.K2 || d.ldn2 % 4 != 0 || !aligned16(d.B2n))) return false;
  if (d.K2 > 0 && (d.K1 & 31) != 0 && d.K2 < 32 - (d.K1 & 31)) return false;  // (nn2_ok: the straddling tile must lie inside segment 2)
  return true;
}

int launch_nn2_prepacked(int nt, const qagnn_gemm_nn_args& a, const void* pk, hipStream_t stream, int np) {
  return launch_nn2_image(nt, a, reinterpret_cast<const float*>(pk), cdiv(a.No, 16), stream, np);
}
}  // namespace qagnn

using namespace qagnn;

extern "C" int64_t qagnn_gemm_nn_prepack_bytes(const qagnn_pack_desc* d, int32_t n) {
  int64_t tot = 0;
  for (int i = 0; i < n; ++i)
    if (prepack_takes(d[i])) tot += nn2_pack_bytes(d[i].No, d[i].K1, d[i].K2, (d[i].pieces == 2 || d[i].pieces == 1) ? d[i].pieces : 3);
  return tot;
}

extern "C" int qagnn_gemm_nn_prepack_clear(int64_t tag) {
  std::lock_guard<std::mutex> lk(g_prepack_mu);
  size_t keep = 0;  // (stable: the lookup walks the registry newest first)
  for (size_t i = 0; i < g_prepack.size(); ++i)
    if (!(tag == 0 || g_prepack[i].tag == tag)) g_prepack[keep++] = g_prepack[i];
  g_prepack.resize(keep);
  return QAGNN_OK;
}

extern "C" int qagnn_gemm_nn_prepack_f32(const qagnn_pack_desc* d, int32_t n, void* out, int64_t out_bytes, int64_t tag, qagnn_stream_t stream_) {
  QAGNN_REQUIRE(d && n > 0 && out && tag != 0 && aligned16(out), QAGNN_EINVAL, "gemm_nn_prepack: bad arguments");
  QAGNN_REQUIRE(out_bytes >= qagnn_gemm_nn_prepack_bytes(d, n), QAGNN_EINVAL, "gemm_nn_prepack: the buffer of %lld bytes is too small", (long long)out_bytes);
  qagnn_gemm_nn_prepack_clear(tag);
  std::vector<PrepackEntry> fresh;
  int64_t off = 0;
  for (int i0 = 0; i0 < n;) {  // one launch per PACK_MAX weights
    nn2::PackArgs pa;
    pa.n = 0;
    pa.nblk = 0;
    int i = i0;
    for (; i < n && pa.n < nn2::PACK_MAX; ++i) {
      if (!prepack_takes(d[i])) continue;
      nn2::PackOne& o = pa.d[pa.n++];
      o.B1n = d[i].B1n; o.B2n = d[i].K2 > 0 ? d[i].B2n : nullptr; o.ldn1 = d[i].ldn1; o.K1 = d[i].K1; o.ldn2 = d[i].ldn2; o.K2 = d[i].K2; o.No = d[i].No;
      o.NJ = cdiv(d[i].No, 16);
      o.jb = cdiv(o.NJ, 4);
      o.blk0 = pa.nblk;
      o.out_off = off;
      o.np = (d[i].pieces == 2 || d[i].pieces == 1) ? d[i].pieces : 3;
      o.nkt = nn2::walk_tiles(o.K1, o.K2);
      pa.nblk += o.jb * o.nkt;
      fresh.push_back(PrepackEntry{(long long)tag, o.B1n, o.B2n, o.ldn1, o.K1, o.ldn2, o.K2, o.No, o.np, static_cast<const unsigned char*>(out) + off});
      off += nn2_pack_bytes(o.No, o.K1, o.K2, o.np);
    }
    if (pa.n > 0) {
      nn2::k_pack_b_multi<<<pa.nblk, 256, 0, (hipStream_t)stream_>>>(pa, static_cast<unsigned char*>(out));
      QAGNN_LAUNCH_CHECK("k_pack_b_multi");
    }
    i0 = i;
  }
  std::lock_guard<std::mutex> lk(g_prepack_mu);
  g_prepack.insert(g_prepack.end(), fresh.begin(), fresh.end());
  return QAGNN_OK;
}
