// 256x256x128 OCP-e4m3 MFMA GEMM for gfx950 (BASELINE configs[4]: fp8 encoder / decoder GEMMs of the L1 SAE).
//
//   C[256*bm .., 256*bn ..] = sum_k A(m,k) * B(n,k)          both operands K-contiguous bytes (fp8 e4m3fn), fp32 accumulate
//
// Same skeleton as gemm256.h (8 waves = 2 (M) x 4 (N), each a 128x64 output = 4x2 tiles of 32x32, LDS-DMA staging into
// two 64 KiB stages, K loop rotated around its barrier, epilogue through the row-major functors), with the K tile kept
// at 128 BYTES per row, i.e. 128 fp8 elements: one tile is two v_mfma_f32_32x32x64_f8f6f4 steps (64 cycles each, twice
// the cycles of the bf16 32x32x16 at four times the K: 2x the bf16 rate) instead of four bf16 steps, so staging bytes,
// LDS reads and barriers per MFMA CYCLE are those of the bf16 kernel while the FLOPs double.
//   Fragment of the 32x32x64 fp8 MFMA: lane (row = lane & 31, half h = lane >> 5) holds the 32 bytes k = 32 h .. 32 h + 31
//   of its row -- two ds_read_b128 of the chunks 4 s + 2 h, 4 s + 2 h + 1 of the XOR-swizzled [256][128 B] image (the
//   swizzle of gemm.h is conflict-free for any logical chunk: the 16 lanes of a ds_read_b128 group differ in (row & 1,
//   (row >> 1) & 7)).  A and B use the same lane -> k map, so the hardware's internal k order does not matter.
//   Registers: the A fragments are single-buffered (A fragment i of the next step is requested right after its last
//   MFMA of this step issued: 8 MFMAs = 512 cycles of lead), the B fragments double-buffered: 64 fragment registers
//   next to 128 accumulators keep two waves per SIMD.
// The unscaled instruction is reached through the scaled builtin with zero scales (hipcc selects v_mfma_f32_32x32x64_f8f6f4).
#pragma once
#include "gemm256.h"
#include "gemm256s.h"

typedef __attribute__((ext_vector_type(8))) int i32x8;

#ifndef G8_SPREAD
#define G8_SPREAD 1          // tools/build_variant.sh A/B switch: 0 = the K loops' side work clustered in front of each 8-MFMA group (rounds 2-5)
#endif

struct Gemm8Args {
  const unsigned char* A;   // [rows_a][lda] fp8, K contiguous
  const unsigned char* B;   // [rows_b][ldb] fp8, K contiguous
  int64_t lda, ldb;         // bytes (= elements)
  int nbm, nbn;             // 256-wide output tiles
  int ktiles;               // 128-element K tiles
};

// source byte offset of the 16 B that lane `lane` of DMA piece p (0..31) moves (piece = rows 8p..8p+7 of the image)
__device__ __forceinline__ unsigned g8_src_off(int p, int lane, int64_t ld) {
  const int r = 8 * p + (lane >> 3), pc = lane & 7;
  return (unsigned)(r * ld + ((pc ^ ((r >> 1) & 7)) << 4));
}

// 32-byte fragment: rows base32 .. base32+31, K step s (0, 1) of the 128-byte tile
__device__ __forceinline__ i32x8 g8_frag(const char* img, int base32, int s, int lane) {
  const int r = base32 + (lane & 31), c0 = 4 * s + 2 * (lane >> 5), sw = (r >> 1) & 7;
  const u32x4 lo = *reinterpret_cast<const u32x4*>(img + r * 128 + ((c0 ^ sw) << 4));
  const u32x4 hi = *reinterpret_cast<const u32x4*>(img + r * 128 + (((c0 + 1) ^ sw) << 4));
  i32x8 v;
  v[0] = (int)lo[0]; v[1] = (int)lo[1]; v[2] = (int)lo[2]; v[3] = (int)lo[3];
  v[4] = (int)hi[0]; v[5] = (int)hi[1]; v[6] = (int)hi[2]; v[7] = (int)hi[3];
  return v;
}

template <class Epi>
__global__ __launch_bounds__(512, 2) void gemm256_fp8_kernel(Gemm8Args g, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 2, wn = w & 3;
  const int nblk = g.nbm * g.nbn;
  const int id = xcd_remap(blockIdx.x, nblk);
  int bm, bn;
  tile_coords(id, g.nbm, g.nbn, bm, bn);

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const unsigned char* a_base = g.A + (int64_t)(bm * G2_BM) * g.lda;
  const unsigned char* b_base = g.B + (int64_t)(bn * G2_BN) * g.ldb;
  unsigned voff_a[4], voff_b[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    voff_a[q] = g8_src_off(4 * w + q, lane, g.lda);
    voff_b[q] = g8_src_off(4 * w + q, lane, g.ldb);
  }
  typedef __attribute__((address_space(3))) char* lptr_t;
  const unsigned smem_base = (unsigned)(uintptr_t)(lptr_t)smem;
  const unsigned piece0 = (unsigned)__builtin_amdgcn_readfirstlane(4 * w * 1024);
  auto issue = [&](int kt, int stage, int q) {
    const unsigned dst = smem_base + stage * G2_STAGE_BYTES + piece0 + q * 1024;
    glds16_x2(a_base + (int64_t)kt * 128, b_base + (int64_t)kt * 128, voff_a[q], voff_b[q], dst, dst + G2_OPER_BYTES);
  };
  // A3 (Epi::DEEP_A_RING, the decoder: K = n_dict, the fp8 latent streams from HBM): three A slots at LDS 0, two B slots
  // behind them; A tile kt + 2 is requested at the START of tile kt, B tile kt + 2 after hand-over kt -- gemm256.h.
  constexpr bool A3 = G2_A3 && epi_deep_a_ring<Epi>::value;
  auto issue_aa = [&](int kt, int slot, int qp) {
    const unsigned dst = smem_base + slot * G2_OPER_BYTES + piece0 + 2 * qp * 1024;
    const unsigned char* src = a_base + (int64_t)kt * 128;
    glds16_x2(src, src, voff_a[2 * qp], voff_a[2 * qp + 1], dst, dst + 1024);
  };
  auto issue_bb = [&](int kt, int slot, int qp) {
    const unsigned dst = smem_base + (3 + slot) * G2_OPER_BYTES + piece0 + 2 * qp * 1024;
    const unsigned char* src = b_base + (int64_t)kt * 128;
    glds16_x2(src, src, voff_b[2 * qp], voff_b[2 * qp + 1], dst, dst + 1024);
  };

  const int kt_last = g.ktiles - 1;
  auto clampk = [&](int kt) { return kt < kt_last ? kt : kt_last; };   // past-the-end tiles re-copy the last one (harmless)
  if constexpr (A3) {
    issue_aa(0, 0, 0); issue_aa(0, 0, 1);
    issue_bb(0, 0, 0); issue_bb(0, 0, 1);
    issue_bb(clampk(1), 1, 0); issue_bb(clampk(1), 1, 1);
    issue_aa(clampk(1), 1, 0); issue_aa(clampk(1), 1, 1);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) issue(0, 0, q);
#pragma unroll
    for (int q = 0; q < 4; ++q) issue(clampk(1), 1, q);
  }
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // tile 0 only (8 older DMA instructions); tile 1 is waited for at the first hand-over
  __syncthreads();

  i32x8 fa[4], fb[2][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) fa[i] = g8_frag(smem, 128 * wm + 32 * i, 0, lane);
#pragma unroll
  for (int j = 0; j < 2; ++j) fb[0][j] = g8_frag(smem + (A3 ? 3 : 1) * G2_OPER_BYTES, 64 * wn + 32 * j, 0, lane);

  int cur = 0, aslot = 0;
#if G8_SPREAD
  // source origins of K tiles min(kt + 1, last) and min(kt + 2, last), carried (an add per K tile instead of a clamp + multiply per DMA call)
  const unsigned char* r1a = a_base + (int64_t)clampk(1) * 128;
  const unsigned char* r1b = b_base + (int64_t)clampk(1) * 128;
  const unsigned char* r2a = a_base + (int64_t)clampk(2) * 128;
  const unsigned char* r2b = b_base + (int64_t)clampk(2) * 128;
  auto issue_p = [&](const unsigned char* pa, const unsigned char* pb, int stage, int q) {
    const unsigned dst = smem_base + stage * G2_STAGE_BYTES + piece0 + q * 1024;
    glds16_x2(pa, pb, voff_a[q], voff_b[q], dst, dst + G2_OPER_BYTES);
  };
  auto issue_aa_p = [&](const unsigned char* src, int slot, int qp) {
    const unsigned dst = smem_base + slot * G2_OPER_BYTES + piece0 + 2 * qp * 1024;
    glds16_x2(src, src, voff_a[2 * qp], voff_a[2 * qp + 1], dst, dst + 1024);
  };
  auto issue_bb_p = [&](const unsigned char* src, int slot, int qp) {
    const unsigned dst = smem_base + (3 + slot) * G2_OPER_BYTES + piece0 + 2 * qp * 1024;
    glds16_x2(src, src, voff_b[2 * qp], voff_b[2 * qp + 1], dst, dst + 1024);
  };
#endif
  for (int kt = 0; kt < g.ktiles; ++kt) {
    const int anext = aslot == 2 ? 0 : aslot + 1, aprev = anext == 2 ? 0 : anext + 1;
    const char* sa = A3 ? smem + aslot * G2_OPER_BYTES : smem + cur * G2_STAGE_BYTES;
    const char* sb = A3 ? smem + (3 + cur) * G2_OPER_BYTES : sa + G2_OPER_BYTES;
    const char* na = A3 ? smem + anext * G2_OPER_BYTES : smem + (cur ^ 1) * G2_STAGE_BYTES;
    const char* nb = A3 ? smem + (3 + (cur ^ 1)) * G2_OPER_BYTES : na + G2_OPER_BYTES;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      // where the NEXT step's fragments come from: step 1 of this tile, or step 0 of the next tile (other stage)
      const char* xa = s == 0 ? sa : na;
      const char* xb = s == 0 ? sb : nb;
      const int xs = s == 0 ? 1 : 0;
#if G8_SPREAD
      // Side work one piece per MFMA gap (round 6: clustered in front of the group -- two DMA calls with their address arithmetic, the hand-over
      // and two fragment reads, ~45 instructions -- it sat in BOTH waves of a SIMD at the same moment, right behind the barrier they share, and a
      // 64-cycle MFMA covers 16 issue slots, not 45).  MFMA m = 2 i + j; behind it: m odd -> A fragment i of the next step (its registers are
      // free); m = 4 / 6 -> the next step's B fragments (the other buffer); step 0: m = 0 / 2 -> the DMA calls of tile kt + 1's second half
      // (A3: A tile kt + 2); step 1: m = 0 -> the hand-over (every read of this stage was issued in step 0; the first read of the next stage
      // follows at m = 1), m = 2 / 4 -> tile kt + 2's first half (A3: B tile kt + 2) into the freed stage.
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const int i = m >> 1, j = m & 1;
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[s][j], fa[i], acc[i][j], 0, 0, 0, 0, 0, 0);   // D^T = B A^T (g2_epilogue)
        __builtin_amdgcn_sched_barrier(0);
        if (j == 1) fa[i] = g8_frag(xa, 128 * wm + 32 * i, xs, lane);
        if (m == 4) fb[xs][0] = g8_frag(xb, 64 * wn, xs, lane);
        if (m == 6) fb[xs][1] = g8_frag(xb, 64 * wn + 32, xs, lane);
        if (s == 0) {
          if (m == 0) { if constexpr (A3) issue_aa_p(r2a, aprev, 0); else issue_p(r1a, r1b, cur ^ 1, 2); }
          if (m == 2) { if constexpr (A3) issue_aa_p(r2a, aprev, 1); else issue_p(r1a, r1b, cur ^ 1, 3); }
        } else {
          if (m == 0) {
            if constexpr (A3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // all but A tile kt+2 (this wave's 4 youngest DMA instructions)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile kt+1 have landed
            __syncthreads();                                    // ... and everybody's; every read of this stage has returned
          }
          if (m == 2) { if constexpr (A3) issue_bb_p(r2b, cur, 0); else issue_p(r2a, r2b, cur, 0); }
          if (m == 4) { if constexpr (A3) issue_bb_p(r2b, cur, 1); else issue_p(r2a, r2b, cur, 1); }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#else
      if (s == 0) {            // second half of tile kt+1's pieces (its first half left right after the last hand-over)
        if constexpr (A3) {    // ... A3: A tile kt+2 into the slot tile kt-1 left
          issue_aa(clampk(kt + 2), aprev, 0);
          issue_aa(clampk(kt + 2), aprev, 1);
        } else {
          issue(clampk(kt + 1), cur ^ 1, 2);
          issue(clampk(kt + 1), cur ^ 1, 3);
        }
      } else {
        if constexpr (A3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // all but A tile kt+2 (this wave's 4 youngest DMA instructions)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile kt+1 have landed
        __syncthreads();                                    // ... and everybody's; every read of this stage has returned
        if constexpr (A3) {
          issue_bb(clampk(kt + 2), cur, 0);
          issue_bb(clampk(kt + 2), cur, 1);
        } else {
          issue(clampk(kt + 2), cur, 0);
          issue(clampk(kt + 2), cur, 1);
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[xs][j] = g8_frag(xb, 64 * wn + 32 * j, xs, lane);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[s][j], fa[i], acc[i][j], 0, 0, 0, 0, 0, 0);   // D^T = B A^T (g2_epilogue)
        fa[i] = g8_frag(xa, 128 * wm + 32 * i, xs, lane);     // its registers are free: next step's fragment i
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
    }
#if G8_SPREAD
    {
      const int64_t step = kt + 3 <= kt_last ? 128 : 0;
      r1a = r2a; r1b = r2b;
      r2a += step; r2b += step;
    }
#endif
    cur ^= 1;
    aslot = anext;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // retire the trailing re-copies before LDS is reused
  __syncthreads();

  // (the operand scales are powers of two, so rounding the raw accumulator to bf16 first and un-scaling afterwards is the
  // same value as bf16(acc / (s_a s_b)): the functors that round first take the one-step bf16 epilogue of gemm256.h)
  if constexpr (epi_rounds_first<Epi>::value && epi_wide8<Epi>::value) g2_epilogue_bf16_w8<false>(acc, smem, bm, bn, 0, epi);
  else if constexpr (epi_rounds_first<Epi>::value) g2_epilogue_bf16<false>(acc, smem, bm, bn, 0, epi);
  else g2_epilogue(acc, smem, bm, bn, 0, epi);
}

template <class Epi>
constexpr int g8_lds_bytes() {
  return G2_A3 && epi_deep_a_ring<Epi>::value ? G2_A3_LDS_BYTES
         : epi_rounds_first<Epi>::value && G2_BF16_LDS_BYTES > G2_LDS_BYTES ? G2_BF16_LDS_BYTES : G2_LDS_BYTES;
}

// Streaming form (gemm256s.h) of the fp8 K = d GEMMs (the fp8 encoder: ten 128-byte K tiles per output tile, so the tile form's
// prologue + epilogue weigh twice what they do in the bf16 kernel): one continuous stream of K tiles per workgroup, the next output
// tile's first two K tiles requested by the hand-over slots of this tile's last two, wave-private epilogue in the CU's last 32 KiB.
// Functors opt in with STREAM (the bf16 interface of gemm256s.h: the accumulator is rounded to bf16 first, exact under the
// power-of-two operand scales).
template <class Epi>
__global__ __launch_bounds__(512, 2) void gemm256s_fp8_kernel(Gemm8Args g, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 2, wn = w & 3;
  const int nk = g.ktiles, ntiles = g.nbm * g.nbn;
  epi.s_begin();
  int blk = blockIdx.x;
  if (blk < ntiles) {
    int bm, bn;
    tile_coords(xcd_remap(blk, ntiles), g.nbm, g.nbn, bm, bn);
    const unsigned char* a_cur = g.A + (int64_t)(bm * G2_BM) * g.lda;
    const unsigned char* b_cur = g.B + (int64_t)(bn * G2_BN) * g.ldb;
    unsigned voff[4];          // (lda == ldb: both operands are [rows][d_p] -- the host launches this form only then)
#pragma unroll
    for (int q = 0; q < 4; ++q) voff[q] = g8_src_off(4 * w + q, lane, g.lda);
    typedef __attribute__((address_space(3))) char* lptr_t;
    const unsigned smem_base = (unsigned)(uintptr_t)(lptr_t)smem;
    const unsigned piece0 = (unsigned)__builtin_amdgcn_readfirstlane(4 * w * 1024);
    auto issue = [&](const unsigned char* pa, const unsigned char* pb, int stage, int q) {
      const unsigned dst = smem_base + stage * G2_STAGE_BYTES + piece0 + q * 1024;
      glds16_x2(pa, pb, voff[q], voff[q], dst, dst + G2_OPER_BYTES);
    };
    char* eb = smem + 2 * G2_STAGE_BYTES + w * 4096;
    const int rr = lane >> 3, c8 = 8 * (lane & 7);
#pragma unroll
    for (int q = 0; q < 4; ++q) issue(a_cur, b_cur, 0, q);
#pragma unroll
    for (int q = 0; q < 4; ++q) issue(a_cur + 128, b_cur + 128, 1, q);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __syncthreads();
    i32x8 fa[4], fb[2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = g8_frag(smem, 128 * wm + 32 * i, 0, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[0][j] = g8_frag(smem + G2_OPER_BYTES, 64 * wn + 32 * j, 0, lane);
    f32x16 acc[4][2];        // (started from a zero MFMA source by the first K step of every tile: gemm256s.h)
    int cur = 0;
#if G8_SPREAD
    // origins of the stream's K tiles kt + 1 and kt + 2 (carried: a select and an add per K tile instead of two per DMA group)
    const unsigned char* q1a = a_cur + 128;
    const unsigned char* q1b = b_cur + 128;
    const unsigned char* q2a = a_cur + 256;
    const unsigned char* q2b = b_cur + 256;
#endif
    for (;;) {
      const int nblk = blk + gridDim.x;
      const bool more = nblk < ntiles;
      int bm2 = bm, bn2 = bn;
      if (more) tile_coords(xcd_remap(nblk, ntiles), g.nbm, g.nbn, bm2, bn2);
      const unsigned char* a_nxt = more ? g.A + (int64_t)(bm2 * G2_BM) * g.lda : a_cur + (int64_t)(nk - 2) * 128;
      const unsigned char* b_nxt = more ? g.B + (int64_t)(bn2 * G2_BN) * g.ldb : b_cur + (int64_t)(nk - 2) * 128;
      auto pa = [&](int j) { return j < nk ? a_cur + (int64_t)j * 128 : a_nxt + (int64_t)(j - nk) * 128; };
      auto pb = [&](int j) { return j < nk ? b_cur + (int64_t)j * 128 : b_nxt + (int64_t)(j - nk) * 128; };
#if G8_SPREAD
      if (nk == 2) { q2a = a_nxt; q2b = b_nxt; }      // (two K tiles per output tile: stream tile 2 is the NEXT output tile's first, known only now)
#endif
      const int row_w = bm * G2_BM + 128 * wm, col_l = bn * G2_BN + 64 * wn + c8;
      typename Epi::SPre pre0[4];
      auto ktile = [&](auto first_tag, int kt) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const char* sa = smem + cur * G2_STAGE_BYTES;
        const char* sb = sa + G2_OPER_BYTES;
        const char* na = smem + (cur ^ 1) * G2_STAGE_BYTES;
        const char* nb = na + G2_OPER_BYTES;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const char* xa = s == 0 ? sa : na;
          const char* xb = s == 0 ? sb : nb;
          const int xs = s == 0 ? 1 : 0;
#if G8_SPREAD
          // (one piece of side work per MFMA gap and carried source origins: see gemm256_fp8_kernel and gemm256s.h)
#pragma unroll
          for (int m = 0; m < 8; ++m) {
            const int i = m >> 1, j = m & 1;
            if constexpr (FIRST) {
              if (s == 0) {
                f32x16 zero;
#pragma unroll
                for (int e = 0; e < 16; ++e) zero[e] = 0.f;
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[s][j], fa[i], zero, 0, 0, 0, 0, 0, 0);
              } else {
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[s][j], fa[i], acc[i][j], 0, 0, 0, 0, 0, 0);
              }
            } else {
              acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[s][j], fa[i], acc[i][j], 0, 0, 0, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (j == 1) fa[i] = g8_frag(xa, 128 * wm + 32 * i, xs, lane);
            if (m == 4) fb[xs][0] = g8_frag(xb, 64 * wn, xs, lane);
            if (m == 6) fb[xs][1] = g8_frag(xb, 64 * wn + 32, xs, lane);
            if (s == 0) {
              if (m == 0) issue(q1a, q1b, cur ^ 1, 2);
              if (m == 2) issue(q1a, q1b, cur ^ 1, 3);
            } else {
              if (m == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
              }
              if (m == 2) issue(q2a, q2b, cur, 0);
              if (m == 4) issue(q2a, q2b, cur, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
#else
          if (s == 0) {
            issue(pa(kt + 1), pb(kt + 1), cur ^ 1, 2);
            issue(pa(kt + 1), pb(kt + 1), cur ^ 1, 3);
          } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            issue(pa(kt + 2), pb(kt + 2), cur, 0);
            issue(pa(kt + 2), pb(kt + 2), cur, 1);
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) fb[xs][j] = g8_frag(xb, 64 * wn + 32 * j, xs, lane);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#ifdef G8S_PROXY16
              // TIMING PROXY ONLY (tools/build_variant.sh g8sproxy16 -DG8S_PROXY16; results WRONG): each 32x32x64 MFMA as two 16x16x128 on
              // the same operand registers and a quarter each of the accumulator -- would the smaller shape lift the clock for e4m3 too?
              {
                typedef __attribute__((ext_vector_type(4))) float f32x4_;
                f32x16& C = acc[i][j];
                f32x4_ q0 = s ? f32x4_{C[4], C[5], C[6], C[7]} : f32x4_{C[0], C[1], C[2], C[3]};
                f32x4_ q1 = s ? f32x4_{C[12], C[13], C[14], C[15]} : f32x4_{C[8], C[9], C[10], C[11]};
                q0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[s][j], fa[i], q0, 0, 0, 0, 0, 0, 0);
                q1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fb[s][j], fa[i], q1, 0, 0, 0, 0, 0, 0);
                if (s) { C[4] = q0[0]; C[5] = q0[1]; C[6] = q0[2]; C[7] = q0[3]; C[12] = q1[0]; C[13] = q1[1]; C[14] = q1[2]; C[15] = q1[3]; }
                else { C[0] = q0[0]; C[1] = q0[1]; C[2] = q0[2]; C[3] = q0[3]; C[8] = q1[0]; C[9] = q1[1]; C[10] = q1[2]; C[11] = q1[3]; }
              }
              if constexpr (false) {
                if (s == 0) {
                } else {
                }
              } else if constexpr (false) {
              }
#else
              if constexpr (FIRST) {
                if (s == 0) {
                  f32x16 zero;
#pragma unroll
                  for (int e = 0; e < 16; ++e) zero[e] = 0.f;
                  acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[s][j], fa[i], zero, 0, 0, 0, 0, 0, 0);
                } else {
                  acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[s][j], fa[i], acc[i][j], 0, 0, 0, 0, 0, 0);
                }
              } else {
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[s][j], fa[i], acc[i][j], 0, 0, 0, 0, 0, 0);
              }
#endif
            }
            fa[i] = g8_frag(xa, 128 * wm + 32 * i, xs, lane);
            __builtin_amdgcn_sched_barrier(0);
          }
#endif
        }
#if G8_SPREAD
        {   // the stream's tile kt + 3: this output tile's, or the first of the next one's (gemm256s.h)
          const bool wrap = kt + 3 == nk;
          q1a = q2a; q1b = q2b;
          q2a = wrap ? a_nxt : q2a + 128;
          q2b = wrap ? b_nxt : q2b + 128;
        }
#endif
        cur ^= 1;
      };
      ktile(std::true_type{}, 0);
      for (int kt = 1; kt < nk; ++kt) ktile(std::false_type{}, kt);
      // (the functor's loads only now: with 64 fragment registers next to the 128 accumulators the K loop has no room to carry them
      // through its last K tile -- requested there, the kernel spilled)
      epi.s_tile(row_w, col_l);
#pragma unroll
      for (int q = 0; q < 4; ++q) pre0[q] = epi.s_prefetch(row_w + 8 * q + rr, col_l);
      if (row_w + 128 > epi.s_rows()) g2s_epilogue<true>(acc, eb, row_w, col_l, pre0, epi);
      else g2s_epilogue<false>(acc, eb, row_w, col_l, pre0, epi);
      if (!more) break;
      blk = nblk; bm = bm2; bn = bn2;
      a_cur = a_nxt; b_cur = b_nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  epi.s_end(reinterpret_cast<float*>(smem + 2 * G2_STAGE_BYTES));
}
