// 256x256x64 bf16 MFMA GEMM for gfx950: the large-shape sibling of gemm.h (same operand modes, K segments, split-K and
// epilogue functors; used whenever both output dimensions are multiples of 256).
//
//   512 threads = 8 waves (2 per SIMD) in a 2 (M) x 4 (N) arrangement, each wave a 128x64 output = 4x2
//   v_mfma_f32_32x32x16_bf16 tiles (128 accumulator registers, <= 256 registers per wave so two waves share a SIMD and
//   one covers the other's LDS-DMA issue and barrier stalls).
//   Per 64-deep K tile a wave issues 32 MFMAs for 24 fragment reads (0.75 LDS reads per MFMA, 1.0 in the 128x128
//   kernel) and 8 LDS-DMA pieces (0.25 KiB of staging per MFMA, 0.5 there).
//   Staging: global_load_lds_dwordx4 straight into two 64 KiB stages; the XOR swizzles of gemm.h are applied to the
//   per-lane SOURCE chunk so that the 1 KiB a piece writes is contiguous in LDS.  The K loop is rotated around its
//   barrier (see the loop) so that tile t+1's first fragments and tile t+2's DMA are in flight under tile t's last MFMAs.
//   Epilogue: the tile leaves as four 128x128 sub-tiles through fp32 LDS, two at a time, one per 256-thread half,
//   through the same functors as gemm.h (their block reductions are 256-thread-group local).
//   Round 6: when both operands have the same layout the K loop runs on v_mfma_f32_16x16x32_bf16 instead -- 8x4 tiles of 16x16 per wave, 64
//   MFMAs of 16 cycles per K tile, one piece of side work per MFMA (G2_M16: row x row, the decoder; G2_M16K: k-major x k-major through the
//   half ring, the weight gradient) -- with the same images, swizzles, rings and epilogues, bit-identical output; the mixed forms keep 32x32x16.
#pragma once
#include "gemm.h"

#ifdef G2X_STAMP
__device__ unsigned long long* g2x_stamps = nullptr;      // tools/kbench: [tile][4] = prologue, K loop, epilogue cycles, start
#endif
constexpr int G2_BM = 256, G2_BN = 256;
constexpr int G2_OPER_BYTES = 256 * GEMM_BK * 2;                      // 32 KiB per operand tile
constexpr int G2_STAGE_BYTES = 2 * G2_OPER_BYTES;                     // 64 KiB
constexpr int G2_SUB_FLOATS = 128 * GEMM_EPI_PITCH;                   // one 128x128 fp32 sub-tile
constexpr int G2_LDS_BYTES = 2 * G2_SUB_FLOATS * 4 > 2 * G2_STAGE_BYTES ? 2 * G2_SUB_FLOATS * 4 : 2 * G2_STAGE_BYTES;

// Source byte offset (relative to the tile origin) of the 16 B that lane `lane` of DMA piece p (0..31) moves; the piece
// lands at LDS bytes [1024 p, 1024 p + 1024) of the operand image.
template <int MODE>
__device__ __forceinline__ unsigned g2_src_off(int p, int lane, int64_t ld) {
  if constexpr (MODE == OP_ROW) {
    // image [256 rows][128 B]: piece = rows 8p..8p+7; physical chunk pc of row r holds source chunk pc ^ ((r >> 1) & 7)
    const int r = 8 * p + (lane >> 3), pc = lane & 7;
    return (unsigned)((r * ld + ((pc ^ ((r >> 1) & 7)) << 3)) * 2);
  } else {
    // two [64 k][128 cols] images of 16 KiB (256-B rows): piece = k rows 4pp..4pp+3 of image p >> 4
    const int sub = p >> 4, r = 4 * (p & 15) + (lane >> 4), pc = lane & 15;
    return (unsigned)((r * ld + sub * 128 + ((pc ^ (((r & 3) << 2) | ((r >> 2) & 3))) << 3)) * 2);
  }
}

template <int MODE>
__device__ __forceinline__ bf16x8 g2_frag(const char* img, int base32, int kk, int lane) {
  if constexpr (MODE == OP_ROW) return frag_read<OP_ROW>(img, base32, kk, lane);
  else return frag_read<OP_KMAJOR>(img + (base32 >> 7) * 16384, base32 & 127, kk, lane);
}

// ---- half-granular staging of the k-major x k-major GEMMs (the K loop's ring of four 32 KiB slots: stage x k half) ------
// A k-major image is k rows of 256 B (g2_src_off), so k rows 32 h .. 32 h + 31 of a sub-image are whole pieces: 8 h .. 8 h + 7
// of it.  Wave w moves two pieces (q = 0, 1) of each operand per half: k rows 32 h + 8 (w % 4) .. + 7 of sub-image w / 4.
// (Row-major operands would need half images with 64-byte rows, and LDS-DMA from 64-byte global segments made every
// row-major shape 8 % slower -- tools/kbench/gemm256_half_row.patch -- so those GEMMs hand their stages over whole.)
__device__ __forceinline__ int g2h_piece(int w, int h, int q) { return (w >> 2) * 16 + 8 * h + 2 * (w & 3) + q; }

// ---- accumulators -> the epilogues' LDS images, for both K-loop forms ------------------------------------------------------------
// 32x32x16 loop: acc[i][j] = one 32x32 tile as f32x16.  The K loops feed the MFMA with the operands swapped -- D^T = B A^T -- so a lane holds
// output row 32 i + lane % 32 and, per group of four accumulator registers, FOUR CONSECUTIVE COLUMNS 32 j + 8 g + 4 (lane / 32).
// 16x16x32 loop (G2_M16): acc[i16][j16] = one 16x16 tile as f32x4 -- lane (r16 = lane % 16, q = lane / 16) holds columns 16 j16 + 4 q .. + 3
// of row 16 i16 + r16.  Either way a lane writes 4 consecutive columns per store; the functor-side reads do not change.
typedef __attribute__((ext_vector_type(4))) float g2_f32x4;
__device__ __forceinline__ void g2_put_f32(const f32x16 (&acc)[4][2], float* dst0, int lane) {
  float* dst = dst0 + (lane & 31) * GEMM_EPI_PITCH + 4 * (lane >> 5);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        *reinterpret_cast<f32x4*>(dst + 32 * i * GEMM_EPI_PITCH + 32 * j + 8 * gq) =
            f32x4{acc[i][j][4 * gq], acc[i][j][4 * gq + 1], acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]};
}
__device__ __forceinline__ void g2_put_f32(const g2_f32x4 (&acc)[8][4], float* dst0, int lane) {
  float* dst = dst0 + (lane & 15) * GEMM_EPI_PITCH + 4 * (lane >> 4);
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<f32x4*>(dst + 16 * i * GEMM_EPI_PITCH + 16 * j) = f32x4{acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
}
template <int PITCH>
__device__ __forceinline__ void g2_put_bf16(const f32x16 (&acc)[4][2], bf16_t* dst0, int lane) {
  bf16_t* dst = dst0 + (lane & 31) * PITCH + 4 * (lane >> 5);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        *reinterpret_cast<bf16x4*>(dst + 32 * i * PITCH + 32 * j + 8 * gq) =
            bf16x4{(bf16_t)acc[i][j][4 * gq], (bf16_t)acc[i][j][4 * gq + 1], (bf16_t)acc[i][j][4 * gq + 2], (bf16_t)acc[i][j][4 * gq + 3]};
}
template <int PITCH>
__device__ __forceinline__ void g2_put_bf16(const g2_f32x4 (&acc)[8][4], bf16_t* dst0, int lane) {
  bf16_t* dst = dst0 + (lane & 15) * PITCH + 4 * (lane >> 4);
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<bf16x4*>(dst + 16 * i * PITCH + 16 * j) = bf16x4{(bf16_t)acc[i][j][0], (bf16_t)acc[i][j][1], (bf16_t)acc[i][j][2], (bf16_t)acc[i][j][3]};
}

// Epilogue shared by the bf16 and fp8 256x256 kernels: the tile leaves as two passes (sub-tile columns), each pass two
// 128x128 sub-tiles (rows), one per 256-thread half, through fp32 LDS and the row-major functor.
template <class Epi, class Acc>
__device__ __forceinline__ void g2_epilogue(Acc& acc, char* smem, int bm, int bn, int split, Epi& epi) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 2, wn = w & 3;
  float* tile = reinterpret_cast<float*>(smem);
  const int half = t >> 8, tl = t & 255;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    if ((wn >> 1) == pass) {
      g2_put_f32(acc, tile + wm * G2_SUB_FLOATS + 64 * (wn & 1), lane);
    }
    lds_barrier();
    const float* src = tile + half * G2_SUB_FLOATS;
    const int row0 = bm * G2_BM + 128 * half, col0 = bn * G2_BN + 128 * pass;
    epi.tile_begin(row0, col0, split);
    {
      const int c4 = (tl & 31) * 4;
      constexpr int NB = epi_prefetch_batch<Epi>::value;
#pragma unroll
      for (int b0 = 0; b0 < 16; b0 += NB) {
        typename Epi::Pre pre[NB];
#pragma unroll
        for (int it = 0; it < NB; ++it) pre[it] = epi.prefetch(row0 + (tl >> 5) + 8 * (b0 + it), col0 + c4);
#pragma unroll
        for (int it = 0; it < NB; ++it) {
          const int row = (tl >> 5) + 8 * (b0 + it);
          const f32x4 v = *reinterpret_cast<const f32x4*>(&src[row * GEMM_EPI_PITCH + c4]);
          epi_apply(epi, b0 + it, row0 + row, col0 + c4, v, pre[it], 0);
        }
      }
    }
    lds_barrier();
    epi.tile_end(tile + half * G2_SUB_FLOATS);
    lds_barrier();

  }
}

// Epilogue for functors whose first act on an accumulator is its rounding to bf16 (static constexpr bool
// ROUNDS_BF16_FIRST: the bf16-autocast GEMM outputs -- c, x_hat, dc, d acts): the WHOLE 256x256 tile goes to LDS as bf16
// (pitch 260 elements: 133 120 bytes) in one step, so no wave sits on its 128 accumulator registers while others are
// served (that cost the functors with global loads 30-70 spilled registers or a shallower prefetch), the fp32 round trip
// through LDS is halved, and a tile needs 3 barriers less.  The functor still receives fp32 values (exactly the rounded
// ones) and works on 128x128 sub-tiles: half h of the workgroup takes rows 128 h.., first the columns 0..127, then 128..255.
constexpr int G2_BF16_PITCH = 260;     // 520-byte rows: the 8-byte writes of 16 lanes (16 rows) fall on 32 different banks
constexpr int G2_BF16_TILE_BYTES = 256 * G2_BF16_PITCH * 2;            // 133 120
constexpr int G2_BF16_SCRATCH_FLOATS = 1024;                           // per 256-thread half, for tile_end()
constexpr int G2_BF16_LDS_BYTES = G2_BF16_TILE_BYTES + 4 * G2_BF16_SCRATCH_FLOATS * 4;   // scratch per (half, pass): no barrier between the passes

template <class E, class = void>
struct epi_rounds_first { static constexpr bool value = false; };
template <class E>
struct epi_rounds_first<E, std::void_t<decltype(E::ROUNDS_BF16_FIRST)>> { static constexpr bool value = E::ROUNDS_BF16_FIRST; };

template <bool FINAL_BARRIER, class Epi, class Acc>
__device__ __forceinline__ void g2_epilogue_bf16(Acc& acc, char* smem, int bm, int bn, int split, Epi& epi) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 2, wn = w & 3;
  bf16_t* tile = reinterpret_cast<bf16_t*>(smem);
  const int half = t >> 8, tl = t & 255, c4 = (tl & 31) * 4;
  const int row0 = bm * G2_BM + 128 * half;
  // A thread's 32 elements (2 passes x 16 rows) go in batches of NB; the global loads of batch q + 1 (Epi::prefetch) are
  // issued before batch q is applied, and those of batch 0 before the accumulators go to LDS: every memory round trip of
  // the epilogue runs under the previous phase (with the loads issued at the head of each batch a functor that reads a
  // [M x n] operand -- the latent in dpre, x in the decoder -- paid four exposed latencies per tile).
  constexpr int NB = epi_prefetch_batch<Epi>::value, BPP = 16 / NB, NQ = 2 * BPP;
  typename Epi::Pre pre[2][NB];
  auto prefetch_batch = [&](int q, typename Epi::Pre (&dst)[NB]) {
    const int pass = q / BPP, b0 = (q % BPP) * NB;
#pragma unroll
    for (int it = 0; it < NB; ++it) dst[it] = epi.prefetch(row0 + (tl >> 5) + 8 * (b0 + it), bn * G2_BN + 128 * pass + c4);
  };
  prefetch_batch(0, pre[0]);
  g2_put_bf16<G2_BF16_PITCH>(acc, tile + (128 * wm) * G2_BF16_PITCH + 64 * wn, lane);
  lds_barrier();
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int pass = q / BPP, b0 = (q % BPP) * NB;
    const int col0 = bn * G2_BN + 128 * pass;
    const bf16_t* src = tile + (128 * half) * G2_BF16_PITCH + 128 * pass;
    if (q + 1 < NQ) prefetch_batch(q + 1, pre[(q + 1) & 1]);
    if (b0 == 0) epi.tile_begin(row0, col0, split);
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      const int row = (tl >> 5) + 8 * (b0 + it);
      const bf16x4 qv = *reinterpret_cast<const bf16x4*>(&src[row * G2_BF16_PITCH + c4]);
      const f32x4 v = {(float)qv[0], (float)qv[1], (float)qv[2], (float)qv[3]};
      epi_apply(epi, b0 + it, row0 + row, col0 + c4, v, pre[q & 1][it], 0);
    }
    if (b0 + NB == 16)       // scratch per (half, pass): no barrier between the passes
      epi.tile_end(reinterpret_cast<float*>(smem + G2_BF16_TILE_BYTES) + (2 * pass + half) * G2_BF16_SCRATCH_FLOATS);
  }
  if (FINAL_BARRIER) lds_barrier();        // the tile and the scratch areas are free again (persistent instantiation only)
}

// The same one-step bf16 epilogue with EIGHT consecutive columns per thread (round 4): functors that declare
// `static constexpr bool WIDE8 = true` and `apply8(row, col, v0, v1, pre0, pre1)` store 16 bytes per lane and row instead of 8 --
// half the store instructions of a store-issue-bound tail (MI355X_MICROARCH.md: 8-byte accesses run at 0.54-0.70 x the 16-byte
// rate).  A thread owns columns 8 (tl % 16) .. + 7 of rows tl / 16 + 16 it (it = 0..7) of each 128-column pass; Epi::prefetch
// keeps its 4-column granularity (two per call of apply8), batches of NB / 2 calls keep the prefetched registers the same.
#ifndef G2_WIDE8
#define G2_WIDE8 1           // tools/build_variant.sh A/B switch: 0 = every functor through the 4-column form
#endif
template <class E, class = void>
struct epi_wide8 { static constexpr bool value = false; };
template <class E>
struct epi_wide8<E, std::void_t<decltype(E::WIDE8)>> { static constexpr bool value = G2_WIDE8 && E::WIDE8; };

template <bool FINAL_BARRIER, class Epi, class Acc>
__device__ __forceinline__ void g2_epilogue_bf16_w8(Acc& acc, char* smem, int bm, int bn, int split, Epi& epi) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 2, wn = w & 3;
  bf16_t* tile = reinterpret_cast<bf16_t*>(smem);
  const int half = t >> 8, tl = t & 255, c8 = (tl & 15) * 8;
  const int row0 = bm * G2_BM + 128 * half;
  constexpr int NB = epi_prefetch_batch<Epi>::value / 2, BPP = 8 / NB, NQ = 2 * BPP;
  static_assert(NB >= 1 && 8 % NB == 0, "prefetch batch");
  typename Epi::Pre pre[2][NB][2];
  auto prefetch_batch = [&](int q, typename Epi::Pre (&dst)[NB][2]) {
    const int pass = q / BPP, b0 = (q % BPP) * NB;
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      const int r = row0 + (tl >> 4) + 16 * (b0 + it), cc = bn * G2_BN + 128 * pass + c8;
      dst[it][0] = epi.prefetch(r, cc);
      dst[it][1] = epi.prefetch(r, cc + 4);
    }
  };
  prefetch_batch(0, pre[0]);
  g2_put_bf16<G2_BF16_PITCH>(acc, tile + (128 * wm) * G2_BF16_PITCH + 64 * wn, lane);
  lds_barrier();
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int pass = q / BPP, b0 = (q % BPP) * NB;
    const int col0 = bn * G2_BN + 128 * pass;
    const bf16_t* src = tile + (128 * half) * G2_BF16_PITCH + 128 * pass;
    if (q + 1 < NQ) prefetch_batch(q + 1, pre[(q + 1) & 1]);
    if (b0 == 0) epi.tile_begin(row0, col0, split);
#pragma unroll
    for (int it = 0; it < NB; ++it) {
      const int row = (tl >> 4) + 16 * (b0 + it);
      // (rows are 520 bytes apart: 8-byte aligned, so two 8-byte reads)
      const bf16x4 q0 = *reinterpret_cast<const bf16x4*>(&src[row * G2_BF16_PITCH + c8]);
      const bf16x4 q1 = *reinterpret_cast<const bf16x4*>(&src[row * G2_BF16_PITCH + c8 + 4]);
      const f32x4 v0 = {(float)q0[0], (float)q0[1], (float)q0[2], (float)q0[3]};
      const f32x4 v1 = {(float)q1[0], (float)q1[1], (float)q1[2], (float)q1[3]};
      epi.apply8(row0 + row, col0 + c8, v0, v1, pre[q & 1][it][0], pre[q & 1][it][1]);
    }
    if (b0 + NB == 8)
      epi.tile_end(reinterpret_cast<float*>(smem + G2_BF16_TILE_BYTES) + (2 * pass + half) * G2_BF16_SCRATCH_FLOATS);
  }
  if (FINAL_BARRIER) lds_barrier();
}

// Functors of long-K row-major GEMMs whose A operand streams from HBM (static constexpr bool DEEP_A_RING: the decoder,
// x_hat = c W^T with K = n_dict) get a THREE-deep ring for the A tiles (see the K loop).
template <class E, class = void>
struct epi_deep_a_ring { static constexpr bool value = false; };
template <class E>
struct epi_deep_a_ring<E, std::void_t<decltype(E::DEEP_A_RING)>> { static constexpr bool value = E::DEEP_A_RING; };
constexpr int G2_A3_LDS_BYTES = 5 * G2_OPER_BYTES;      // 3 A slots + 2 B slots = 160 KiB: the whole LDS of a CU

#ifndef G2_A3
#define G2_A3 1              // tools/kbench A/B switch: 0 = no three-deep A ring (Epi::DEEP_A_RING ignored)
#endif
#ifndef G2_M16
#define G2_M16 1             // tools/kbench A/B switch: 0 = the row x row K loop on v_mfma_f32_32x32x16_bf16 like the other operand modes
#endif
#ifndef G2_RUNPTR
#define G2_RUNPTR 1          // tools/kbench A/B switch: 0 = the 16x16x32 loops form tile kt + 2's source pointers anew per K tile
#endif
#ifndef G2_M16K
#define G2_M16K 1            // tools/kbench A/B switch: 0 = the k-major x k-major K loop (weight gradient, half ring) on v_mfma_f32_32x32x16_bf16
#endif
#ifndef G2_HALF_KMAJOR
#define G2_HALF_KMAJOR 1     // tools/kbench A/B switch: 0 = the k-major GEMMs hand their stages over whole, like the others
#endif

// One 256x256 output tile (workgroup-level id `blk` of nblk).  g.nbm / g.nbn count 256-wide tiles here.
template <int AMODE, int BMODE, bool PERSIST, class Epi>
__device__ __forceinline__ void gemm256_tile(const GemmArgs& g, Epi& epi, char* smem, int blk, int nbm, int nbn, int ktiles0,
                                             int ktiles, int splits) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 2, wn = w & 3;
  const int nblk = nbm * nbn * splits;
  int id, split, kt_begin, kt_end;
  if (!PERSIST && g.tail_tiles > 0) {
    // (splits == 1) whole tiles first, then the pieces of the tail tiles: piece-major, so that the workgroups of a round walk
    // the same K range of the shared operand
    const int main_tiles = nbm * nbn - g.tail_tiles;
    if (blk < main_tiles) {
      id = xcd_remap(blk, main_tiles);
      split = 0;
      kt_begin = 0;
      kt_end = ktiles;
    } else {
      const int j = blk - main_tiles, piece = j / g.tail_tiles, tt = j - piece * g.tail_tiles;
      id = main_tiles + tt;
      split = 1 + piece * g.tail_tiles + tt;
      kt_begin = (int)((int64_t)ktiles * piece / g.tail_pieces);
      kt_end = (int)((int64_t)ktiles * (piece + 1) / g.tail_pieces);
    }
  } else {
    id = xcd_remap(blk, nblk);
    split = id / (nbm * nbn);
    id -= split * (nbm * nbn);
    kt_begin = (int)((int64_t)ktiles * split / splits);
    kt_end = (int)((int64_t)ktiles * (split + 1) / splits);
  }
  int bm, bn;
  tile_coords(id, nbm, nbn, bm, bn, g.group_m > 0 ? g.group_m : GEMM_GROUP_M);

#ifdef G2X_PHASE
  // experiment: half of the FIRST round's workgroups start late by G2X_PHASE x 8128 cycles, so that the CUs' epilogue bursts
  // (all of them otherwise write their 128 KiB tiles at the same moment) fall under the other half's K loops
  if ((int)blockIdx.x < 256 && ((blockIdx.x >> 3) & 1)) {
#pragma unroll 1
    for (int i = 0; i < G2X_PHASE; ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif
#ifdef G2X_STAMP
  unsigned long long st0 = __builtin_readcyclecounter(), st1 = 0, st2 = 0;
#endif
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // Row-major x row-major (the decoder and every other GEMM whose operands both have K contiguous): the K loop runs on
  // v_mfma_f32_16x16x32_bf16 -- 64 MFMAs of 16 cycles per K tile instead of 32 of 32, with the fragment reads and DMA pieces spread one per
  // MFMA as in the streaming form (gemm256s.h, where the shape is described): same images, same swizzle, same ring, bit-identical output.
  constexpr bool M16 = G2_M16 && AMODE == OP_ROW && BMODE == OP_ROW;
  g2_f32x4 acc16[8][4];
  if constexpr (M16 || (G2_M16K && G2_HALF_KMAJOR && AMODE == OP_KMAJOR && BMODE == OP_KMAJOR)) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc16[i][j] = g2_f32x4{0.f, 0.f, 0.f, 0.f};
  }

  auto a_ptr = [&](int kt) -> const bf16_t* {
    const bool s1 = kt >= ktiles0;
    const bf16_t* base = s1 ? g.A1 : g.A0;
    const int k = (s1 ? kt - ktiles0 : kt) * GEMM_BK;
    if constexpr (AMODE == OP_ROW) return base + (int64_t)(bm * G2_BM) * g.lda + k;
    else return base + (int64_t)k * g.lda + bm * G2_BM;
  };
  auto b_ptr = [&](int kt) -> const bf16_t* {
    const bool s1 = kt >= ktiles0;
    const bf16_t* base = s1 ? g.B1 : g.B0;
    const int k = (s1 ? kt - ktiles0 : kt) * GEMM_BK;
    if constexpr (BMODE == OP_ROW) return base + (int64_t)(bn * G2_BN) * g.ldb + k;
    else return base + (int64_t)k * g.ldb + bn * G2_BN;
  };

  // wave w moves pieces 4w..4w+3 of both operand tiles; pair q = (A piece 4w+q, B piece 4w+q)
  unsigned voff_a[4], voff_b[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    voff_a[q] = g2_src_off<AMODE>(4 * w + q, lane, g.lda);
    voff_b[q] = g2_src_off<BMODE>(4 * w + q, lane, g.ldb);
  }
  typedef __attribute__((address_space(3))) char* lptr_t;
  const unsigned smem_base = (unsigned)(uintptr_t)(lptr_t)smem;
  const unsigned piece0 = (unsigned)__builtin_amdgcn_readfirstlane(4 * w * 1024);
  auto issue = [&](int kt, int stage, int q) {
    const unsigned dst = smem_base + stage * G2_STAGE_BYTES + piece0 + q * 1024;
    glds16_x2(a_ptr(kt), b_ptr(kt), voff_a[q], voff_b[q], dst, dst + G2_OPER_BYTES);
  };
  // A3 (row-major x row-major with Epi::DEEP_A_RING: the decoder, K = n_dict, whose A operand -- the latent -- streams from
  // HBM while the weight panel stays in L2): three 32 KiB slots for the A tiles at LDS 0, two for the B tiles behind them.
  // A tile t + 2 is requested at the START of tile t (into the slot tile t - 1 left at its hand-over): two tiles of lead;
  // B tile t + 2 right after hand-over t, as before.  With two whole stages a wave waited ~770 cycles per K tile for its
  // own pieces at this shape (tools/kbench -DG2X_WAITSTAMP).
  constexpr bool A3 = G2_A3 && epi_deep_a_ring<Epi>::value;
  auto issue_aa = [&](int kt, int slot, int qp) {
    const unsigned dst = smem_base + slot * G2_OPER_BYTES + piece0 + 2 * qp * 1024;
    glds16_x2(a_ptr(kt), a_ptr(kt), voff_a[2 * qp], voff_a[2 * qp + 1], dst, dst + 1024);
  };
  auto issue_bb = [&](int kt, int slot, int qp) {
    const unsigned dst = smem_base + (3 + slot) * G2_OPER_BYTES + piece0 + 2 * qp * 1024;
    glds16_x2(b_ptr(kt), b_ptr(kt), voff_b[2 * qp], voff_b[2 * qp + 1], dst, dst + 1024);
  };
  // HALF (k-major x k-major: the weight-gradient GEMMs, K = the batch rows, every operand byte streams from HBM once): the
  // two stages are handed over in k HALVES -- a ring of four 32 KiB slots, three half tiles (1.5 K tiles) of lead for every
  // piece instead of one tile.  With whole-stage hand-over a wave waited ~410 cycles per K tile for its own pieces
  // (tools/kbench -DG2X_WAITSTAMP); with halves ~20, and the GEMM is 5-7 % faster with bit-identical output.
  constexpr bool HALF = !A3 && G2_HALF_KMAJOR && AMODE == OP_KMAJOR && BMODE == OP_KMAJOR;
  if constexpr (HALF) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      voff_a[q] = g2_src_off<AMODE>(g2h_piece(w, q >> 1, q & 1), lane, g.lda);
      voff_b[q] = g2_src_off<BMODE>(g2h_piece(w, q >> 1, q & 1), lane, g.ldb);
    }
  }
  // (M16K) the half ring's K loop on v_mfma_f32_16x16x32_bf16: one 32-deep K step = one half.  Transposed fragment reads of a 16-column block
  // `blk` of a k-major image: the 16-lane group G = lane / 16 reads k rows 32 h + 8 G + q (q = lane % 16 / 4) and + 4, columns 16 blk + 4 p ..
  // + 3 (p = lane % 4), and receives column lane % 16 with k = 8 G .. 8 G + 7 -- the operand layout of the instruction.  In the image's chunk
  // swizzle (chunk ^ ((row & 3) << 2 | (row >> 2 & 3))) the block enters as an XOR of address bits 5-7 and the second read is the first's
  // address ^ 16, + 1024: two lane offsets per operand serve every read; the half is an immediate (8192 h).  Conflict-free: the two groups of a
  // 32-lane service half sit 8 rows apart, their swizzles differ in chunk bit 1, each covers eight distinct 16-byte chunks.
  constexpr bool M16K = G2_M16K && HALF;
  unsigned ka0 = 0, ka1 = 0, kb0 = 0, kb1 = 0;
  if constexpr (M16K) {
    const int G = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int swz = (q << 2) | ((2 * G) & 3);
    const int rowb = (8 * G + q) * 256 + 8 * (p & 1);
    ka0 = (unsigned)(rowb + (((p >> 1) ^ swz) << 4) + wm * 16384);
    kb0 = (unsigned)(rowb + ((((p >> 1) ^ swz) ^ (8 * (wn & 1))) << 4) + (wn >> 1) * 16384 + G2_OPER_BYTES);
    ka1 = ka0 ^ 16u;
    kb1 = kb0 ^ 16u;
    asm volatile("" : "+v"(ka0), "+v"(ka1), "+v"(kb0), "+v"(kb1));
  }
  auto tr16 = [&](unsigned o0, unsigned o1, int blk, int h) -> bf16x8 {
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4_t;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const unsigned x = (unsigned)blk << 5;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(uintptr_t)((o0 ^ x) + 8192u * h));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t)(uintptr_t)((o1 ^ x) + 8192u * h + 1024u));
    return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  auto issue_h = [&](int kt, int stage, int h, int q) {
    const unsigned dst = smem_base + stage * G2_STAGE_BYTES + (unsigned)__builtin_amdgcn_readfirstlane(g2h_piece(w, h, q) * 1024);
    glds16_x2(a_ptr(kt), b_ptr(kt), voff_a[2 * h + q], voff_b[2 * h + q], dst, dst + G2_OPER_BYTES);
  };

  // Tile t lives in stage t & 1.  The hand-over barrier B_t sits inside the LAST MFMA group of tile t: by then every
  // fragment of tile t is in registers and tile t+1 has landed, so the slots after B_t already read tile t+1's first
  // fragments (no tile starts with an exposed LDS round trip) and refill the freed stage with tile t+2 (half of its
  // pieces right after B_t, half in the first group of tile t+1: a full tile of lead time).
  // Every group is 8 MFMAs with ONE piece of other work in the slot behind each (an LDS-DMA pair, or the read of one
  // fragment of the next group): clustered in front of the group, the 6 fragment reads and the DMA pairs left the matrix
  // pipe idle while they issued (tools/kbench: without the reads the K = 1280 GEMM ran 22 % faster, without the DMA 15 %),
  // and the last group ran its barrier + 4 DMA pieces + 6 reads before its first MFMA.
  const int kt_last = kt_end - 1;
  auto clampk = [&](int kt) { return kt < kt_last ? kt : kt_last; };   // past-the-end tiles re-copy the last one (harmless)
  // (16x16x32 loops) the source tile origins of K tile min(kt + 2, last) as RUNNING pointers: a_ptr() / b_ptr() are a segment select and a
  // 64-bit multiply by the leading dimension each -- 43 scalar instructions in one MFMA gap per K tile, at the same moment in both waves of a
  // SIMD (they leave the same barrier): here a compare, a select and a 64-bit add
  constexpr bool RUN2 = G2_RUNPTR && ((M16 && A3) || M16K);
  const bf16_t* pa2 = nullptr;
  const bf16_t* pb2 = nullptr;
  const bf16_t* pa2_seg1 = nullptr;
  const bf16_t* pb2_seg1 = nullptr;
  int64_t pa2_step = 0, pb2_step = 0;
  if constexpr (RUN2) {
    if (kt_begin < kt_end) {
      pa2 = a_ptr(clampk(kt_begin + 2));
      pb2 = b_ptr(clampk(kt_begin + 2));
    }
    if (g.A1 != nullptr) {      // (one K segment: kn == ktiles0 never holds inside the walk, the origins stay unused)
      pa2_seg1 = AMODE == OP_ROW ? g.A1 + (int64_t)(bm * G2_BM) * g.lda : g.A1 + bm * G2_BM;
      pb2_seg1 = BMODE == OP_ROW ? g.B1 + (int64_t)(bn * G2_BN) * g.ldb : g.B1 + bn * G2_BN;
    }
    pa2_step = AMODE == OP_ROW ? (int64_t)GEMM_BK : (int64_t)GEMM_BK * g.lda;
    pb2_step = BMODE == OP_ROW ? (int64_t)GEMM_BK : (int64_t)GEMM_BK * g.ldb;
  }
  auto issue_aa2 = [&](int slot, int qp) {
    const unsigned dst = smem_base + slot * G2_OPER_BYTES + piece0 + 2 * qp * 1024;
    glds16_x2(pa2, pa2, voff_a[2 * qp], voff_a[2 * qp + 1], dst, dst + 1024);
  };
  auto issue_bb2 = [&](int slot, int qp) {
    const unsigned dst = smem_base + (3 + slot) * G2_OPER_BYTES + piece0 + 2 * qp * 1024;
    glds16_x2(pb2, pb2, voff_b[2 * qp], voff_b[2 * qp + 1], dst, dst + 1024);
  };
  auto issue_h2 = [&](int stage, int h, int q) {
    const unsigned dst = smem_base + stage * G2_STAGE_BYTES + (unsigned)__builtin_amdgcn_readfirstlane(g2h_piece(w, h, q) * 1024);
    glds16_x2(pa2, pb2, voff_a[2 * h + q], voff_b[2 * h + q], dst, dst + G2_OPER_BYTES);
  };
  bf16x8 fa[2][4], fb[2][2];
  auto frag_a = [&](const char* img, int base32, int ks) { return g2_frag<AMODE>(img, base32, ks, lane); };
  auto frag_b = [&](const char* img, int base32, int ks) { return g2_frag<BMODE>(img, base32, ks, lane); };
  if (kt_begin < kt_end) {
    if constexpr (A3) {
      issue_aa(kt_begin, 0, 0); issue_aa(kt_begin, 0, 1);
      issue_bb(kt_begin, 0, 0); issue_bb(kt_begin, 0, 1);
      issue_bb(clampk(kt_begin + 1), 1, 0); issue_bb(clampk(kt_begin + 1), 1, 1);
      issue_aa(clampk(kt_begin + 1), 1, 0); issue_aa(clampk(kt_begin + 1), 1, 1);
    } else if constexpr (HALF) {
#pragma unroll
      for (int q = 0; q < 4; ++q) issue_h(kt_begin, 0, q >> 1, q & 1);
#pragma unroll
      for (int q = 0; q < 4; ++q) issue_h(clampk(kt_begin + 1), 1, q >> 1, q & 1);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) issue(kt_begin, 0, q);
#pragma unroll
      for (int q = 0; q < 4; ++q) issue(clampk(kt_begin + 1), 1, q);
    }
  }
#ifdef G2X_PROLOGUE_WAIT_ALL
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
  // only tile 0 has to be there (this wave's 8 older DMA instructions); tile 1 keeps landing under tile 0's MFMAs and is
  // waited for at the first hand-over barrier, like every later tile
  if constexpr (HALF) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // the first HALF of tile 0 (4 of this wave's 16 DMA instructions)
  else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#endif
  __syncthreads();
  // (M16) fragments of the 16x16x32 MFMA: 16 rows x 32 k -- lane (r16 = lane % 16, q = lane / 16) reads the 16-byte chunk q + 4 ks of row
  // r16 of its row block; the chunk swizzle (chunk ^ (row >> 1 & 7)) does not depend on the block.  A fragments in a ring of four (block
  // t + 3 requested behind block t's first MFMA), B fragments double-buffered per K step.
  bf16x8 fa16[4], fb16[2][4];
  int offA16[2] = {0, 0}, offB16[2] = {0, 0};
  if constexpr (M16) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int r16 = lane & 15, c = (lane >> 4) + 4 * ks;
      const int o = r16 * 128 + ((c ^ ((r16 >> 1) & 7)) << 4);
      offA16[ks] = o + (128 * wm) * 128;
      offB16[ks] = o + (64 * wn) * 128;
      asm volatile("" : "+v"(offA16[ks]), "+v"(offB16[ks]));
    }
  }
  auto fragA16 = [&](const char* img, int i16, int ks) { return *reinterpret_cast<const bf16x8*>(img + (16 * i16) * 128 + offA16[ks]); };
  auto fragB16 = [&](const char* img, int j16, int ks) { return *reinterpret_cast<const bf16x8*>(img + (16 * j16) * 128 + offB16[ks]); };
  if (kt_begin < kt_end) {
    if constexpr (M16K) {
#pragma unroll
      for (int t3 = 0; t3 < 3; ++t3) fa16[t3] = tr16(smem_base + ka0, smem_base + ka1, t3, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb16[0][j] = tr16(smem_base + kb0, smem_base + kb1, j, 0);
    } else if constexpr (M16) {
#pragma unroll
      for (int t3 = 0; t3 < 3; ++t3) fa16[t3] = fragA16(smem, t3, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb16[0][j] = fragB16(smem + (A3 ? 3 : 1) * G2_OPER_BYTES, j, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[0][i] = frag_a(smem, 128 * wm + 32 * i, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[0][j] = frag_b(smem + (A3 ? 3 : 1) * G2_OPER_BYTES, 64 * wn + 32 * j, 0);
    }
  }

#ifdef G2X_STAMP
  st1 = __builtin_readcyclecounter();
#endif
  int cur = 0;
#ifdef G2X_WAITSTAMP
  unsigned long long g2x_wait_vm = 0, g2x_wait_bar = 0;
#endif
  int aslot = 0;        // A3: the A slot of tile kt
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int anext = aslot == 2 ? 0 : aslot + 1, aprev = anext == 2 ? 0 : anext + 1;
    const char* sa = A3 ? smem + aslot * G2_OPER_BYTES : smem + cur * G2_STAGE_BYTES;
    const char* sb = A3 ? smem + (3 + cur) * G2_OPER_BYTES : sa + G2_OPER_BYTES;
    const char* na = A3 ? smem + anext * G2_OPER_BYTES : smem + (cur ^ 1) * G2_STAGE_BYTES;
    const char* nb = A3 ? smem + (3 + (cur ^ 1)) * G2_OPER_BYTES : na + G2_OPER_BYTES;
    // fragment f (0..3: A rows 32 f, 4..5: B columns 32 (f - 4)) of K step `ks` of the tile whose images are (ia, ib)
    auto ldfrag = [&](const char* ia, const char* ib, int ks, int f) {
      if (f < 4) fa[ks & 1][f] = frag_a(ia, 128 * wm + 32 * f, ks);
      else fb[ks & 1][f - 4] = frag_b(ib, 64 * wn + 32 * (f - 4), ks);
    };
    if constexpr (M16K) {
      // half h of the tile = MFMAs 32 h .. 32 h + 31: m = 4 t + j, A block t against B block j.  A block t + 3 behind block t's first MFMA (t + 3 >= 8:
      // the next half's, behind the hand-over); the hand-over H_u at m = 18 (block 7 was requested at m = 16; half u + 1 has landed: all but this
      // wave's 8 youngest DMA instructions); behind it the next half's four B fragments and half u + 4's two DMA pairs into the freed slot.
      const unsigned cb = smem_base + (unsigned)cur * G2_STAGE_BYTES, nx = smem_base + (unsigned)(cur ^ 1) * G2_STAGE_BYTES;
      const unsigned a0c = cb + ka0, a1c = cb + ka1, b0c = cb + kb0, b1c = cb + kb1;
      const unsigned a0n = nx + ka0, a1n = nx + ka1, b0n = nx + kb0, b1n = nx + kb1;
      static_for<0, 64>([&](auto n_tag) {
        constexpr int n = decltype(n_tag)::value, h = n >> 5, m = n & 31, t16 = m >> 2, j = m & 3;
        asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc16[t16][j]) : "v"(fb16[h][j]), "v"(fa16[t16 & 3]));   // D^T = B A^T
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (j == 0) {
          constexpr int tt = t16 + 3;
          if constexpr (tt < 8) fa16[tt & 3] = tr16(a0c, a1c, tt, h);
          else if constexpr (h == 0) fa16[tt & 3] = tr16(a0c, a1c, tt - 8, 1);
          else fa16[tt & 3] = tr16(a0n, a1n, tt - 8, 0);
        }
        if constexpr (m == 18) {
          asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
          __syncthreads();
        }
        if constexpr (m == 19 || m == 21 || m == 22 || m == 23) {
          constexpr int jb = m == 19 ? 0 : m - 20;
          if constexpr (h == 0) fb16[1][jb] = tr16(b0c, b1c, jb, 1);
          else fb16[0][jb] = tr16(b0n, b1n, jb, 0);
        }
        if constexpr (m == 26) { if constexpr (RUN2) issue_h2(cur, h, 0); else issue_h(clampk(kt + 2), cur, h, 0); }
        if constexpr (m == 30) { if constexpr (RUN2) issue_h2(cur, h, 1); else issue_h(clampk(kt + 2), cur, h, 1); }
        __builtin_amdgcn_sched_barrier(0);
      });
    } else if constexpr (M16) {
      // MFMA n = 4 t + j: row block t % 8 of K step t / 8 against column block j.  Side work, one piece per MFMA: A block t + 3 at j = 0
      // (t + 3 >= 16: the next tile's, behind the hand-over), K step 1's B fragments at n = 5..17, DMA pieces at n = 1 / 10 (A3: A tile
      // kt + 2 into the slot tile kt - 1 left; else the second half of tile kt + 1), the hand-over at n = 50 (row block 15 was requested at
      // n = 48: every read of this tile's images is issued), behind it tile kt + 2's other pieces and the next tile's K-step-0 B fragments.
      static_for<0, 64>([&](auto n_tag) {
        constexpr int n = decltype(n_tag)::value, t16 = n >> 2, ks = t16 >> 3, i16 = t16 & 7, j = n & 3;
        // (the accumulator TIED to the destination: see gemm256s.h -- no MFMA here depends on one closer than 31 instructions before it)
        asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc16[i16][j]) : "v"(fb16[ks][j]), "v"(fa16[t16 & 3]));   // D^T = B A^T
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (j == 0) {
          constexpr int tt = t16 + 3;
          if constexpr (tt < 16) fa16[tt & 3] = fragA16(sa, tt & 7, tt >> 3);
          else fa16[tt & 3] = fragA16(na, tt - 16, 0);
        }
        if constexpr (n == 5 || n == 9 || n == 13 || n == 17) fb16[1][(n - 5) >> 2] = fragB16(sb, (n - 5) >> 2, 1);
        if constexpr (n == 1) { if constexpr (RUN2) issue_aa2(aprev, 0); else if constexpr (A3) issue_aa(clampk(kt + 2), aprev, 0); else issue(clampk(kt + 1), cur ^ 1, 2); }
        if constexpr (n == 10) { if constexpr (RUN2) issue_aa2(aprev, 1); else if constexpr (A3) issue_aa(clampk(kt + 2), aprev, 1); else issue(clampk(kt + 1), cur ^ 1, 3); }
        if constexpr (n == 50) {
          if constexpr (A3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // all but A tile kt+2 (this wave's 4 youngest DMA instructions)
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
        }
        if constexpr (n == 51) { if constexpr (RUN2) issue_bb2(cur, 0); else if constexpr (A3) issue_bb(clampk(kt + 2), cur, 0); else issue(clampk(kt + 2), cur, 0); }
        if constexpr (n == 53 || n == 54 || n == 57 || n == 58) {
          constexpr int jb = n == 53 ? 0 : n == 54 ? 1 : n == 57 ? 2 : 3;
          fb16[0][jb] = fragB16(nb, jb, 0);
        }
        if constexpr (n == 59) { if constexpr (RUN2) issue_bb2(cur, 1); else if constexpr (A3) issue_bb(clampk(kt + 2), cur, 1); else issue(clampk(kt + 2), cur, 1); }
        __builtin_amdgcn_sched_barrier(0);
      });
    } else {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
#ifdef G2_PROXY16
        // TIMING PROXY ONLY (kbench -DG2_PROXY16; results WRONG): each 32x32x16 MFMA as two 16x16x32 on the same operand registers and a
        // quarter each of the accumulator -- what would the tile-form K loops (weight gradient, decoder) gain from the other shape?
        {
          typedef __attribute__((ext_vector_type(4))) float f32x4_;
          f32x16& C = acc[m >> 1][m & 1];
          f32x4_ q0 = (kk & 1) ? f32x4_{C[4], C[5], C[6], C[7]} : f32x4_{C[0], C[1], C[2], C[3]};
          f32x4_ q1 = (kk & 1) ? f32x4_{C[12], C[13], C[14], C[15]} : f32x4_{C[8], C[9], C[10], C[11]};
          q0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk & 1][m & 1], fa[kk & 1][m >> 1], q0, 0, 0, 0);
          q1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk & 1][m & 1], fa[kk & 1][m >> 1], q1, 0, 0, 0);
          if (kk & 1) { C[4] = q0[0]; C[5] = q0[1]; C[6] = q0[2]; C[7] = q0[3]; C[12] = q1[0]; C[13] = q1[1]; C[14] = q1[2]; C[15] = q1[3]; }
          else { C[0] = q0[0]; C[1] = q0[1]; C[2] = q0[2]; C[3] = q0[3]; C[8] = q1[0]; C[9] = q1[1]; C[10] = q1[2]; C[11] = q1[3]; }
        }
#else
        acc[m >> 1][m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kk & 1][m & 1], fa[kk & 1][m >> 1], acc[m >> 1][m & 1], 0, 0, 0);   // D^T = B A^T
#endif
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (HALF) {
          // half u = (kt, kk / 2).  Even K step: the fragments of the odd one.  Odd K step: hand-over H_u -- half u + 1 has
          // landed (this wave: all but its 8 youngest DMA instructions, the halves u + 2 and u + 3), every fragment of
          // half u is in registers -- then half u + 4 into the freed slot and the first fragments of half u + 1.
          if ((kk & 1) == 0) {
            if (m < 6) ldfrag(sa, sb, kk + 1, m);
          } else {
            const char* ia = kk == 1 ? sa : na;
            const char* ib = kk == 1 ? sb : nb;
            const int ks = kk == 1 ? 2 : 0;
            if (m == 1) {
#ifdef G2X_WAITSTAMP
              const unsigned long long w0 = __builtin_readcyclecounter();
              asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
              const unsigned long long w1 = __builtin_readcyclecounter();
              __syncthreads();
              const unsigned long long w2 = __builtin_readcyclecounter();
              g2x_wait_vm += w1 - w0; g2x_wait_bar += w2 - w1;
#else
              asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
              __syncthreads();
#endif
            } else if (m == 2) {
              issue_h(clampk(kt + 2), cur, kk >> 1, 0);
            } else if (m >= 3 && m <= 5) {
              ldfrag(ia, ib, ks, 2 * (m - 3));
              ldfrag(ia, ib, ks, 2 * (m - 3) + 1);
            } else if (m == 6) {
              issue_h(clampk(kt + 2), cur, kk >> 1, 1);
            }
          }
        } else
        if (kk == 0) {            // second half of tile kt+1's pieces (A3: A tile kt+2) + the fragments of K step 1
          if (m == 0) { if constexpr (A3) issue_aa(clampk(kt + 2), aprev, 0); else issue(clampk(kt + 1), cur ^ 1, 2); }
          else if (m == 3) { if constexpr (A3) issue_aa(clampk(kt + 2), aprev, 1); else issue(clampk(kt + 1), cur ^ 1, 3); }
          else ldfrag(sa, sb, 1, m < 3 ? m - 1 : m - 2);
        } else if (kk < 3) {      // the fragments of K step kk + 1
          if (m < 6) ldfrag(sa, sb, kk + 1, m);
        } else {                  // hand-over, first half of tile kt+2's pieces, tile kt+1's first fragments
          if (m == 1) {
#ifdef G2X_WAITSTAMP
            const unsigned long long w0 = __builtin_readcyclecounter();
            if constexpr (A3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long w1 = __builtin_readcyclecounter();
            __syncthreads();
            const unsigned long long w2 = __builtin_readcyclecounter();
            g2x_wait_vm += w1 - w0; g2x_wait_bar += w2 - w1;
#else
            if constexpr (A3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // all but A tile kt+2 (this wave's 4 youngest DMA instructions)
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile kt+1 have landed
            __syncthreads();                                    // B_kt: ... and everybody's; all reads of this stage are done
#endif
          } else if (m == 2) {
            if constexpr (A3) issue_bb(clampk(kt + 2), cur, 0); else
            issue(clampk(kt + 2), cur, 0);
          } else if (m == 3) {
            ldfrag(na, nb, 0, 0);
            ldfrag(na, nb, 0, 1);
          } else if (m == 4) {
            ldfrag(na, nb, 0, 2);
            ldfrag(na, nb, 0, 3);
          } else if (m == 5) {
            ldfrag(na, nb, 0, 4);
            ldfrag(na, nb, 0, 5);
          } else if (m == 6) {
            if constexpr (A3) issue_bb(clampk(kt + 2), cur, 1); else
            issue(clampk(kt + 2), cur, 1);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    }      // (!M16)
    if constexpr (RUN2) {      // -> the origins of tile min(kt + 3, last)
      const int kn = kt + 3;
      if (kn <= kt_last) {
        const bool seg = kn == ktiles0;
        pa2 = seg ? pa2_seg1 : pa2 + pa2_step;
        pb2 = seg ? pb2_seg1 : pb2 + pb2_step;
      }
    }
    cur ^= 1;
    aslot = anext;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // retire the trailing re-copies before LDS is reused
  __syncthreads();

#ifdef G2X_STAMP
  st2 = __builtin_readcyclecounter();
#endif
  if constexpr (M16 || M16K) {
    if constexpr (epi_rounds_first<Epi>::value && epi_wide8<Epi>::value) g2_epilogue_bf16_w8<PERSIST>(acc16, smem, bm, bn, split, epi);
    else if constexpr (epi_rounds_first<Epi>::value) g2_epilogue_bf16<PERSIST>(acc16, smem, bm, bn, split, epi);
    else g2_epilogue(acc16, smem, bm, bn, split, epi);
  } else {
    if constexpr (epi_rounds_first<Epi>::value && epi_wide8<Epi>::value) g2_epilogue_bf16_w8<PERSIST>(acc, smem, bm, bn, split, epi);
    else if constexpr (epi_rounds_first<Epi>::value) g2_epilogue_bf16<PERSIST>(acc, smem, bm, bn, split, epi);
    else g2_epilogue(acc, smem, bm, bn, split, epi);      // (both end with a barrier: LDS is free again)
  }
#ifdef G2X_STAMP
  if (threadIdx.x == 0 && g2x_stamps) {
    unsigned long long st3 = __builtin_readcyclecounter();
    unsigned long long* o = g2x_stamps + 4 * (size_t)blk;
    o[0] = st1 - st0; o[1] = st2 - st1; o[2] = st3 - st2; o[3] = st0;
#ifdef G2X_WAITSTAMP
    o[0] = g2x_wait_vm; o[2] = g2x_wait_bar;      // (wave 0's waits at the hand-over barriers, summed over the K tiles)
#endif
  }
#endif
}

// PERSIST = false: one tile per workgroup (grid = the static tile count; with GemmArgs::dyn the workgroups beyond the
// dynamic extent exit at once).  PERSIST = true (dyn launches sized by a host-side ESTIMATE of the extent, e.g. the AuxK
// GEMMs while few latents are dead: thousands of workgroups that start only to exit cost ~0.3 ms per launch, each needs a
// CU's whole LDS before it can do so): a workgroup walks the tiles blockIdx.x, blockIdx.x + gridDim.x, ... -- correct for
// any grid size (gridDim.x a multiple of 8 keeps a workgroup's tiles on its XCD under xcd_remap).  Kept a separate
// instantiation: the tile loop costs the K loop ~10 % (register allocation across the back edge).
template <int AMODE, int BMODE, class Epi, bool PERSIST = false>
__global__ __launch_bounds__(512, 2) void gemm256_bf16_kernel(GemmArgs g, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int nbm, nbn, ktiles0, ktiles, splits;
  if (!gemm_dyn_dims(g, 2, nbm, nbn, ktiles0, ktiles, splits)) return;
  const int nblk = nbm * nbn * splits;
  if constexpr (PERSIST) {
    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) gemm256_tile<AMODE, BMODE, true>(g, epi, smem, blk, nbm, nbn, ktiles0, ktiles, splits);
  } else {
    const int nlaunch = g.tail_tiles > 0 ? nbm * nbn + g.tail_tiles * (g.tail_pieces - 1) : nblk;
    if ((int)blockIdx.x < nlaunch) gemm256_tile<AMODE, BMODE, false>(g, epi, smem, blockIdx.x, nbm, nbn, ktiles0, ktiles, splits);
  }
}
