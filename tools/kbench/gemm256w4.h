// 256x256x64 bf16 MFMA GEMM for gfx950, K-contiguous operands (OP_ROW x OP_ROW: the encoder, decoder and dpre products
// of the d >= 768 paths), ONE wave per SIMD:
//
//   256 threads = 4 waves in a 2 (M) x 2 (N) arrangement, each wave a 128x128 output = 8x8 v_mfma_f32_16x16x32_bf16
//   blocks (256 accumulator registers).  Against the 8-wave kernel of gemm256.h (128x64 per wave, 32x32x16 MFMAs,
//   LDS-DMA staging) per 64-deep K tile and CU:
//     fragment reads   128 KiB instead of 192 KiB   (0.25 ds_read_b128 per 16-cycle MFMA instead of 0.75 per 32-cycle one),
//     staging          16 global_load_dwordx4 + 16 ds_write_b128 per wave (~20 issue cycles per KiB) instead of 8
//                      LDS-DMA pieces per wave at 60-185 issue cycles each -- tools/kbench: the DMA issue and the fragment
//                      reads, not the MFMAs, bounded that kernel,
//     the 16x16x32 shape holds a higher clock under the chip's power management than 32x32x16 (1.98 vs 1.68 GHz in a bare
//     loop, tools/mfma_clock_probe.hip).
//   The operands are swapped in the MFMA (D^T = B A^T), so a lane holds FOUR CONSECUTIVE COLUMNS of one output row: the
//   accumulators go to the epilogue's fp32 LDS tile as 16-byte writes.
//
// K loop: a tile is four panels (k32 step ks = 0,1  x  row half p = 0,1) of 32 MFMAs, walked column by column (4 rows per
// column), with at most one piece of other work in the slot behind each MFMA:
//   * A fragments (4 per panel) are double-buffered by panel; the next panel's are read during this one;
//   * B fragments (8 per k32 step) are single-buffered: in a p = 1 panel, B[c] of the next k32 step is requested right
//     after column c's last MFMA and is first used a whole panel (28 MFMAs) later;
//   * staging: the registers of piece q are written to LDS, and re-loaded from global memory with the piece of the tile
//     after that, four pieces per panel: a global load has a full tile (~2000 cycles) to land.
// Tile t lives in stage t & 1.  One barrier per tile, between panels 2 and 3: by then tile t+1 is complete in the other
// stage (panel 3 reads its first fragments from there) and nobody reads stage t & 1 any more (panel 3 starts refilling it
// with tile t+2).
#pragma once
#include <type_traits>

#include "../../freud_amd/csrc/gemm256.h"

// fragment of v_mfma_f32_16x16x32_bf16: 16 rows from base16 of a [256][128 B] swizzled image, k32 step ks;
// lane -> row base16 + lane % 16, 16-byte chunk 4 ks + lane / 16
__device__ __forceinline__ bf16x8 g4_frag(const char* img, int base16, int ks, int lane) {
  const int r = base16 + (lane & 15), c = 4 * ks + (lane >> 4);
  return *reinterpret_cast<const bf16x8*>(img + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
}

// Epilogue: the tile leaves as two passes (column halves), each pass two 128x128 sub-tiles (the two waves of that column
// half) through fp32 LDS and the row-major functor; the functors' block reductions are 256-thread, i.e. workgroup wide.
template <class Epi>
__device__ __forceinline__ void g4_epilogue(f32x4 (&acc)[8][8], char* smem, int bm, int bn, int split, Epi& epi) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
  float* tile = reinterpret_cast<float*>(smem);
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    if (wn == pass) {
      float* dst = tile + wm * G2_SUB_FLOATS + (lane & 15) * GEMM_EPI_PITCH + 4 * (lane >> 4);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(dst + 16 * i * GEMM_EPI_PITCH + 16 * j) = acc[i][j];
    }
    __syncthreads();
#pragma unroll 1
    for (int sub = 0; sub < 2; ++sub) {
      const float* src = tile + sub * G2_SUB_FLOATS;
      const int row0 = bm * G2_BM + 128 * sub, col0 = bn * G2_BN + 128 * pass;
      epi.tile_begin(row0, col0, split);
      {
        const int c4 = (t & 31) * 4;
        typename Epi::Pre pre[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) pre[it] = epi.prefetch(row0 + (t >> 5) + 8 * it, col0 + c4);
#pragma unroll
        for (int it = 0; it < 16; ++it) {
          const int row = (t >> 5) + 8 * it;
          const f32x4 v = *reinterpret_cast<const f32x4*>(&src[row * GEMM_EPI_PITCH + c4]);
          epi.apply(row0 + row, col0 + c4, v, pre[it]);
        }
      }
      __syncthreads();
      epi.tile_end(tile + sub * G2_SUB_FLOATS);
      __syncthreads();
    }
  }
}

// g.nbm / g.nbn count 256-wide tiles.  Both operands OP_ROW (K contiguous).
template <class Epi>
__global__ __launch_bounds__(256, 1) void gemm256w4_bf16_kernel(GemmArgs g, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;

  const int nblk = g.nbm * g.nbn * g.splits;
  int id = xcd_remap(blockIdx.x, nblk);
  const int split = id / (g.nbm * g.nbn);
  id -= split * (g.nbm * g.nbn);
  int bm, bn;
  tile_coords(id, g.nbm, g.nbn, bm, bn);
  const int ktiles = (g.seg1_gate != nullptr && *g.seg1_gate == 0) ? g.ktiles0 : g.ktiles;
  const int kt_begin = (int)((int64_t)ktiles * split / g.splits);
  const int kt_end = (int)((int64_t)ktiles * (split + 1) / g.splits);

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // staging: piece q (0..7: A, 8..15: B) = rows 32 (q & 7) .. + 31 of the operand tile; thread -> row t / 8, 16-byte chunk t % 8
  const int srow = t >> 3, sch = t & 7;
  const int kt_last = kt_end - 1;
  auto clampk = [&](int kt) { return kt < kt_last ? kt : kt_last; };   // past-the-end tiles re-copy the last one (harmless)
  // uniform (SGPR) piece origin + one 32-bit per-thread byte offset per operand: saddr-form global loads, no 64-bit
  // per-piece pointers in VGPRs
  const unsigned voff_a = (unsigned)((srow * g.lda + sch * 8) * 2), voff_b = (unsigned)((srow * g.ldb + sch * 8) * 2);
  auto src_ptr = [&](int kt, int q) -> const u32x4* {
    const bool s1 = kt >= g.ktiles0;
    const int k = (s1 ? kt - g.ktiles0 : kt) * GEMM_BK;
    if (q < 8) {
      const char* ub = reinterpret_cast<const char*>((s1 ? g.A1 : g.A0) + (int64_t)(bm * G2_BM + 32 * q) * g.lda + k);
      return reinterpret_cast<const u32x4*>(ub + voff_a);
    }
    const char* ub = reinterpret_cast<const char*>((s1 ? g.B1 : g.B0) + (int64_t)(bn * G2_BN + 32 * (q - 8)) * g.ldb + k);
    return reinterpret_cast<const u32x4*>(ub + voff_b);
  };
  const int st_off = srow * 128 + ((sch ^ ((srow >> 1) & 7)) << 4);     // (32 q + srow) >> 1 & 7 == srow >> 1 & 7
  // Two register sets: piece q of tile T lives in stg[T & 1][q] from its global load (issued right after the same registers
  // were written to LDS with tile T-2's piece) to its LDS write: two tiles of lead time, 32 KiB in flight per wave --
  // with one set (one tile of lead) the loop waited on memory latency (tools/kbench: 5.7 ms against 4.2 ms without the
  // loads at M=65536, N=1280, K=40960).  Tile parity is relative to kt_begin; the K loop is unrolled by two tiles.
  u32x4 stg[2][16];
  auto gload = [&](int set, int kt, int q) { stg[set][q] = *src_ptr(kt, q); };
  auto lstore = [&](int set, int stage, int q) {
    *reinterpret_cast<u32x4*>(smem + stage * G2_STAGE_BYTES + (q >> 3) * G2_OPER_BYTES + (q & 7) * 4096 + st_off) = stg[set][q];
  };

  bf16x8 fa[2][4], fb[8];
  if (kt_begin < kt_end) {
#pragma unroll
    for (int q = 0; q < 16; ++q) gload(0, kt_begin, q);
#pragma unroll
    for (int q = 0; q < 16; ++q) gload(1, clampk(kt_begin + 1), q);
#pragma unroll
    for (int q = 0; q < 16; ++q) lstore(0, 0, q);
#pragma unroll
    for (int q = 0; q < 16; ++q) gload(0, clampk(kt_begin + 2), q);
    // "panel 3 of tile -1": pieces 0..5 of tile 1 -> stage 1, their registers refilled from tile 3
#pragma unroll
    for (int q = 0; q < 6; ++q) lstore(1, 1, q);
#pragma unroll
    for (int q = 0; q < 6; ++q) gload(1, clampk(kt_begin + 3), q);
  }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (kt_begin < kt_end) {
#pragma unroll
    for (int f = 0; f < 4; ++f) fa[0][f] = g4_frag(smem, 128 * wm + 16 * f, 0, lane);
#pragma unroll
    for (int c = 0; c < 8; ++c) fb[c] = g4_frag(smem + G2_OPER_BYTES, 128 * wn + 16 * c, 0, lane);
  }

  // one tile; PAR = its parity (stage and register set).  Slot plan of a panel (slot = 4 c + r behind MFMA (row r, column c)):
  //   c = 0: r = 0..2, c = 1: r = 0   the next panel's 4 A fragments (first thing in the panel: a full panel of lead)
  //   r = 3, p = 1 panels             B[c] of the next k32 step
  //   c = 1..6: r = 1 / r = 2         LDS write of a staging piece / global load that refills its registers:
  //                                   panel 3: pieces 0..5 of tile kt+2 (into this stage, free after the barrier),
  //                                   panel 0: pieces 6..10, panel 1: pieces 11..15 of tile kt+1 (other stage); none in panel 2,
  //                                   so the barrier's wait for this wave's LDS writes is short
  auto tile_body = [&](int kt, auto par_tag) {
    constexpr int PAR = decltype(par_tag)::value;
    const char* sa = smem + PAR * G2_STAGE_BYTES;
    const char* sb = sa + G2_OPER_BYTES;
    const char* na = smem + (PAR ^ 1) * G2_STAGE_BYTES;
    const char* nb = na + G2_OPER_BYTES;
#pragma unroll
    for (int pn = 0; pn < 4; ++pn) {
      const int p = pn & 1;
      const int npn = (pn + 1) & 3, nks = npn >> 1, np = npn & 1;       // the panel after this one
#pragma unroll
      for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 4 * p + r;
          // (asm with an AGPR constraint: left to itself hipcc moves accumulator blocks between AGPRs and VGPRs across the
          // back edge, ~600 v_accvgpr moves per tile)
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][c]) : "v"(fb[c]), "v"(fa[pn & 1][r]));
          __builtin_amdgcn_sched_barrier(0);
          const int af = c == 0 ? r : 3;
          if ((c == 0 && r < 3) || (c == 1 && r == 0)) {
#ifndef G4X_NOFRAG
            fa[(pn + 1) & 1][af] = g4_frag(pn == 3 ? na : sa, 128 * wm + 64 * np + 16 * af, nks, lane);
#endif
          } else if (r == 3 && p == 1) {
#ifndef G4X_NOFRAG
            fb[c] = g4_frag(pn == 3 ? nb : sb, 128 * wn + 16 * c, pn == 3 ? 0 : 1, lane);
#endif
          } else if ((r == 1 || r == 2) && c >= 1 && c <= 6 && pn != 2) {
            const int wi = c - 1;
            const int q = pn == 3 ? wi : (pn == 0 ? 6 + wi : 11 + wi);
            if (pn == 3 || wi < 5) {
              if (r == 1) {
#ifndef G4X_NOLSTORE
                lstore(pn == 3 ? PAR : PAR ^ 1, pn == 3 ? PAR : PAR ^ 1, q);
#endif
              } else {
#ifndef G4X_NOGLOAD
                gload(pn == 3 ? PAR : PAR ^ 1, clampk(pn == 3 ? kt + 4 : kt + 3), q);
#endif
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#ifndef G4X_NOBAR
      if (pn == 2) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // my writes of tile kt+1 are in LDS (and my reads of this stage done)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
    }
  };
  // (the launcher guarantees an even number of K tiles per split)
  for (int kt = kt_begin; kt < kt_end; kt += 2) {
    tile_body(kt, std::integral_constant<int, 0>{});
    tile_body(kt + 1, std::integral_constant<int, 1>{});
  }
  __syncthreads();     // trailing re-copies retired (vmcnt(0)) and written before LDS is reused

  g4_epilogue(acc, smem, bm, bn, split, epi);
}

// ---- the same kernel on v_mfma_f32_32x32x16_bf16 (4x4 blocks per wave): 64 MFMAs of 32 cycles per tile, so every gap has
// 24 spare issue cycles (8 with the 16-cycle shape) for its one piece of other work.
// Panel = k16 step ks (0..3) x row half p: 8 MFMAs, column by column (2 rows per column); slot s = 2 c + r.
//   s = 0, 2          : the next panel's two A fragments
//   s = 1, 3, 5, 7    : (p = 1) B[c] of the next k16 step
//   the other slots   : staging, alternately the LDS write of a piece and the global load that refills its registers;
//                       panels 0..6 move pieces 1..15 of tile kt+1 (other stage), panel 7 piece 0 of tile kt+2 (this stage,
//                       free after the barrier at the end of panel 6).
__host__ __device__ constexpr int g4m_free_rank(int pn, int s) {      // rank of slot s among the panel's staging slots, or -1
  if ((pn & 1) == 0) return s == 1 ? 0 : (s >= 3 ? s - 2 : -1);
  return s == 4 ? 0 : (s == 6 ? 1 : -1);
}
__host__ __device__ constexpr int g4m_free_base(int pn) {
  int b = 0;
  for (int j = 0; j < pn; ++j) b += (j & 1) ? 2 : 6;
  return b;
}

template <class Epi>
__device__ __forceinline__ void g4m_epilogue(f32x16 (&acc)[4][4], char* smem, int bm, int bn, int split, Epi& epi) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
  float* tile = reinterpret_cast<float*>(smem);
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    if (wn == pass) {
      float* dst = tile + wm * G2_SUB_FLOATS + (lane & 31) * GEMM_EPI_PITCH + 4 * (lane >> 5);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq)
            *reinterpret_cast<f32x4*>(dst + 32 * i * GEMM_EPI_PITCH + 32 * j + 8 * gq) =
                f32x4{acc[i][j][4 * gq], acc[i][j][4 * gq + 1], acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]};
    }
    __syncthreads();
#pragma unroll 1
    for (int sub = 0; sub < 2; ++sub) {
      const float* src = tile + sub * G2_SUB_FLOATS;
      const int row0 = bm * G2_BM + 128 * sub, col0 = bn * G2_BN + 128 * pass;
      epi.tile_begin(row0, col0, split);
      {
        const int c4 = (t & 31) * 4;
        typename Epi::Pre pre[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) pre[it] = epi.prefetch(row0 + (t >> 5) + 8 * it, col0 + c4);
#pragma unroll
        for (int it = 0; it < 16; ++it) {
          const int row = (t >> 5) + 8 * it;
          const f32x4 v = *reinterpret_cast<const f32x4*>(&src[row * GEMM_EPI_PITCH + c4]);
          epi.apply(row0 + row, col0 + c4, v, pre[it]);
        }
      }
      __syncthreads();
      epi.tile_end(tile + sub * G2_SUB_FLOATS);
      __syncthreads();
    }
  }
}

template <class Epi>
__global__ __launch_bounds__(256, 1) void gemm256w4m_bf16_kernel(GemmArgs g, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;

  const int nblk = g.nbm * g.nbn * g.splits;
  int id = xcd_remap(blockIdx.x, nblk);
  const int split = id / (g.nbm * g.nbn);
  id -= split * (g.nbm * g.nbn);
  int bm, bn;
  tile_coords(id, g.nbm, g.nbn, bm, bn);
  const int ktiles = (g.seg1_gate != nullptr && *g.seg1_gate == 0) ? g.ktiles0 : g.ktiles;
  const int kt_begin = (int)((int64_t)ktiles * split / g.splits);
  const int kt_end = (int)((int64_t)ktiles * (split + 1) / g.splits);

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int srow = t >> 3, sch = t & 7;
  const int kt_last = kt_end - 1;
  auto clampk = [&](int kt) { return kt < kt_last ? kt : kt_last; };
  const unsigned voff_a = (unsigned)((srow * g.lda + sch * 8) * 2), voff_b = (unsigned)((srow * g.ldb + sch * 8) * 2);
  auto src_ptr = [&](int kt, int q) -> const u32x4* {
    const bool s1 = kt >= g.ktiles0;
    const int k = (s1 ? kt - g.ktiles0 : kt) * GEMM_BK;
    if (q < 8) {
      const char* ub = reinterpret_cast<const char*>((s1 ? g.A1 : g.A0) + (int64_t)(bm * G2_BM + 32 * q) * g.lda + k);
      return reinterpret_cast<const u32x4*>(ub + voff_a);
    }
    const char* ub = reinterpret_cast<const char*>((s1 ? g.B1 : g.B0) + (int64_t)(bn * G2_BN + 32 * (q - 8)) * g.ldb + k);
    return reinterpret_cast<const u32x4*>(ub + voff_b);
  };
  const int st_off = srow * 128 + ((sch ^ ((srow >> 1) & 7)) << 4);
  u32x4 stg[2][16];
  auto gload = [&](int set, int kt, int q) { stg[set][q] = *src_ptr(kt, q); };
  auto lstore = [&](int set, int stage, int q) {
    *reinterpret_cast<u32x4*>(smem + stage * G2_STAGE_BYTES + (q >> 3) * G2_OPER_BYTES + (q & 7) * 4096 + st_off) = stg[set][q];
  };

  bf16x8 fa[2][2], fb[4];
  if (kt_begin < kt_end) {
#pragma unroll
    for (int q = 0; q < 16; ++q) gload(0, kt_begin, q);
#pragma unroll
    for (int q = 0; q < 16; ++q) gload(1, clampk(kt_begin + 1), q);
#pragma unroll
    for (int q = 0; q < 16; ++q) lstore(0, 0, q);
#pragma unroll
    for (int q = 0; q < 16; ++q) gload(0, clampk(kt_begin + 2), q);
    lstore(1, 1, 0);                              // "panel 7 of tile -1": piece 0 of tile 1
    gload(1, clampk(kt_begin + 3), 0);
  }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  if (kt_begin < kt_end) {
#pragma unroll
    for (int f = 0; f < 2; ++f) fa[0][f] = frag_read<OP_ROW>(smem, 128 * wm + 32 * f, 0, lane);
#pragma unroll
    for (int c = 0; c < 4; ++c) fb[c] = frag_read<OP_ROW>(smem + G2_OPER_BYTES, 128 * wn + 32 * c, 0, lane);
  }

  auto tile_body = [&](int kt, auto par_tag) {
    constexpr int PAR = decltype(par_tag)::value;
    const char* sa = smem + PAR * G2_STAGE_BYTES;
    const char* sb = sa + G2_OPER_BYTES;
    const char* na = smem + (PAR ^ 1) * G2_STAGE_BYTES;
    const char* nb = na + G2_OPER_BYTES;
#pragma unroll
    for (int pn = 0; pn < 8; ++pn) {
      const int p = pn & 1;
      const int npn = (pn + 1) & 7, nks = npn >> 1, np = npn & 1;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int i = 2 * p + r, s = 2 * c + r;
          asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i][c]) : "v"(fb[c]), "v"(fa[pn & 1][r]));
          __builtin_amdgcn_sched_barrier(0);
          if (s == 0 || s == 2) {
            fa[(pn + 1) & 1][s >> 1] = frag_read<OP_ROW>(pn == 7 ? na : sa, 128 * wm + 64 * np + 32 * (s >> 1), nks, lane);
          } else if (p == 1 && r == 1) {
            fb[c] = frag_read<OP_ROW>(pn == 7 ? nb : sb, 128 * wn + 32 * c, pn == 7 ? 0 : (pn >> 1) + 1, lane);
          } else {
            const int rank = g4m_free_rank(pn, s);
            if (rank >= 0) {
              const int kk = g4m_free_base(pn) + rank;
              const int q = pn == 7 ? 0 : 1 + (kk >> 1);
              const bool is_load = pn == 7 ? rank == 1 : (kk & 1) == 1;
              if (!is_load) {
#ifndef G4X_NOLSTORE
                lstore(pn == 7 ? PAR : PAR ^ 1, pn == 7 ? PAR : PAR ^ 1, q);
#endif
              } else {
#ifndef G4X_NOGLOAD
                gload(pn == 7 ? PAR : PAR ^ 1, clampk(pn == 7 ? kt + 4 : kt + 3), q);
#endif
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (pn == 6) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  for (int kt = kt_begin; kt < kt_end; kt += 2) {
    tile_body(kt, std::integral_constant<int, 0>{});
    tile_body(kt + 1, std::integral_constant<int, 1>{});
  }
  __syncthreads();

  g4m_epilogue(acc, smem, bm, bn, split, epi);
}
