// Streaming form of the 256x256x64 bf16 GEMM (gemm256.h) for the K = d_model GEMMs of the generic paths -- encoder
// c = relu(x W + b) (l1autoencoder.py:74), dpre = (dx_hat W + 1/M) [c > 0] (its autograd), TopK encoder (topkautoencoder.py:72-77):
// row-major x row-major, one K segment, tens of thousands of output tiles of only 12-20 K tiles each.
//
// What gemm256.h pays per OUTPUT tile at these shapes (tools/kbench -DG2X_STAMP, K = 1280): prologue 5.9 % (the first two K tiles
// arrive by LDS-DMA while nothing computes), epilogue 11.4 % (all eight waves park their accumulators in ONE 133 KB LDS tile
// behind three workgroup barriers; the stages are gone, so the next tile cannot be requested) -- 17 % of a tile with the matrix
// pipe idle, 26 % at K = 768.  Here a workgroup walks its tiles as ONE continuous stream of K tiles:
//   * the hand-over slots that would re-copy "a K tile past the end" request the NEXT output tile's K tiles 0 and 1 instead, so
//     when the last MFMA of a tile has issued the next tile's first stage has landed and its first fragments are in registers:
//     no prologue after the first tile;
//   * the two stages stay where they are (2 x 64 KiB); the epilogue is WAVE-PRIVATE in the CU's remaining 32 KiB: each wave
//     transposes its own 128x64 accumulator block through its own 4 KiB, one 32x64 row block at a time (8 ds_write_b64 +
//     4 ds_read_b128 per block, XOR-swizzled: conflict-free both ways), and leaves as full 128-byte lines (8 lanes x 16 B per
//     row).  No workgroup barrier, no wave waits for another; the LDS pipe executes one wave's reads and writes in order, so
//     block i + 1 is written right behind block i's reads without a wait in between;
//   * the functor's loads (bias; the latent for the dpre gate) are requested at the START of the tile's last K tile: a memory
//     latency under 32 MFMAs instead of in front of the first store.
// Functors opt in with `static constexpr bool STREAM = true` and the s_* interface below; per-lane coordinates are fixed for the
// whole kernel (lane l owns columns 8 (l % 8) .. + 7 of its wave's 64 and rows l / 8 + 8 q + 32 i of its 128), so per-column
// state (bias, column sums) lives in registers.  Same arithmetic per element as the tile form; reductions over a tile are taken
// per wave in a fixed order (deterministic; fp32 sums associate differently from the tile form's).
//
//   s_begin()                               once per workgroup
//   s_tile(row0, col)                       at the start of a tile's LAST K tile: per-tile loads (this lane's 8 columns from `col`)
//   SPre s_prefetch(row, col)               the global loads of one (row, 8 columns) element, issued one row block ahead
//   s_rows()                                M (rows >= M are forced to zero / left out by the PARTIAL form of s_apply)
//   s_apply<PARTIAL>(row, col, v0, v1, pre) v = the bf16-rounded accumulators as floats; PARTIAL: the wave's block crosses row M
//   s_tile_end(row0_wave, col)              wave-level reductions of the tile (all 64 lanes call)
//   s_end(scratch)                          once per workgroup: block-level reductions (all 512 threads call; 64 floats of LDS)
#pragma once
#include "gemm256.h"

constexpr int G2S_EPI_BYTES = 8 * 4096;                                // 4 KiB per wave
constexpr int G2S_LDS_BYTES = 2 * G2_STAGE_BYTES + G2S_EPI_BYTES;      // 163 840: the CU's whole LDS

template <class E, class = void>
struct epi_stream { static constexpr bool value = false; };
template <class E>
struct epi_stream<E, std::void_t<decltype(E::STREAM)>> { static constexpr bool value = E::STREAM; };

#ifndef G2S_PRIO
#define G2S_PRIO 0           // tools/kbench experiment (see the K loop)
#endif
#ifndef G2S_STATIC
#define G2S_STATIC 1         // round 6: the K loop unrolled over the two LDS stages (compile-time stage; K tiles per output tile must be EVEN:
#endif                       // the host launches the streaming form only then).  0 = round 5's loop with a run-time stage
#ifndef G2S_RUNPTR
#define G2S_RUNPTR 1          // the 16x16x32 loop carries the source origins of the stream's K tiles kt + 1 / kt + 2 instead of forming them (a select and
                              // a 64-bit add per operand and DMA gap, twice per K tile, in both waves of a SIMD at once)
#endif
#ifndef G2S_M16
#define G2S_M16 (G2S_STATIC)  // round 6: the K loop on v_mfma_f32_16x16x32_bf16 (needs the static-stage form).  The chip is power-managed and holds a higher
#endif                        // clock on this shape: stand-alone, random operands, K = 1280: 5.50 -> 4.97 ms as a timing proxy (profiles/r06_kbench_proxy16.txt)
#ifndef G2_STREAM
#define G2_STREAM 1          // tools/build_variant.sh A/B switch: 0 = the K = d GEMMs through gemm256.h's persistent tile form
#endif

// The wave's 128x64 accumulator block -> functor, through the wave's own 4 KiB of LDS (see the header).
//   write: lane (r = lane % 32, h = lane / 32) holds, per (j, g), columns 32 j + 8 g + 4 h .. + 3 of row 32 i + r: 8-byte half
//          h ^ (r >> 3 & 1) of 16-byte chunk (4 j + g) ^ (r & 7) of LDS row r (128-byte rows);
//   read:  lane (rr = lane / 8, c = lane % 8), q = 0..3: row rr + 8 q, logical chunk c = physical chunk c ^ rr, its halves swapped
//          when q is odd (a compile-time register rename).
template <bool PARTIAL, class Epi>
__device__ __forceinline__ void g2s_epilogue(f32x16 (&acc)[4][2], char* eb, int row_w, int col_l, typename Epi::SPre (&pre0)[4], Epi& epi) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, rr = lane >> 3, c = lane & 7;
  char* wbase = eb + r * 128 + ((h ^ ((r >> 3) & 1)) << 3);
  const char* rbase = eb + rr * 128 + ((c ^ rr) << 4);
  const int sw = r & 7;
  auto write_block = [&](int i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<bf16x4*>(wbase + (((4 * j + g) ^ sw) << 4)) =
            bf16x4{(bf16_t)acc[i][j][4 * g], (bf16_t)acc[i][j][4 * g + 1], (bf16_t)acc[i][j][4 * g + 2], (bf16_t)acc[i][j][4 * g + 3]};
  };
  typename Epi::SPre pre[2][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) pre[0][q] = pre0[q];
  write_block(0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    bf16x8 raw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) raw[q] = *reinterpret_cast<const bf16x8*>(rbase + q * 1024);
    if (i + 1 < 4) {
      // (the LDS executes a wave's operations in order: these writes land behind the reads above without a wait)
      write_block(i + 1);
#pragma unroll
      for (int q = 0; q < 4; ++q) pre[(i + 1) & 1][q] = epi.s_prefetch(row_w + 32 * (i + 1) + 8 * q + rr, col_l);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bf16x8 v = raw[q];
      const int lo = (q & 1) ? 4 : 0, hi = 4 - lo;
      const f32x4 v0 = {(float)v[lo], (float)v[lo + 1], (float)v[lo + 2], (float)v[lo + 3]};
      const f32x4 v1 = {(float)v[hi], (float)v[hi + 1], (float)v[hi + 2], (float)v[hi + 3]};
      epi.template s_apply<PARTIAL>(row_w + 32 * i + 8 * q + rr, col_l, v0, v1, pre[i & 1][q]);
    }
  }
  epi.s_tile_end(row_w, col_l);
}

// The same for functors that need the fp32 accumulators (static constexpr bool STREAM_F32: the TopK encoder adds its bias BEFORE the
// single rounding to bf16, topkautoencoder.py:75 under autocast): one 32x32 MFMA tile (4 KiB of fp32) at a time.
//   write: lane (r, h), g = 0..3: columns 8 g + 4 h .. + 3 of row r = 16-byte chunk 2 g + h, stored at chunk (2 g + h) ^ f(r),
//          f(r) = (r & 7) ^ (r >> 2 & 1) (a permutation of 0..7 over 8 consecutive rows: the 8-lane groups of ds_write_b128 are
//          conflict-free; the 16-lane groups of the reads below -- rows {0,3,5,6} / {1,2,4,7} + 8 k, every second chunk -- too);
//   read:  lane (rq = lane / 4, cp = lane % 4), q = 0, 1: row rq + 16 q, columns 8 cp .. + 7 = chunks 2 cp and 2 cp + 1.
// The functor sees s_apply(e = 4 i + 2 j + q, row, col, v0, v1) with col = the lane's 8 columns of the 32-column block j.
template <class Epi>
__device__ __forceinline__ void g2s_epilogue_f32(f32x16 (&acc)[4][2], char* eb, int row_w, int col_w, Epi& epi) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5, rq = lane >> 2, cp = lane & 3;
  auto fsw = [](int row) { return (row & 7) ^ ((row >> 2) & 1); };
  char* wbase = eb + r * 128;
  const int wsw = fsw(r);
  const char* rbase = eb + rq * 128;
  const int rsw = fsw(rq);           // (row rq + 16: the same low bits)
  auto write_block = [&](int b) {
    const int i = b >> 1, j = b & 1;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<f32x4*>(wbase + (((2 * g + h) ^ wsw) << 4)) =
          f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
  };
  write_block(0);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    f32x4 raw[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      raw[q][0] = *reinterpret_cast<const f32x4*>(rbase + q * 2048 + (((2 * cp) ^ rsw) << 4));
      raw[q][1] = *reinterpret_cast<const f32x4*>(rbase + q * 2048 + (((2 * cp + 1) ^ rsw) << 4));
    }
    if (b + 1 < 8) write_block(b + 1);
    const int i = b >> 1, j = b & 1;
#pragma unroll
    for (int q = 0; q < 2; ++q)
      epi.s_apply(4 * i + 2 * j + q, row_w + 32 * i + 16 * q + rq, col_w + 32 * j + 8 * cp, raw[q][0], raw[q][1]);
  }
  epi.s_tile_end(row_w, col_w);
}

// ---- the same two epilogues for the accumulators of the 16x16x32 K loop (G2S_M16): acc[i16][j16] = one 16x16 tile as f32x4 -- lane
// (r16 = lane % 16, q = lane / 16) holds columns 16 j16 + 4 q .. + 3 of row 16 i16 + r16 (operands swapped as in the 32x32 loop: a lane
// owns consecutive COLUMNS).  The LDS images the functor-side reads expect are the same; only who writes which 8 (16) bytes changes:
//   bf16 form: row rr = 16 (i16 & 1) + r16 of the 32-row block, 16-byte chunk 2 j16 + (q >> 1), 8-byte half q & 1;
//   fp32 form: row rr of the 32x32 tile, 16-byte chunk 4 (j16 & 1) + q.
typedef __attribute__((ext_vector_type(4))) float g2s_f32x4;
template <bool PARTIAL, class Epi>
__device__ __forceinline__ void g2s_epilogue16(g2s_f32x4 (&acc)[8][4], char* eb, int row_w, int col_l, typename Epi::SPre (&pre0)[4], Epi& epi) {
  const int lane = threadIdx.x & 63, r16 = lane & 15, q4 = lane >> 4, rr = lane >> 3, c = lane & 7;
  const char* rbase = eb + rr * 128 + ((c ^ rr) << 4);
  auto write_block = [&](int I) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int row = 16 * a + r16;
      char* wb = eb + row * 128 + (((q4 & 1) ^ ((row >> 3) & 1)) << 3);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const g2s_f32x4 v = acc[2 * I + a][j];
        *reinterpret_cast<bf16x4*>(wb + (((2 * j + (q4 >> 1)) ^ (row & 7)) << 4)) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
      }
    }
  };
  typename Epi::SPre pre[2][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) pre[0][q] = pre0[q];
  write_block(0);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    bf16x8 raw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) raw[q] = *reinterpret_cast<const bf16x8*>(rbase + q * 1024);
    if (i + 1 < 4) {
      write_block(i + 1);
#pragma unroll
      for (int q = 0; q < 4; ++q) pre[(i + 1) & 1][q] = epi.s_prefetch(row_w + 32 * (i + 1) + 8 * q + rr, col_l);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bf16x8 v = raw[q];
      const int lo = (q & 1) ? 4 : 0, hi = 4 - lo;
      const f32x4 v0 = {(float)v[lo], (float)v[lo + 1], (float)v[lo + 2], (float)v[lo + 3]};
      const f32x4 v1 = {(float)v[hi], (float)v[hi + 1], (float)v[hi + 2], (float)v[hi + 3]};
      epi.template s_apply<PARTIAL>(row_w + 32 * i + 8 * q + rr, col_l, v0, v1, pre[i & 1][q]);
    }
  }
  epi.s_tile_end(row_w, col_l);
}

template <class Epi>
__device__ __forceinline__ void g2s_epilogue16_f32(g2s_f32x4 (&acc)[8][4], char* eb, int row_w, int col_w, Epi& epi) {
  const int lane = threadIdx.x & 63, r16 = lane & 15, q4 = lane >> 4, rq = lane >> 2, cp = lane & 3;
  auto fsw = [](int row) { return (row & 7) ^ ((row >> 2) & 1); };
  const char* rbase = eb + rq * 128;
  const int rsw = fsw(rq);           // (row rq + 16: the same low bits)
  auto write_block = [&](int b) {
    const int i = b >> 1, j = b & 1;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int row = 16 * a + r16;
#pragma unroll
      for (int bq = 0; bq < 2; ++bq) {
        const g2s_f32x4 v = acc[2 * i + a][2 * j + bq];
        *reinterpret_cast<f32x4*>(eb + row * 128 + (((4 * bq + q4) ^ fsw(row)) << 4)) = f32x4{v[0], v[1], v[2], v[3]};
      }
    }
  };
  write_block(0);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    f32x4 raw[2][2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      raw[q][0] = *reinterpret_cast<const f32x4*>(rbase + q * 2048 + (((2 * cp) ^ rsw) << 4));
      raw[q][1] = *reinterpret_cast<const f32x4*>(rbase + q * 2048 + (((2 * cp + 1) ^ rsw) << 4));
    }
    if (b + 1 < 8) write_block(b + 1);
    const int i = b >> 1, j = b & 1;
#pragma unroll
    for (int q = 0; q < 2; ++q)
      epi.s_apply(4 * i + 2 * j + q, row_w + 32 * i + 16 * q + rq, col_w + 32 * j + 8 * cp, raw[q][0], raw[q][1]);
  }
  epi.s_tile_end(row_w, col_w);
}

template <class E, class = void>
struct epi_stream_f32 { static constexpr bool value = false; };
template <class E>
struct epi_stream_f32<E, std::void_t<decltype(E::STREAM_F32)>> { static constexpr bool value = E::STREAM_F32; };

template <class Epi>
__global__ __launch_bounds__(512, 2) void gemm256s_bf16_kernel(GemmArgs g, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 2, wn = w & 3;
  const int nbm = g.nbm, nbn = g.nbn, nk = g.ktiles, ntiles = nbm * nbn;
  const int group_m = g.group_m > 0 ? g.group_m : GEMM_GROUP_M;
  epi.s_begin();
  int blk = blockIdx.x;
  if (blk < ntiles) {
    int bm, bn;
    tile_coords(xcd_remap(blk, ntiles), nbm, nbn, bm, bn, group_m);
    const bf16_t* a_cur = g.A0 + (int64_t)(bm * G2_BM) * g.lda;
    const bf16_t* b_cur = g.B0 + (int64_t)(bn * G2_BN) * g.ldb;

    unsigned voff[4];          // (lda == ldb: both operands are [rows][d_p] -- the host launches this form only then)
#pragma unroll
    for (int q = 0; q < 4; ++q) voff[q] = g2_src_off<OP_ROW>(4 * w + q, lane, g.lda);
    typedef __attribute__((address_space(3))) char* lptr_t;
    const unsigned smem_base = (unsigned)(uintptr_t)(lptr_t)smem;
    const unsigned piece0 = (unsigned)__builtin_amdgcn_readfirstlane(4 * w * 1024);
#if G2S_STATIC
    // Round 6 (the same finding as in bwd_fused.h): with the stage a run-time variable every K tile rebuilt its LDS addresses -- 32 vector
    // additions (stage base + lane offset in front of the fragment reads, whose 16-bit immediate cannot reach the second stage) and ~35
    // scalar instructions (stage bases, DMA destinations) per wave, 2 waves per SIMD.  LDS layout of this form: [A stage 0 | A stage 1 |
    // B stage 0 | B stage 1] (4 x 32 KiB): both stages of an operand are within the immediate's reach of ONE set of four lane offsets
    // (one per K step: the chunk swizzle XORs the K step into the lane's row bits, so it is not an additive constant).
    auto issue = [&](const bf16_t* pa, const bf16_t* pb, int stage, int q) {
      const unsigned dst = smem_base + stage * G2_OPER_BYTES + piece0 + q * 1024;
      glds16_x2(pa, pb, voff[q], voff[q], dst, dst + 2 * G2_OPER_BYTES);
    };
    int offA[4], offB[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int r = lane & 31, c = 2 * ks + (lane >> 5);
      const int o = r * 128 + ((c ^ ((r >> 1) & 7)) << 4);            // frag_read<OP_ROW>'s offset for base32 = 0
      offA[ks] = o + (128 * wm) * 128;
      offB[ks] = o + (64 * wn) * 128 + 2 * G2_OPER_BYTES;
      asm volatile("" : "+v"(offA[ks]), "+v"(offB[ks]));             // opaque: kept as registers, not re-derived per read
    }
    auto fragA = [&](int stage, int i, int ks) { return *reinterpret_cast<const bf16x8*>(smem + stage * G2_OPER_BYTES + (32 * i) * 128 + offA[ks]); };
    auto fragB = [&](int stage, int j, int ks) { return *reinterpret_cast<const bf16x8*>(smem + stage * G2_OPER_BYTES + (32 * j) * 128 + offB[ks]); };
#if G2S_M16
    // fragments of v_mfma_f32_16x16x32_bf16: 16 rows x 32 k -- lane (r16 = lane % 16, q = lane / 16) reads the 16-byte chunk q + 4 ks of row
    // r16 of its row block.  The images and their chunk swizzle (chunk ^ (row >> 1 & 7)) are the 32x32 loop's; tools/lds_bank_check.py-style
    // count for the four 16-lane service groups of ds_read_b128: rows {0-3, 12-15} with chunk c and rows 4-11 with chunk c + 1 hit 16
    // distinct (row parity, chunk) slots for c = 0 and c = 4 -- conflict-free as it stands.
    int offA16[2], offB16[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int r16 = lane & 15, c = (lane >> 4) + 4 * ks;
      const int o = r16 * 128 + ((c ^ ((r16 >> 1) & 7)) << 4);
      offA16[ks] = o + (128 * wm) * 128;
      offB16[ks] = o + (64 * wn) * 128 + 2 * G2_OPER_BYTES;
      asm volatile("" : "+v"(offA16[ks]), "+v"(offB16[ks]));
    }
    auto fragA16 = [&](int stage, int i16, int ks) { return *reinterpret_cast<const bf16x8*>(smem + stage * G2_OPER_BYTES + (16 * i16) * 128 + offA16[ks]); };
    auto fragB16 = [&](int stage, int j16, int ks) { return *reinterpret_cast<const bf16x8*>(smem + stage * G2_OPER_BYTES + (16 * j16) * 128 + offB16[ks]); };
#endif
#else
    auto issue = [&](const bf16_t* pa, const bf16_t* pb, int stage, int q) {
      const unsigned dst = smem_base + stage * G2_STAGE_BYTES + piece0 + q * 1024;
      glds16_x2(pa, pb, voff[q], voff[q], dst, dst + G2_OPER_BYTES);
    };
#endif
    char* eb = smem + 2 * G2_STAGE_BYTES + w * 4096;
    const int rr = lane >> 3, c8 = 8 * (lane & 7);

    // the first tile's K tiles 0 and 1 (nk >= 2: the host launches this form for K >= 128 only)
#pragma unroll
    for (int q = 0; q < 4; ++q) issue(a_cur, b_cur, 0, q);
#pragma unroll
    for (int q = 0; q < 4; ++q) issue(a_cur + GEMM_BK, b_cur + GEMM_BK, 1, q);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __syncthreads();
    bf16x8 fa[2][4], fb[2][2];
#if G2S_M16
    // A fragments: a ring of four (row block t = 8 ks + i16 of the K tile in slot t % 4, requested three row blocks ahead); B fragments:
    // the four column blocks of a K step, double-buffered by K step
    bf16x8 fa16[4], fb16[2][4];
    g2s_f32x4 acc16[8][4];
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3) fa16[t3] = fragA16(0, t3, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb16[0][j] = fragB16(0, j, 0);
#elif G2S_STATIC
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[0][i] = fragA(0, i, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[0][j] = fragB(0, j, 0);
#else
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[0][i] = frag_read<OP_ROW>(smem, 128 * wm + 32 * i, 0, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[0][j] = frag_read<OP_ROW>(smem + G2_OPER_BYTES, 64 * wn + 32 * j, 0, lane);
#endif

    // (no zeroing: the first K step of every tile multiplies into a ZERO source -- an inline constant of the MFMA -- instead of into
    // the accumulators: 128 vector moves per wave and tile less in an epilogue that is bound by its vector instructions)
    f32x16 acc[4][2];
    int cur = 0;
#ifdef G2X_STAMP
    unsigned long long st_loop = 0, st_epi = 0, st_tiles = 0;
    const unsigned long long st_k0 = __builtin_readcyclecounter();
#endif
#if G2S_M16 && G2S_RUNPTR
    // origins of the stream's K tiles kt + 1 and kt + 2 (nk == 2: tile 2 is the next output tile's first -- set at the top of the tile loop)
    const bf16_t* q1a = a_cur + GEMM_BK;
    const bf16_t* q1b = b_cur + GEMM_BK;
    const bf16_t* q2a = a_cur + 2 * GEMM_BK;
    const bf16_t* q2b = b_cur + 2 * GEMM_BK;
#endif
    for (;;) {
      // the tile behind this one in the workgroup's walk (none: the stream ends by re-copying this tile's last K tile, harmless)
      const int nblk = blk + gridDim.x;
      const bool more = nblk < ntiles;
      int bm2 = bm, bn2 = bn;
      if (more) tile_coords(xcd_remap(nblk, ntiles), nbm, nbn, bm2, bn2, group_m);
      const bf16_t* a_nxt = more ? g.A0 + (int64_t)(bm2 * G2_BM) * g.lda : a_cur + (int64_t)(nk - 2) * GEMM_BK;
      const bf16_t* b_nxt = more ? g.B0 + (int64_t)(bn2 * G2_BN) * g.ldb : b_cur + (int64_t)(nk - 2) * GEMM_BK;
      // K tile j of the stream as seen from this tile: j < nk this tile's, j >= nk the next tile's j - nk
      auto pa = [&](int j) { return j < nk ? a_cur + (int64_t)j * GEMM_BK : a_nxt + (int64_t)(j - nk) * GEMM_BK; };
      auto pb = [&](int j) { return j < nk ? b_cur + (int64_t)j * GEMM_BK : b_nxt + (int64_t)(j - nk) * GEMM_BK; };
#if G2S_M16 && G2S_RUNPTR
      if (nk == 2) { q2a = a_nxt; q2b = b_nxt; }      // (two K tiles per output tile: stream tile 2 is the NEXT output tile's first, known only now)
#endif
      const int row_w = bm * G2_BM + 128 * wm, col_l = bn * G2_BN + 64 * wn + c8;
      typename Epi::SPre pre0[4];
#ifdef G2X_STAMP
      const unsigned long long st0 = __builtin_readcyclecounter();
#endif

      // one K tile of the stream; FIRST: the tile's K tile 0 (its first K step starts the accumulators from zero)
#if G2S_M16
      // one K tile on the 16x16x32 shape: 64 MFMAs n = 4 t + j (t = 8 ks + i16: K step and 16-row block, j: 16-column block), one piece of
      // other work behind each -- the A fragment of row block t + 3 at j == 0 (blocks 13-15 request the NEXT tile's blocks 0-2 from the
      // other stage: after the hand-over), K step 1's B fragments at n = 5..17, the second half of tile kt+1's DMA pieces at n = 1 / 10,
      // the hand-over at n = 50 (every read of this stage is issued by then: row block 15 was requested at n = 48), behind it the first half
      // of tile kt+2's pieces and the next tile's K-step-0 B fragments.
      auto ktile16 = [&](auto first_tag, auto cur_tag, int kt) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int CUR = decltype(cur_tag)::value;
        if (!FIRST && kt == nk - 1) {      // the functor's loads for this tile: a memory latency under the last K tile's MFMAs
          if constexpr (epi_stream_f32<Epi>::value) {
            epi.s_tile(row_w, bn * G2_BN + 64 * wn + 8 * (lane & 3));
          } else {
            epi.s_tile(row_w, col_l);
#pragma unroll
            for (int q = 0; q < 4; ++q) pre0[q] = epi.s_prefetch(row_w + 8 * q + rr, col_l);
          }
        }
        static_for<0, 64>([&](auto n_tag) {
          constexpr int n = decltype(n_tag)::value, t = n >> 2, ks = t >> 3, i16 = t & 7, j = n & 3;
          // (inline asm with the accumulator TIED to the destination: through the builtin, whose destination may differ from its source,
          // the allocator renamed the accumulators between the two peeled K tiles of an output tile and the loop -- both copies live across
          // the transition, 250 registers instead of ~200, and the real functors spilled their prefetched operands.  No MFMA of this loop
          // depends on an MFMA closer than 31 instructions before it, and the epilogue reads a block's accumulators hundreds of cycles
          // after their last update: the hazards the compiler no longer sees cannot occur.)
          if constexpr (FIRST && ks == 0) {
            asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc16[i16][j]) : "v"(fb16[0][j]), "v"(fa16[t & 3]));
          } else {
            asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc16[i16][j]) : "v"(fb16[ks][j]), "v"(fa16[t & 3]));   // D^T = B A^T
          }
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (j == 0) {
            constexpr int tt = t + 3;
            if constexpr (tt < 16) fa16[tt & 3] = fragA16(CUR, tt & 7, tt >> 3);
            else fa16[tt & 3] = fragA16(CUR ^ 1, tt - 16, 0);
          }
          if constexpr (n == 5 || n == 9 || n == 13 || n == 17) fb16[1][(n - 5) >> 2] = fragB16(CUR, (n - 5) >> 2, 1);
#if G2S_RUNPTR
          if constexpr (n == 1) issue(q1a, q1b, CUR ^ 1, 2);
          if constexpr (n == 10) issue(q1a, q1b, CUR ^ 1, 3);
#else
          if constexpr (n == 1) issue(pa(kt + 1), pb(kt + 1), CUR ^ 1, 2);
          if constexpr (n == 10) issue(pa(kt + 1), pb(kt + 1), CUR ^ 1, 3);
#endif
          if constexpr (n == 50) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
          }
#if G2S_RUNPTR
          if constexpr (n == 51) issue(q2a, q2b, CUR, 0);
#else
          if constexpr (n == 51) issue(pa(kt + 2), pb(kt + 2), CUR, 0);
#endif
          if constexpr (n == 53 || n == 54 || n == 57 || n == 58) fb16[0][n == 53 ? 0 : n == 54 ? 1 : n == 57 ? 2 : 3] = fragB16(CUR ^ 1, n == 53 ? 0 : n == 54 ? 1 : n == 57 ? 2 : 3, 0);
#if G2S_RUNPTR
          if constexpr (n == 59) issue(q2a, q2b, CUR, 1);
#else
          if constexpr (n == 59) issue(pa(kt + 2), pb(kt + 2), CUR, 1);
#endif
          __builtin_amdgcn_sched_barrier(0);
        });
#if G2S_RUNPTR
        {   // the stream's tile kt + 3: this output tile's, or the first of the next one's (its second and third follow by + GEMM_BK)
          const bool wrap = kt + 3 == nk;
          q1a = q2a; q1b = q2b;
          q2a = wrap ? a_nxt : q2a + GEMM_BK;
          q2b = wrap ? b_nxt : q2b + GEMM_BK;
        }
#endif
      };
#endif
#if G2S_STATIC
      auto ktile = [&](auto first_tag, auto cur_tag, int kt) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int cur = decltype(cur_tag)::value;        // (shadows the run-time `cur`, which this form does not use)
        constexpr int sa = cur, sb = cur, na = cur ^ 1, nb = cur ^ 1;      // "images" are stage numbers here
        auto ldfrag = [&](int ia, int ib, int ks, int f) {
          if (f < 4) fa[ks & 1][f] = fragA(ia, f, ks);
          else fb[ks & 1][f - 4] = fragB(ib, f - 4, ks);
        };
#else
      auto ktile = [&](auto first_tag, int kt) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const char* sa = smem + cur * G2_STAGE_BYTES;
        const char* sb = sa + G2_OPER_BYTES;
        const char* na = smem + (cur ^ 1) * G2_STAGE_BYTES;
        const char* nb = na + G2_OPER_BYTES;
        auto ldfrag = [&](const char* ia, const char* ib, int ks, int f) {
          if (f < 4) fa[ks & 1][f] = frag_read<OP_ROW>(ia, 128 * wm + 32 * f, ks, lane);
          else fb[ks & 1][f - 4] = frag_read<OP_ROW>(ib, 64 * wn + 32 * (f - 4), ks, lane);
        };
#endif
        if (!FIRST && kt == nk - 1) {      // the functor's loads for this tile: a memory latency under the last K tile's MFMAs
#if G2S_PRIO
          // experiment: waves 0-3 run the last K tile's MFMAs ahead of their SIMD partners (waves 4-7), so that each half's
          // epilogue -- vector instructions, LDS, stores -- falls beside the other half's matrix work instead of beside its epilogue
          if (w < 4) __builtin_amdgcn_s_setprio(3);
#endif
          if constexpr (epi_stream_f32<Epi>::value) {
            epi.s_tile(row_w, bn * G2_BN + 64 * wn + 8 * (lane & 3));
          } else {
            epi.s_tile(row_w, col_l);
#pragma unroll
            for (int q = 0; q < 4; ++q) pre0[q] = epi.s_prefetch(row_w + 8 * q + rr, col_l);
          }
        }
        // (the schedule of gemm256.h's K loop: one piece of other work behind each MFMA)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
          for (int m = 0; m < 8; ++m) {
#ifdef G2S_PROXY16
            // TIMING PROXY ONLY (tools/build_variant.sh g2sproxy16 -DG2S_PROXY16; results are WRONG): every 32x32x16 MFMA replaced by two
            // 16x16x32 MFMAs on the same operand registers and a quarter each of the same accumulator -- the same FLOPs, LDS and register
            // traffic, the other MFMA shape: what would a 16x16x32 K loop get under the chip's power limit with two waves per SIMD?
            {
              typedef __attribute__((ext_vector_type(4))) float f32x4_;
              f32x16& C = acc[m >> 1][m & 1];
              f32x4_ q0 = (kk & 1) ? f32x4_{C[4], C[5], C[6], C[7]} : f32x4_{C[0], C[1], C[2], C[3]};
              f32x4_ q1 = (kk & 1) ? f32x4_{C[12], C[13], C[14], C[15]} : f32x4_{C[8], C[9], C[10], C[11]};
              if (FIRST && kk == 0) { q0 = f32x4_{0.f, 0.f, 0.f, 0.f}; q1 = q0; }
              q0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk & 1][m & 1], fa[kk & 1][m >> 1], q0, 0, 0, 0);
              q1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk & 1][m & 1], fa[kk & 1][m >> 1], q1, 0, 0, 0);
              if (kk & 1) { C[4] = q0[0]; C[5] = q0[1]; C[6] = q0[2]; C[7] = q0[3]; C[12] = q1[0]; C[13] = q1[1]; C[14] = q1[2]; C[15] = q1[3]; }
              else { C[0] = q0[0]; C[1] = q0[1]; C[2] = q0[2]; C[3] = q0[3]; C[8] = q1[0]; C[9] = q1[1]; C[10] = q1[2]; C[11] = q1[3]; }
            }
            if constexpr (false) {
              if (kk == 0) {
              } else {
              }
            }
#else
            if constexpr (FIRST) {
              if (kk == 0) {
                f32x16 zero;
#pragma unroll
                for (int e = 0; e < 16; ++e) zero[e] = 0.f;
                acc[m >> 1][m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][m & 1], fa[0][m >> 1], zero, 0, 0, 0);
              } else {
                acc[m >> 1][m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kk & 1][m & 1], fa[kk & 1][m >> 1], acc[m >> 1][m & 1], 0, 0, 0);
              }
            } else {
              acc[m >> 1][m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kk & 1][m & 1], fa[kk & 1][m >> 1], acc[m >> 1][m & 1], 0, 0, 0);   // D^T = B A^T
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
            if (kk == 0) {            // second half of stream tile kt+1's pieces + the fragments of K step 1
              if (m == 0) issue(pa(kt + 1), pb(kt + 1), cur ^ 1, 2);
              else if (m == 3) issue(pa(kt + 1), pb(kt + 1), cur ^ 1, 3);
              else ldfrag(sa, sb, 1, m < 3 ? m - 1 : m - 2);
            } else if (kk < 3) {
              if (m < 6) ldfrag(sa, sb, kk + 1, m);
            } else {                  // hand-over, first half of stream tile kt+2's pieces, stream tile kt+1's first fragments
              if (m == 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
              } else if (m == 2) {
                issue(pa(kt + 2), pb(kt + 2), cur, 0);
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
                issue(pa(kt + 2), pb(kt + 2), cur, 1);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#if !G2S_STATIC
        cur ^= 1;
#endif
      };
#if G2S_STATIC
      {   // nk is even (host-side condition of this form): every output tile starts in stage 0
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
#if G2S_M16
        ktile16(std::true_type{}, S0{}, 0);
        ktile16(std::false_type{}, S1{}, 1);
        for (int kt = 2; kt < nk; kt += 2) {
          ktile16(std::false_type{}, S0{}, kt);
          ktile16(std::false_type{}, S1{}, kt + 1);
        }
#else
        ktile(std::true_type{}, S0{}, 0);
        ktile(std::false_type{}, S1{}, 1);
        for (int kt = 2; kt < nk; kt += 2) {
          ktile(std::false_type{}, S0{}, kt);
          ktile(std::false_type{}, S1{}, kt + 1);
        }
#endif
      }
#else
      ktile(std::true_type{}, 0);
      for (int kt = 1; kt < nk; ++kt) ktile(std::false_type{}, kt);
#endif
#if G2S_PRIO
      __builtin_amdgcn_s_setprio(0);
#endif
#ifdef G2X_STAMP
      const unsigned long long st1 = __builtin_readcyclecounter();
#endif

      const bool partial = row_w + 128 > epi.s_rows();       // (wave-uniform) rows beyond M in this wave's block: the masking form
#if G2S_M16
      if constexpr (epi_stream_f32<Epi>::value) g2s_epilogue16_f32(acc16, eb, row_w, bn * G2_BN + 64 * wn, epi);
      else if (partial) g2s_epilogue16<true>(acc16, eb, row_w, col_l, pre0, epi);
      else g2s_epilogue16<false>(acc16, eb, row_w, col_l, pre0, epi);
#else
      if constexpr (epi_stream_f32<Epi>::value) g2s_epilogue_f32(acc, eb, row_w, bn * G2_BN + 64 * wn, epi);
      else if (partial) g2s_epilogue<true>(acc, eb, row_w, col_l, pre0, epi);
      else g2s_epilogue<false>(acc, eb, row_w, col_l, pre0, epi);
#endif
#ifdef G2X_STAMP
      st_loop += st1 - st0; st_epi += __builtin_readcyclecounter() - st1; ++st_tiles;
#endif
      if (!more) break;
      blk = nblk; bm = bm2; bn = bn2;
      a_cur = a_nxt; b_cur = b_nxt;
    }
#ifdef G2X_STAMP
    if (threadIdx.x == 0 && g2x_stamps) {
      unsigned long long* o = g2x_stamps + 4 * (size_t)blockIdx.x;
      o[0] = st_tiles; o[1] = st_loop; o[2] = st_epi; o[3] = __builtin_readcyclecounter() - st_k0;
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing re-copies must have landed before the CU's LDS changes hands
  }
  __syncthreads();
  epi.s_end(reinterpret_cast<float*>(smem + 2 * G2_STAGE_BYTES));
}
