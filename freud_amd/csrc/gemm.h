// 128x128x64 bf16 MFMA GEMM for gfx950 with LDS-staged tiles and a row-major epilogue functor.
//
//   C[128*bm .. , 128*bn ..] = sum_k A(m,k) * B(n,k)            (fp32 accumulate)
//
// Operand storage modes (per operand):
//   OP_ROW    : K-contiguous, P[(row)*ld + k]      -> fragments by ds_read_b128 from an XOR-swizzled
//               [128][64] image (conflict-free for the 4x16-lane groups of ds_read_b128);
//   OP_KMAJOR : K is the slow dimension, P[(k)*ld + row] -> fragments by ds_read_b64_tr_b16 (hardware
//               transposing read) from a [64][128] image with 256-B rows, chunk-XOR swizzled.
// The K range may be the concatenation of two (A,B) pairs (used for the tied-weight gradient
// dW = dx_hat^T c + x^T dpre, one accumulator) and may be split over blockIdx ranges (split-K).
//
// 256 threads = 4 waves in a 2x2 arrangement, each wave a 64x64 output (2x2 v_mfma_f32_32x32x16_bf16
// tiles, 64 accumulator registers); two LDS stages of 32 KiB; register-staged global loads issued one
// K-tile ahead (loads for tile t+1 are in flight while tile t is multiplied).
#pragma once
#include <type_traits>

#include "common.h"

enum { OP_ROW = 0, OP_KMAJOR = 1 };

struct GemmArgs {
  const bf16_t* A0;
  const bf16_t* B0;
  const bf16_t* A1;  // second K segment (may be null)
  const bf16_t* B1;
  int64_t lda, ldb;  // elements
  int nbm, nbn;      // output tiles
  int ktiles0;       // 64-wide K tiles in segment 0
  int ktiles;        // total K tiles (segment 0 + segment 1)
  int splits;        // split-K factor (grid = nbm*nbn*splits)
  const int* seg1_gate;  // device flag (may be null): *seg1_gate == 0 drops K segment 1 (the TopK AuxK pair when no latent is dead)
  // One dimension of the problem may be known only on the device (the number of dead latents of the TopK AuxK branch,
  // padded to 256): dyn points to {count, count_p / 128, count_p / 256, count_p / 64} and dyn_dim says what it replaces --
  // GEMM_DYN_M: the output tile rows (nbm), GEMM_DYN_N: the output tile columns (nbn), GEMM_DYN_K: the K tiles (one
  // segment).  The launch covers the static maximum; workgroups beyond the dynamic extent exit at once (count 0: all).
  const int* dyn;
  int dyn_dim;
  int dyn_splits_min;    // GEMM_DYN_M only, > 0: the split-K factor is chosen ON THE DEVICE from the dynamic extent --
                         // gemm_dyn_splits(tiles, dyn_splits_min, splits, ktiles) with `splits` the maximum the launch and
                         // the slab buffer provide -- so that a small extent (a few output tiles, K = M long) still fills
                         // the chip and the result never depends on a host-side estimate (run-to-run bitwise equal)
  int grid_cover;        // dyn launches, > 0: workgroups that cover every possible extent (host-computed; default = the static
                         // tile count times `splits`)
  int grid_hint;         // dyn launches of the 256x256 kernel: an ESTIMATE of the 128x128 tiles of the dynamic extent (0 = none).
                         // Well below the static maximum it selects the PERSIST instantiation with a grid of that size.
  // 256x256 kernel, static launches with splits == 1 (gemm_tail_plan): the LAST tail_tiles output tiles -- what is left of the
  // tile count after whole rounds of one workgroup per CU -- are computed in tail_pieces K ranges each, so that the last
  // round costs 1 / tail_pieces of a tile's time instead of a whole one.  The functor sees slab id 0 for the other tiles and
  // 1 + piece * tail_tiles + (index of the tail tile) for the pieces; grid = nbm * nbn - tail_tiles + tail_tiles * tail_pieces.
  int tail_tiles, tail_pieces;
  int group_m;           // tile-walk patch height (tile_coords); 0 = GEMM_GROUP_M.  The long-K decoder takes 4 (round 4, same-box A/B
                         // at C4: 4.63-4.68 against 4.80-4.91 ms with 8, the other GEMMs within noise -- profiles/r04_ab_group_m_c4.txt)
};

// The plan for `tiles` 256x256 output tiles of `ktiles` K tiles on 256 CUs: tail tiles and pieces (0, 0: the tile count fills
// whole rounds, or a piece would be shorter than 16 K tiles).
__host__ __device__ __forceinline__ void gemm_tail_plan(int tiles, int ktiles, int& tail_tiles, int& tail_pieces) {
  tail_tiles = tiles % 256;
  tail_pieces = tail_tiles > 0 ? 256 / tail_tiles : 0;
  if (tail_pieces > 16) tail_pieces = 16;
  while (tail_pieces > 1 && ktiles / tail_pieces < 16) --tail_pieces;
  if (tail_pieces < 2) tail_tiles = tail_pieces = 0;
}
enum { GEMM_DYN_NONE = 0, GEMM_DYN_M = 1, GEMM_DYN_N = 2, GEMM_DYN_K = 3 };

// split-K factor for `tiles` output tiles (256 CUs, one workgroup each at a time): the factor in [1, smax] with at least 8 K
// tiles per split that minimises  rounds(tiles * sp) / sp  +  a reduction price proportional to sp * tiles  -- the cost
// model of the host's choose_splits(), evaluated on the device (same inputs -> same factor in the GEMM and in the kernel
// that adds its slabs up).  smin is a floor for callers that want one.
__host__ __device__ __forceinline__ int gemm_dyn_splits(int tiles, int smin, int smax, int ktiles) {
  if (tiles < 1) tiles = 1;
  int best = 1;
  float best_cost = 1e30f;
  for (int sp = 1; sp <= smax; ++sp) {
    if (sp > 1 && ktiles / sp < 8) break;
    const int rounds = (tiles * sp + 255) / 256;
    const float cost = (float)rounds / (float)sp + 0.04f * (float)sp * (float)tiles * (1.0f / 288.0f);
    if (cost < best_cost - 1e-6f) {
      best_cost = cost;
      best = sp;
    }
  }
  return best < smin ? smin : best;
}

// Resolves the device-side dimension; TILE128 = 1 for the 128x128 kernel, 2 for the 256x256 kernels.  False = nothing to do.
__device__ __forceinline__ bool gemm_dyn_dims(const GemmArgs& g, int tile_idx, int& nbm, int& nbn, int& ktiles0, int& ktiles,
                                              int& splits) {
  nbm = g.nbm;
  nbn = g.nbn;
  splits = g.splits;
  ktiles0 = g.ktiles0;
  ktiles = (g.seg1_gate != nullptr && *g.seg1_gate == 0) ? g.ktiles0 : g.ktiles;
  if (g.dyn != nullptr) {
    if (g.dyn[0] <= 0) return false;
    if (g.dyn_dim == GEMM_DYN_M) nbm = g.dyn[tile_idx];
    else if (g.dyn_dim == GEMM_DYN_N) nbn = g.dyn[tile_idx];
    else ktiles0 = ktiles = g.dyn[3];
    if (g.dyn_dim == GEMM_DYN_M && g.dyn_splits_min > 0) splits = gemm_dyn_splits(nbm * nbn, g.dyn_splits_min, g.splits, ktiles);
  }
  return true;
}

constexpr int GEMM_BM = 128, GEMM_BN = 128, GEMM_BK = 64;
constexpr int GEMM_STAGE_BYTES = 32768;              // A 16 KiB + B 16 KiB
constexpr int GEMM_EPI_PITCH = 132;                  // floats
constexpr int GEMM_LDS_BYTES = 128 * GEMM_EPI_PITCH * 4;  // 67584 >= 2 stages

// Bijective XCD-aware remap (blocks b and b+8 share an XCD's L2: give each XCD a contiguous id range).
__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n >> 3, r = n & 7, xcd = id & 7, idx = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Output-tile coordinates of (remapped) workgroup id: tiles are walked in groups of GEMM_GROUP_M tile rows, row index
// fastest, so that the ~32 workgroups an XCD runs at a time form a compact GROUP_M x (32 / GROUP_M) patch and share
// their A and B tiles in that XCD's L2 (a plain row-major walk shares only A).
#ifndef GEMM_GROUP_M_
#define GEMM_GROUP_M_ 8      // tools/build_variant.sh can sweep it (4 x 8 / 8 x 4 / 16 x 2 patches per XCD)
#endif
constexpr int GEMM_GROUP_M = GEMM_GROUP_M_;
__device__ __forceinline__ void tile_coords(int id, int nbm, int nbn, int& bm, int& bn, int group_m = GEMM_GROUP_M) {
  const int per_group = group_m * nbn;
  const int grp = id / per_group, r = id - grp * per_group;
  const int rows = nbm - grp * group_m < group_m ? nbm - grp * group_m : group_m;
  bn = r / rows;
  bm = grp * group_m + (r - bn * rows);
}

template <int MODE>
__device__ __forceinline__ void tile_load(u32x4 (&regs)[4], const bf16_t* base, int64_t ld, int t) {
  if constexpr (MODE == OP_ROW) {
    // [128 rows][64 k]: thread -> row t/8 + 32 i, 16-B chunk t%8
    const bf16_t* p = base + (int64_t)(t >> 3) * ld + (t & 7) * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) regs[i] = *reinterpret_cast<const u32x4*>(p + (int64_t)(32 * i) * ld);
  } else {
    // [64 k][128 cols]: thread -> k row t/16 + 16 i, 16-B chunk t%16
    const bf16_t* p = base + (int64_t)(t >> 4) * ld + (t & 15) * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) regs[i] = *reinterpret_cast<const u32x4*>(p + (int64_t)(16 * i) * ld);
  }
}

template <int MODE>
__device__ __forceinline__ void tile_store(const u32x4 (&regs)[4], char* lds, int t) {
  if constexpr (MODE == OP_ROW) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (t >> 3) + 32 * i, c = t & 7;
      *reinterpret_cast<u32x4*>(lds + r * 128 + ((c ^ ((r >> 1) & 7)) << 4)) = regs[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = (t >> 4) + 16 * i, c = t & 15;
      *reinterpret_cast<u32x4*>(lds + r * 256 + ((c ^ (((r & 3) << 2) | ((r >> 2) & 3))) << 4)) = regs[i];
    }
  }
}

// Fragment of the 32x32x16 MFMA for the 32 output rows/cols starting at `base32` of the tile, k-step kk.
template <int MODE>
__device__ __forceinline__ bf16x8 frag_read(const char* lds, int base32, int kk, int lane) {
  if constexpr (MODE == OP_ROW) {
    const int r = base32 + (lane & 31), c = 2 * kk + (lane >> 5);
    return *reinterpret_cast<const bf16x8*>(lds + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
  } else {
    const int h = lane >> 5, g = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
    const int col = base32 + 16 * g + 4 * p;
    const int r0 = 16 * kk + 8 * h + q, r1 = r0 + 4;
    const int o0 = r0 * 256 + (((col >> 3) ^ (((r0 & 3) << 2) | ((r0 >> 2) & 3))) << 4) + (col & 7) * 2;
    const int o1 = r1 * 256 + (((col >> 3) ^ (((r1 & 3) << 2) | ((r1 >> 2) & 3))) << 4) + (col & 7) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, lds + o0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, lds + o1));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  }
}

// apply() dispatch: a functor that keeps per-call state (EpiTopkEnc's row maxima) offers apply_it(it, ...) with the index of
// the call within the tile -- a compile-time constant in the unrolled epilogue loops, so that state stays in registers.
template <class Epi>
__device__ __forceinline__ auto epi_apply(Epi& e, int it, int row, int col, f32x4 v, const typename Epi::Pre& p, int)
    -> decltype(e.apply_it(it, row, col, v, p)) {
  return e.apply_it(it, row, col, v, p);
}
template <class Epi>
__device__ __forceinline__ void epi_apply(Epi& e, int, int row, int col, f32x4 v, const typename Epi::Pre& p, long) {
  e.apply(row, col, v, p);
}

// How many of a thread's 16 (row, 4-column) elements of a tile have their global loads in flight at once: all 16 unless the
// functor says otherwise (static constexpr int PREFETCH_BATCH).  In the 256x256 kernel half of the waves still hold their
// 128 accumulator registers while the first two sub-tiles are processed, so a functor with a large `Pre` (16 x 4-8
// registers) spilled 30-70 registers to scratch right there.
template <class E, class = void>
struct epi_prefetch_batch { static constexpr int value = 16; };
template <class E>
struct epi_prefetch_batch<E, std::void_t<decltype(E::PREFETCH_BATCH)>> { static constexpr int value = E::PREFETCH_BATCH; };

// Epi requirements:
//   __device__ void tile_begin(int row0, int col0, int split);
//   struct Pre;  __device__ Pre prefetch(int row, int col) const;      // the global loads of apply(), issued early
//   __device__ void apply(int row, int col, f32x4 v, const Pre&);      // 4 consecutive columns
//   __device__ void tile_end(float* lds_scratch);        // block-wide reductions (all 256 threads call)
template <int AMODE, int BMODE, class Epi>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmArgs g, Epi epi) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;

  int nbm, nbn, ktiles0, ktiles, splits;
  if (!gemm_dyn_dims(g, 1, nbm, nbn, ktiles0, ktiles, splits)) return;
  const int nblk = nbm * nbn * splits;
  if ((int)blockIdx.x >= nblk) return;
  int id = xcd_remap(blockIdx.x, nblk);
  const int split = id / (nbm * nbn);
  id -= split * (nbm * nbn);
  int bm, bn;
  tile_coords(id, nbm, nbn, bm, bn, g.group_m > 0 ? g.group_m : GEMM_GROUP_M);
  const int kt_begin = (int)((int64_t)ktiles * split / splits);
  const int kt_end = (int)((int64_t)ktiles * (split + 1) / splits);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto a_ptr = [&](int kt) -> const bf16_t* {
    const bool s1 = kt >= ktiles0;
    const bf16_t* base = s1 ? g.A1 : g.A0;
    const int k = (s1 ? kt - ktiles0 : kt) * GEMM_BK;
    if constexpr (AMODE == OP_ROW) return base + (int64_t)(bm * GEMM_BM) * g.lda + k;
    else return base + (int64_t)k * g.lda + bm * GEMM_BM;
  };
  auto b_ptr = [&](int kt) -> const bf16_t* {
    const bool s1 = kt >= ktiles0;
    const bf16_t* base = s1 ? g.B1 : g.B0;
    const int k = (s1 ? kt - ktiles0 : kt) * GEMM_BK;
    if constexpr (BMODE == OP_ROW) return base + (int64_t)(bn * GEMM_BN) * g.ldb + k;
    else return base + (int64_t)k * g.ldb + bn * GEMM_BN;
  };

  u32x4 ra[4], rb[4];
  if (kt_begin < kt_end) {
    tile_load<AMODE>(ra, a_ptr(kt_begin), g.lda, t);
    tile_load<BMODE>(rb, b_ptr(kt_begin), g.ldb, t);
    tile_store<AMODE>(ra, smem, t);
    tile_store<BMODE>(rb, smem + 16384, t);
  }
  __syncthreads();

  int cur = 0;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const bool more = kt + 1 < kt_end;
    if (more) {
      tile_load<AMODE>(ra, a_ptr(kt + 1), g.lda, t);
      tile_load<BMODE>(rb, b_ptr(kt + 1), g.ldb, t);
    }
    const char* sa = smem + cur * GEMM_STAGE_BYTES;
    const char* sb = sa + 16384;
    // fragments of k-step kk+1 are requested from LDS before the MFMAs of k-step kk issue (hipcc on its own reuses
    // one register set and serialises read -> wait -> MFMA); sched_barrier pins that order
    bf16x8 fa[2][2], fb[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) fa[0][i] = frag_read<AMODE>(sa, 64 * wm + 32 * i, 0, lane);
#pragma unroll
    for (int j = 0; j < 2; ++j) fb[0][j] = frag_read<BMODE>(sb, 64 * wn + 32 * j, 0, lane);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      if (kk + 1 < 4) {
#pragma unroll
        for (int i = 0; i < 2; ++i) fa[(kk + 1) & 1][i] = frag_read<AMODE>(sa, 64 * wm + 32 * i, kk + 1, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) fb[(kk + 1) & 1][j] = frag_read<BMODE>(sb, 64 * wn + 32 * j, kk + 1, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk & 1][i], fb[kk & 1][j], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) {
      char* na = smem + (cur ^ 1) * GEMM_STAGE_BYTES;
      tile_store<AMODE>(ra, na, t);
      tile_store<BMODE>(rb, na + 16384, t);
    }
    __syncthreads();
    cur ^= 1;
  }

  // ---- epilogue: accumulators -> fp32 LDS tile -> row-major functor (coalesced global access) ----
  float* tile = reinterpret_cast<float*>(smem);
  {
    const int h = lane >> 5, c = lane & 31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = 64 * wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
          tile[row * GEMM_EPI_PITCH + 64 * wn + 32 * j + c] = acc[i][j][r];
        }
  }
  __syncthreads();
  epi.tile_begin(bm * GEMM_BM, bn * GEMM_BN, split);
  {
    const int c4 = (t & 31) * 4;
    constexpr int NB = epi_prefetch_batch<Epi>::value;
#pragma unroll
    for (int b0 = 0; b0 < 16; b0 += NB) {
      typename Epi::Pre pre[NB];
#pragma unroll
      for (int it = 0; it < NB; ++it) pre[it] = epi.prefetch(bm * GEMM_BM + (t >> 5) + 8 * (b0 + it), bn * GEMM_BN + c4);
#pragma unroll
      for (int it = 0; it < NB; ++it) {
        const int row = (t >> 5) + 8 * (b0 + it);
        const f32x4 v = *reinterpret_cast<const f32x4*>(&tile[row * GEMM_EPI_PITCH + c4]);
        epi_apply(epi, b0 + it, bm * GEMM_BM + row, bn * GEMM_BN + c4, v, pre[it], 0);
      }
    }
  }
  __syncthreads();
  epi.tile_end(tile);
}
