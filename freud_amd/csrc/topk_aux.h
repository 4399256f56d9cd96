// AuxK branch of the TopK SAE on the COMPACTED set of dead latents (topkautoencoder.py:108-127).
//
// The AuxK selection takes k_aux = d/2 of the dead latents per row: 384 entries per row at d = 768, six times the main
// selection, and at least 1.5 % of any dead set -- too dense for the gather kernels (a gathered W_dec row per selected
// pair: 38 GB per step at C3) and, while few latents are dead, far too narrow for [M x n] dense GEMMs over the whole
// dictionary.  So the dead latents are compacted: dead_cols[r] = r-th dead latent, ND = their number, ND_p = ND rounded up
// to 256, and the whole branch runs as dense MFMA GEMMs of width ND_p -- cost proportional to the number of dead latents:
//     A_aux [M x ND_p]   masked selection, compact (topk_select_reg_kernel<.., COMPACT>)
//     e_hat  = A_aux W_dec[dead]            K = ND_p   (EpiAuxDecode: + b_dec, - e, sum of squares, d e_hat)
//     d A    = d e_hat W_dec[dead]^T        K = d      (EpiTopkDpre: ReLU / selection gate, column sums)
//     d W_dec[dead] = A_aux^T d e_hat,  d W_enc[dead] = d A^T sae_in      K = M, split-K slabs
// and the three gradients are added to the rows of the dead latents (aux_scatter_* below) after the main selection's CSC
// backward wrote them.  ND lives on the device only: the launches cover the static maximum and workgroups beyond the
// dynamic extent exit (GemmArgs::dyn), so the step still has no host synchronisation.
#pragma once
#include "topk_kernels.h"

// tkd: [0] ND  [1] ND_p / 128  [2] ND_p / 256  [3] ND_p / 64  [4] ND_p
enum { TKD_ND = 0, TKD_T128 = 1, TKD_T256 = 2, TKD_KT = 3, TKD_NDP = 4 };

// dead_mask_kernel + compaction.  One block of 1024 threads, thread t owns the contiguous columns [t per, (t+1) per).
//   dead[i], did_fire[i] = 0, tk / tkf as dead_mask_kernel;
//   dead_cols[r] (r < ND_p; -1 beyond ND), vec_rank[g] = number of dead columns before column 8 g, vec_bits[g] = dead bits of
//   columns 8 g .. 8 g + 7.
__global__ __launch_bounds__(1024) void dead_compact_kernel(const long long* __restrict__ nfsf, unsigned char* __restrict__ dead,
                                                             float* __restrict__ did_fire, int n, int n_p, double threshold, int d,
                                                             int* __restrict__ tk, float* __restrict__ tkf, int* __restrict__ tkd,
                                                             int* __restrict__ dead_cols, int* __restrict__ vec_rank,
                                                             unsigned char* __restrict__ vec_bits) {
  __shared__ int wtot[16];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int per = ((n_p / 8 + 1023) / 1024) * 8;           // columns per thread, a multiple of 8
  const int c0 = t * per;
  int cnt = 0;
  // the thread's dead bits stay in registers for the second walk (per <= 128 columns: n_p <= 131072): reading dead[] back from
  // global memory was a chain of dependent round trips in this one-workgroup kernel
  unsigned long long dbits[2] = {0ull, 0ull};
  const bool inreg = per <= 128;
  // (eight counters per trip from clamped addresses, all in flight before the first is used: as a plain loop with its two exit tests per
  // element every load was waited for on its own -- 24 memory latencies in a row per thread of this one-workgroup kernel at n = 24 576)
  for (int i0 = c0; i0 < c0 + per && i0 < n_p; i0 += 8) {       // per and n_p are multiples of 8
    long long fv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) fv[u] = nfsf[i0 + u < n ? i0 + u : n - 1];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u;
      const bool dd = i < n && (double)fv[u] > threshold;
      dead[i] = dd;
      did_fire[i] = 0.f;
      cnt += dd;
      if (dd && inreg) dbits[(i - c0) >> 6] |= 1ull << ((i - c0) & 63);
    }
  }
  int inc = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(inc, o, 64);
    if (lane >= o) inc += u;
  }
  if (lane == 63) wtot[w] = inc;
  __syncthreads();
  int base = 0, total = 0;
  for (int ww = 0; ww < 16; ++ww) {
    if (ww < w) base += wtot[ww];
    total += wtot[ww];
  }
  int pos = base + inc - cnt;
  for (int i = c0; i < c0 + per && i < n_p; i += 8) {
    unsigned bits = 0;
    vec_rank[i >> 3] = pos;
    unsigned b8 = 0;
    if (inreg) b8 = (unsigned)(dbits[((i - c0) >> 6) & 1] >> ((i - c0) & 63)) & 0xFFu;      // (i - c0 is a multiple of 8)
    else
      for (int e = 0; e < 8; ++e) b8 |= dead[i + e] ? (1u << e) : 0u;                         // written by this thread above
    for (int e = 0; e < 8; ++e)
      if ((b8 >> e) & 1u) {
        bits |= 1u << e;
        dead_cols[pos++] = i + e;
      }
    vec_bits[i >> 3] = (unsigned char)bits;
  }
  // the transposed copy the compact select reads (topk_select_reg_kernel<12, true>: thread t of a row's workgroup owns the vectors
  // t, t + 256, ...): [256][16] bytes behind the table, zeros past the row's end
  if (n_p <= 32768) {
    __syncthreads();                       // (the table above: global writes of this workgroup)
    unsigned char* vt = vec_bits + vec_bits_t_offset(n_p);
    for (int i = t; i < VEC_BITS_T_BYTES; i += 1024) {
      const int g = (i & 15) * 256 + (i >> 4);
      vt[i] = g < n_p / 8 ? vec_bits[g] : (unsigned char)0;
    }
  }
  // ND_p: a multiple of 256 where the dictionary's padded size is one (the 256x256 GEMM kernel then applies), else of 128 --
  // never beyond n_p, the width of every compact buffer
  const int gran = (n_p & 255) ? 128 : 256;
  const int nd_p = ((total + gran - 1) / gran) * gran;
  for (int r = total + t; r < nd_p; r += 1024) dead_cols[r] = -1;
  if (t == 0) {
    const int k_aux_full = d / 2;
    tk[0] = total;
    tk[1] = total < k_aux_full ? total : k_aux_full;
    tkf[0] = fminf((float)total / (float)k_aux_full, 1.0f);
    tkd[TKD_ND] = total;
    tkd[TKD_T128] = nd_p / 128;
    tkd[TKD_T256] = nd_p / 256;
    tkd[TKD_KT] = nd_p / 64;
    tkd[TKD_NDP] = nd_p;
  }
}

// Wdd[r][:] = Wd_b[dead_cols[r]][:] for r < ND_p (zero rows beyond ND).  One wave per row, 4 rows per block; grid covers n_p.
__global__ __launch_bounds__(256) void aux_gather_rows_kernel(const bf16_t* __restrict__ Wd_b, const int* __restrict__ dead_cols,
                                                               const int* __restrict__ tkd, bf16_t* __restrict__ Wdd, int d_p) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= tkd[TKD_NDP]) return;
  const int j = dead_cols[r];
  u32x4* dst = reinterpret_cast<u32x4*>(Wdd + (int64_t)r * d_p);
  const u32x4* src = reinterpret_cast<const u32x4*>(Wd_b + (int64_t)(j >= 0 ? j : 0) * d_p);
  for (int i = lane; i < d_p / 8; i += 64) dst[i] = j >= 0 ? src[i] : u32x4{0u, 0u, 0u, 0u};
}

// AuxK decode epilogue: e_hat = bf16(acc) + b_dec; dh = e_hat - e (the aux decode predicts the main residual); sum of
// squares per 128x128 tile -> part[tile] (the caller zero-fills part[0 .. M_p) first; topk_finalize_kernel sums all of it).
struct EpiAuxDecode {
  static constexpr bool ROUNDS_BF16_FIRST = true;     // gemm256.h: the tile goes through LDS as bf16
  static constexpr bool DEEP_A_RING = true;           // gemm256.h: K = the dead latents, the AuxK activations stream from HBM
  static constexpr int PREFETCH_BATCH = EPI_BATCH_HEAVY;
  const float* e;       // [M_p][d_p]
  const float* b_dec;   // [d_p]
  float* dh;            // [M_p][d_p]
  float* part;
  int64_t M;
  int d, d_p, nbn;
  float sq;
  int tile_id;
  __device__ void tile_begin(int row0, int col0, int) {
    sq = 0.f;
    tile_id = (row0 / GEMM_BM) * nbn + col0 / GEMM_BN;
  }
  struct Pre { f32x4 ev, b; };
  __device__ Pre prefetch(int row, int col) const {
    return Pre{*reinterpret_cast<const f32x4*>(e + (int64_t)row * d_p + col), *reinterpret_cast<const f32x4*>(b_dec + col)};
  }
  __device__ void apply(int row, int col, f32x4 v, const Pre& pre) {
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float out = 0.f;
      if (row < M && col + j < d) {
        out = bf16_round(v[j]) + pre.b[j] - pre.ev[j];
        sq += out * out;
      }
      o[j] = out;
    }
    *reinterpret_cast<f32x4*>(dh + (int64_t)row * d_p + col) = o;
  }
  __device__ void tile_end(float* scratch) {
    const float s = block_sum_256_lds(sq, scratch);
    if ((threadIdx.x & 255) == 0) part[tile_id] = s;
  }
};

// g[dead_cols[r]][:] += sum_s slab[s][r][:]  (r < ND; one wave per row, 4 rows per block; grid covers n_p).  The split-K
// factor is the one the GEMM chose on the device from the same dead count (gemm_dyn_splits; tile = 128 or 256 rows / columns).
__global__ __launch_bounds__(256) void aux_scatter_rows_kernel(const float* __restrict__ slab, int64_t slab_stride, int smin, int smax,
                                                                int ktiles, int tile, const int* __restrict__ dead_cols,
                                                                const int* __restrict__ tkd, float* __restrict__ g, int d_p) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= tkd[TKD_ND]) return;
  const int splits = gemm_dyn_splits((tkd[TKD_NDP] / tile) * (d_p / tile), smin, smax, ktiles);
  const int j = dead_cols[r];
  f32x4* dst = reinterpret_cast<f32x4*>(g + (int64_t)j * d_p);
  for (int i = lane; i < d_p / 4; i += 64) {
    f32x4 a = reinterpret_cast<const f32x4*>(slab + (int64_t)r * d_p)[i];
    for (int s = 1; s < splits; ++s) a += reinterpret_cast<const f32x4*>(slab + s * slab_stride + (int64_t)r * d_p)[i];
    dst[i] += a;
  }
}

// d b_enc of the dead latents: exact[j] += sum_tiles part[tile][r], gbe[j] = bf16(exact[j]) (the encoder bias enters the bf16
// addmm as a bf16 cast: its gradient is one rounding of the whole column sum, main + aux).  64 compact columns per
// workgroup, 16 waves each a slice of the row tiles, combined through LDS in wave order (fixed order: deterministic).
__global__ __launch_bounds__(1024) void aux_scatter_dbe_kernel(const float* __restrict__ part, int ntiles, int ld,
                                                                const int* __restrict__ dead_cols, const int* __restrict__ tkd,
                                                                float* __restrict__ exact, float* __restrict__ gbe) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = blockIdx.x * 64 + lane;
  const int nd = tkd[TKD_ND];
  if (blockIdx.x * 64 >= nd) return;
  float s = 0.f;
  if (r < nd)
    for (int i = w; i < ntiles; i += 16) s += part[(int64_t)i * ld + r];
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && r < nd) {
    float tot = 0.f;
    for (int ww = 0; ww < 16; ++ww) tot += red[ww][lane];
    const int j = dead_cols[r];
    const float v = exact[j] + tot;
    exact[j] = v;
    gbe[j] = bf16_round(v);
  }
}
