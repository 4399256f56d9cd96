// Kernels of the L1 (tied-weight) SAE train step around the MFMA GEMMs: weight/activation
// preparation, GEMM epilogues, gradient reduction, clip + Adam/RAdam.
// Reference arithmetic: src/models/l1autoencoder.py:29-36,69-95; src/scripts/train_sae.py:429-451.
#pragma once
#include "gemm.h"

// ------------------------------------------------------------------------------------------
// prep_w: W <- W / max(||W[:,j]||_2, 1e-12) in place (l1autoencoder.py:71-73), and the two bf16
// GEMM operand copies: Wb[d_p][n_p] (K = n contiguous, decoder) and Wt[n_p][d_p] (K = d contiguous,
// encoder / dc).  Phase 1: per-column sums of squares over 32-row slabs (fixed order ->
// deterministic); phase 2: 64x64 tiles normalise, cast and transpose through LDS.
// ------------------------------------------------------------------------------------------
// The partial sum of a (32-row slab, column) has ONE definition, shared with optimizer_l1_kernel (which leaves the same
// partials of the weights it has just updated, so that a training step needs no colnorm pass): thread ty of 8 adds the
// squares of rows ty, ty + 8, ty + 16, ty + 24 in that order (ss = fma(v, v, ss) from 0: spelled out, so that every kernel
// that forms these sums -- colnorm_partial, optimizer_l1, optimizer_l1_cols -- rounds alike whatever the compiler would
// contract), the eight results are added as ((0+1)+(2+3))+((4+5)+(6+7)).
__device__ __forceinline__ float colnorm_tree8(const float (*red)[128], int col) {
  return ((red[0][col] + red[1][col]) + (red[2][col] + red[3][col])) + ((red[4][col] + red[5][col]) + (red[6][col] + red[7][col]));
}

__global__ __launch_bounds__(256) void colnorm_partial_kernel(const float* __restrict__ W, float* __restrict__ part,
                                                               int n_p) {
  // grid (n_p/128, d_p/32); thread -> columns 4 tx .. 4 tx + 3 (tx < 32), rows ty, ty + 8, ty + 16, ty + 24 of the 32-row slab
  __shared__ float red[8][128];
  const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
  const int col = blockIdx.x * 128 + 4 * tx, r0 = blockIdx.y * 32;
  f32x4 ss = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(W + (int64_t)(r0 + ty + 8 * i) * n_p + col);
#pragma unroll
    for (int q = 0; q < 4; ++q) ss[q] = __builtin_fmaf(v[q], v[q], ss[q]);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) red[ty][4 * tx + q] = ss[q];
  __syncthreads();
  if (t < 128) part[(int64_t)blockIdx.y * n_p + blockIdx.x * 128 + t] = colnorm_tree8(red, t);
}

__global__ __launch_bounds__(256) void normalize_cast_kernel(float* __restrict__ W, const float* __restrict__ part,
                                                              int nslab, bf16_t* __restrict__ Wb,
                                                              bf16_t* __restrict__ Wt, int d_p, int n_p) {
  // grid (n_p/64, d_p/64): one 64x64 tile
  __shared__ float denom_s[64];
  __shared__ __attribute__((aligned(16))) bf16_t tT[64][72];
  const int t = threadIdx.x, tx = t & 63, ty = t >> 6;
  const int col = blockIdx.x * 64 + tx, r0 = blockIdx.y * 64;
  if (ty == 0) {
    float ss = 0.f;
#pragma unroll 8
    for (int i = 0; i < nslab; ++i) ss += part[(int64_t)i * n_p + col];      // (unrolled: eight partials in flight, same order)
    denom_s[tx] = fmaxf(sqrtf(ss), 1e-12f);
  }
  __syncthreads();
  const float denom = denom_s[tx];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = ty + 4 * i;
    const int64_t o = (int64_t)(r0 + r) * n_p + col;
    const float v = W[o] / denom;
    W[o] = v;
    const bf16_t b = (bf16_t)v;
    Wb[o] = b;
    tT[tx][r] = b;
  }
  __syncthreads();
  // 64 columns x 64 rows -> Wt[col][r0 .. r0+64): 8 pieces of 16 B per column, 512 pieces
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int piece = t + 256 * it, c = piece >> 3, sgm = piece & 7;
    *reinterpret_cast<u32x4*>(Wt + (int64_t)(blockIdx.x * 64 + c) * d_p + r0 + sgm * 8) =
        *reinterpret_cast<const u32x4*>(&tT[c][sgm * 8]);
  }
}

// ------------------------------------------------------------------------------------------
// prep_x: activation rows (fp32 / fp16 / bf16, [M][d]) -> zero-padded bf16 GEMM operand
// xb[M_p][d_p]; counts the entries equal to -1 (mse_loss's ignored_index, l1autoencoder.py:31).
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ float load_as_float(const T* p) { return (float)*p; }

template <typename T> struct Vec8;
template <> struct Vec8<float> { typedef __attribute__((ext_vector_type(8))) float type; };
template <> struct Vec8<_Float16> { typedef __attribute__((ext_vector_type(8))) _Float16 type; };
template <> struct Vec8<bf16_t> { typedef bf16x8 type; };

// VEC: d % 8 == 0 and x 16-B aligned -> one vector load per 8 elements.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void prep_x_kernel(const T* __restrict__ x, bf16_t* __restrict__ xb,
                                                      unsigned int* __restrict__ masked_count, int64_t M, int d,
                                                      int64_t M_p, int d_p) {
  const unsigned int chunks_per_row = (unsigned int)d_p >> 3;
  const unsigned int total = (unsigned int)(M_p * chunks_per_row);   // host guarantees < 2^32
  unsigned int masked = 0;
  __shared__ unsigned int red[4];
  for (unsigned int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int64_t row = i / chunks_per_row;
    const int c0 = (int)(i - (unsigned int)row * chunks_per_row) * 8;
    bf16x8 o;
    if (VEC && row < M && c0 + 8 <= d) {
      const typename Vec8<T>::type v = *reinterpret_cast<const typename Vec8<T>::type*>(x + row * d + c0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (float)v[j];
        masked += (f == -1.0f);
        o[j] = (bf16_t)f;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = 0.f;
        if (row < M && c0 + j < d) {
          v = load_as_float(x + row * d + c0 + j);
          masked += (v == -1.0f);
        }
        o[j] = (bf16_t)v;
      }
    }
    *reinterpret_cast<bf16x8*>(xb + row * d_p + c0) = o;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) masked += (unsigned int)__shfl_xor((int)masked, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = masked;
  __syncthreads();
  if (threadIdx.x == 0) masked_count[blockIdx.x] = red[0] + red[1] + red[2] + red[3];   // per-block partial, no atomics
}

// scal[0] = count of unmasked entries, scal[1] = alpha/count (d loss / d squared-error term), scal[2] = 1/M
// gstats != null (data parallel, dp_kernels.h): the count and the row number are the GLOBAL ones, summed over the ranks.
__global__ __launch_bounds__(256) void finalize_count_kernel(const unsigned int* masked_part, int nparts, float* scal,
                                                              int64_t M, int d, float alpha, const double* gstats) {
  __shared__ unsigned int red[4];
  unsigned int m = 0;
  for (int i = threadIdx.x; i < nparts; i += 256) m += masked_part[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m += (unsigned int)__shfl_xor((int)m, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double count = gstats ? gstats[0] : (double)M * d - (double)(red[0] + red[1] + red[2] + red[3]);
    scal[0] = (float)count;
    scal[1] = alpha / (float)count;
    scal[2] = 1.0f / (float)(gstats ? gstats[1] : (double)M);
    scal[3] = (float)((double)M * d - (double)(red[0] + red[1] + red[2] + red[3]));     // this rank's own count
  }
}

// ------------------------------------------------------------------------------------------
// Store of a large output that is next read only after it has left every cache (the [M x n] latent / pre / dpre
// streams): non-temporal, so that it does not evict the operand tiles the co-resident workgroups share (same-box A/B at
// d=1280 n=40960: encoder GEMM 7.1 -> 6.4 ms, dpre 7.85 -> 7.55; TopK encoder at d=768 3.03 -> 2.55 ms)
#ifndef EPI_BATCH_HEAVY
#define EPI_BATCH_HEAVY 8      // prefetch batch of the functors with a large Pre (gemm.h: epi_prefetch_batch)
#endif
#define EPI_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
// ... and the matching read of such a stream inside an epilogue (dpre GEMM reading the latent: 7.6 -> 7.1 ms)
#define EPI_LOAD(ptr) __builtin_nontemporal_load(ptr)
#ifdef EPI_KO_SKIP_STORE
#define EPI_ENC_STORE(ptr, val) asm volatile("" :: "v"(val))      // knock-out build: the encoder's latent is computed and dropped
#else
#define EPI_ENC_STORE(ptr, val) EPI_STORE(ptr, val)
#endif

// GEMM epilogues (row-major over the fp32 tile, 4 consecutive columns per call).  Two phases per thread and tile:
// prefetch(row, col) -> Pre issues every global LOAD the element needs (all of a thread's 16 prefetches are in flight
// before the first apply, so an epilogue pass costs about one memory latency instead of 16), apply() computes and stores.
// ------------------------------------------------------------------------------------------
// c = relu(bf16(x W) + b)  (l1autoencoder.py:74), rows >= M forced to 0; L1 partial sum per tile.
struct EpiEnc {
  static constexpr bool ROUNDS_BF16_FIRST = true;     // gemm256.h: the tile goes through LDS as bf16
  bf16_t* c;            // [M_p][n_p]
  const float* bias;    // [n_p]
  float* l1_part;       // [tiles]
  int64_t M;
  int n_p, nbn;
  // (the timing experiment "no latent store" -- round 2's `bench.py --dbg 70` -- is a BUILD switch since round 5: -DEPI_KO_SKIP_STORE.  As a
  // run-time member it put a branch around the store of every s_apply call, i.e. a basic-block boundary per 8 latents in the epilogue.)
  float l1;
  int tile_id;
  __device__ void tile_begin(int row0, int col0, int) {
    l1 = 0.f;
    tile_id = (row0 / GEMM_BM) * nbn + col0 / GEMM_BN;
  }
  struct Pre { f32x4 b; };
  __device__ Pre prefetch(int, int col) const { return Pre{*reinterpret_cast<const f32x4*>(bias + col)}; }
  __device__ void apply(int row, int col, f32x4 v, const Pre& pre) {
    const f32x4 b = pre.b;
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float cv = fmaxf(bf16_round(v[j]) + b[j], 0.f);
      if (row >= M) cv = 0.f;
      l1 += cv;
      o[j] = (bf16_t)cv;
    }
    EPI_ENC_STORE(reinterpret_cast<bf16x4*>(c + (int64_t)row * n_p + col), o);
  }
  static constexpr bool WIDE8 = true;                 // gemm256.h: eight columns per thread, 16-byte latent stores
  __device__ void apply8(int row, int col, f32x4 v0, f32x4 v1, const Pre& p0, const Pre& p1) {
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float c0 = fmaxf(bf16_round(v0[j]) + p0.b[j], 0.f), c1 = fmaxf(bf16_round(v1[j]) + p1.b[j], 0.f);
      if (row >= M) c0 = c1 = 0.f;
      l1 += c0 + c1;
      o[j] = (bf16_t)c0;
      o[4 + j] = (bf16_t)c1;
    }
    EPI_ENC_STORE(reinterpret_cast<bf16x8*>(c + (int64_t)row * n_p + col), o);
  }
  __device__ void tile_end(float* scratch) {
    const float s = block_sum_256_lds(l1, scratch);
    if ((threadIdx.x & 255) == 0) l1_part[tile_id] = s;
  }
  // ---- streaming form (gemm256s.h): the lane's 8 bias values per tile in registers, the L1 sum kept per thread over ALL the
  // workgroup's tiles and left as ONE partial per workgroup (l1_part[blockIdx.x]; the host zeroes the other entries)
  static constexpr bool STREAM = true;
  struct SPre {};
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  f32x2 sbp[4], l1p;          // the lane's 8 bias values as pairs (v_pk_add_f32), the L1 sum as a pair of partial sums
  __device__ void s_begin() { l1p = f32x2{0.f, 0.f}; }
  __device__ int64_t s_rows() const { return M; }
  __device__ void s_tile(int, int col) {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + col), b1 = *reinterpret_cast<const f32x4*>(bias + col + 4);
    sbp[0] = f32x2{b0[0], b0[1]}; sbp[1] = f32x2{b0[2], b0[3]}; sbp[2] = f32x2{b1[0], b1[1]}; sbp[3] = f32x2{b1[2], b1[3]};
  }
  __device__ SPre s_prefetch(int, int) const { return SPre{}; }
  // v: the accumulators ALREADY rounded to bf16 (exact as floats).  Packed fp32 adds for the bias and the L1 sum: this epilogue is
  // bound by its vector-instruction count (28 per 8 latents here, 44 in the element-wise form)
  template <bool PARTIAL>
  __device__ void s_apply(int row, int col, f32x4 v0, f32x4 v1, const SPre&) {
    f32x2 sv[4] = {f32x2{v0[0], v0[1]} + sbp[0], f32x2{v0[2], v0[3]} + sbp[1], f32x2{v1[0], v1[1]} + sbp[2], f32x2{v1[2], v1[3]} + sbp[3]};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sv[k] = f32x2{fmaxf(sv[k][0], 0.f), fmaxf(sv[k][1], 0.f)};
      if (PARTIAL && row >= M) sv[k] = f32x2{0.f, 0.f};
      l1p += sv[k];
    }
    const bf16x8 o = {(bf16_t)sv[0][0], (bf16_t)sv[0][1], (bf16_t)sv[1][0], (bf16_t)sv[1][1],
                      (bf16_t)sv[2][0], (bf16_t)sv[2][1], (bf16_t)sv[3][0], (bf16_t)sv[3][1]};
    EPI_ENC_STORE(reinterpret_cast<bf16x8*>(c + (int64_t)row * n_p + col), o);
  }
  __device__ void s_tile_end(int, int) {}
  __device__ void s_end(float* scratch) {
    const float v = wave_sum(l1p[0] + l1p[1]);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    lds_barrier();
    if (threadIdx.x == 0) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += scratch[w];
      l1_part[blockIdx.x] = s;
    }
  }
};

// x_hat = bf16(c W^T); masked residual; dx_hat = bf16(2 alpha (x_hat - x) [x != -1] / count)
// (l1autoencoder.py:84,86,29-36 and its backward).  Partial sums: masked sq-err, unmasked sq-err.
template <typename T>
struct EpiDec {
  static constexpr int PREFETCH_BATCH = EPI_BATCH_HEAVY;
  static constexpr bool ROUNDS_BF16_FIRST = true;     // (exact in the fp8 kernel too: vs is a power of two, gemm256_fp8.h)
  static constexpr bool DEEP_A_RING = true;           // gemm256.h: K = n_dict, the latent streams from HBM
  const T* x;           // original activations [M][d]
  bf16_t* dxh;          // [M_p][d_p]
  const float* scal;    // scal[1] = alpha / count
  float* sq_part;       // [tiles][2]
  int64_t M;
  int d, d_p, nbn;
  const float* vscale;  // fp8 decoder GEMM: *vscale = 1 / (s_c s_w) undoes the operand scales (null: 1)
  float sq, plain, scale, vs;
  int tile_id;
  __device__ void tile_begin(int row0, int col0, int) {
    sq = plain = 0.f;
    scale = scal[1];
    vs = vscale ? *vscale : 1.0f;
    tile_id = (row0 / GEMM_BM) * nbn + col0 / GEMM_BN;
  }
  struct Pre { float xv[4]; };
  // vec_all (host: no padded row or column -- M == M_p, d == d_p -- and x aligned to four elements): ONE unconditional vector load
  // per call.  The general form below is four predicated scalar loads, and hipcc keeps each in its own exec-masked block WITH its
  // conversion, i.e. with a wait: 128 dependent round trips per thread and tile in the decoder's epilogue (found in round 4 through
  // the same pattern in csc_fill_kernel; a decoder tile's "prefetch one batch ahead" never overlapped anything).
  int vec_all;
  __device__ Pre prefetch(int row, int col) const {
    Pre p;
    if (vec_all) {
      typedef T vec4_t __attribute__((ext_vector_type(4)));
      const vec4_t q = *reinterpret_cast<const vec4_t*>(x + (int64_t)row * d + col);
#pragma unroll
      for (int j = 0; j < 4; ++j) p.xv[j] = (float)q[j];
      return p;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) p.xv[j] = (row < M && col + j < d) ? load_as_float(x + (int64_t)row * d + col + j) : 0.f;
    return p;
  }
  __device__ void apply(int row, int col, f32x4 v, const Pre& pre) {
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float g = 0.f;
      if (row < M && col + j < d) {
        const float xv = pre.xv[j];
        const float e = bf16_round(v[j] * vs) - xv;
        plain += e * e;
        if (xv != -1.0f) {
          sq += e * e;
          g = (e * 2.0f) * scale;
        }
      }
      o[j] = (bf16_t)g;
    }
    *reinterpret_cast<bf16x4*>(dxh + (int64_t)row * d_p + col) = o;
  }
  __device__ void tile_end(float* scratch) {
    const float a = block_sum_256_lds(sq, scratch);
    const float b = block_sum_256_lds(plain, scratch + 8);
    if ((threadIdx.x & 255) == 0) {
      sq_part[2 * tile_id] = a;
      sq_part[2 * tile_id + 1] = b;
    }
  }
};

// dpre = (bf16(dx_hat W) + sign(c)/M) * [c > 0]; db partial column sums per row-tile.
struct EpiDpre {
  static constexpr bool ROUNDS_BF16_FIRST = true;     // gemm256.h: the tile goes through LDS as bf16
  static constexpr int PREFETCH_BATCH = EPI_BATCH_HEAVY;
  const bf16_t* c;      // [M_p][n_p]
  bf16_t* dpre;         // [M_p][n_p]
  float* db_part;       // [nbm][n_p]
  const float* scal;    // scal[2] = 1/M
  int n_p;
  float colsum[4];
  float inv_m;
  int row_tile, col0_;
  __device__ void tile_begin(int row0, int col0, int) {
    colsum[0] = colsum[1] = colsum[2] = colsum[3] = 0.f;
    inv_m = scal[2];
    row_tile = row0 / GEMM_BM;
    col0_ = col0;
  }
  struct Pre { bf16x4 cv; };
  __device__ Pre prefetch(int row, int col) const { return Pre{EPI_LOAD(reinterpret_cast<const bf16x4*>(c + (int64_t)row * n_p + col))}; }
  __device__ void apply(int row, int col, f32x4 v, const Pre& pre) {
    const bf16x4 cv = pre.cv;
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float g = ((float)cv[j] > 0.f) ? (bf16_round(v[j]) + inv_m) : 0.f;
      colsum[j] += g;
      o[j] = (bf16_t)g;
    }
    EPI_STORE(reinterpret_cast<bf16x4*>(dpre + (int64_t)row * n_p + col), o);
  }
  __device__ void tile_end(float* scratch) {
    // thread t owns columns 4*(t&31).. of row group t>>5: reduce the 8 row groups through LDS
    const int t = threadIdx.x & 255;
    f32x4 cs = {colsum[0], colsum[1], colsum[2], colsum[3]};
    *reinterpret_cast<f32x4*>(scratch + (t >> 5) * 128 + (t & 31) * 4) = cs;
    lds_barrier();
    if (t < 128) {
      float s = 0.f;
#pragma unroll
      for (int gidx = 0; gidx < 8; ++gidx) s += scratch[gidx * 128 + t];
      db_part[(int64_t)row_tile * n_p + col0_ + t] = s;
    }
  }
  // ---- streaming form (gemm256s.h): a wave owns 128 rows x 64 columns, i.e. exactly the rows of one db_part row; the lane's 8
  // column sums over its 16 rows are added over the 8 lanes that share the columns (fixed order) and stored by lanes 0-7
  static constexpr bool STREAM = true;
  struct SPre { u32x4 cw; };          // eight latents as bf16 pairs
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  f32x2 scp[4], invp;
  __device__ void s_begin() {}
  __device__ int64_t s_rows() const { return (int64_t)1 << 62; }      // (rows >= M carry c = 0: the gate already zeroes them)
  __device__ void s_tile(int, int) {
    invp = f32x2{scal[2], scal[2]};
#pragma unroll
    for (int k = 0; k < 4; ++k) scp[k] = f32x2{0.f, 0.f};
  }
#ifdef DPRE_KO_GATE      // TIMING EXPERIMENT ONLY (results WRONG): no read of the latent -- what would a 1-bit gate mask from the encoder buy this GEMM at most?
  __device__ SPre s_prefetch(int row, int col) const { return SPre{u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}}; }
#else
  __device__ SPre s_prefetch(int row, int col) const { return SPre{EPI_LOAD(reinterpret_cast<const u32x4*>(c + (int64_t)row * n_p + col))}; }
#endif
  // the gate c > 0 on the bf16 BITS: the upper element of a pair is positive iff the dword, as a signed integer, exceeds 0xFFFF; the
  // lower one iff the dword shifted left by 16 is positive (-0.0, which a max(x, 0) may leave, is negative as an integer)
  template <bool PARTIAL>
  __device__ void s_apply(int row, int col, f32x4 v0, f32x4 v1, const SPre& pre) {
    f32x2 g[4] = {f32x2{v0[0], v0[1]} + invp, f32x2{v0[2], v0[3]} + invp, f32x2{v1[0], v1[1]} + invp, f32x2{v1[2], v1[3]} + invp};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int cw = (int)pre.cw[k];
      g[k] = f32x2{(cw << 16) > 0 ? g[k][0] : 0.f, cw > 0xFFFF ? g[k][1] : 0.f};
      scp[k] += g[k];
    }
    const bf16x8 o = {(bf16_t)g[0][0], (bf16_t)g[0][1], (bf16_t)g[1][0], (bf16_t)g[1][1],
                      (bf16_t)g[2][0], (bf16_t)g[2][1], (bf16_t)g[3][0], (bf16_t)g[3][1]};
    EPI_STORE(reinterpret_cast<bf16x8*>(dpre + (int64_t)row * n_p + col), o);
  }
  __device__ void s_tile_end(int row_w, int col) {
    f32x4 lo, hi;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = scp[j >> 1][j & 1];
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (j < 4) lo[j] = v; else hi[j - 4] = v;
    }
    if ((threadIdx.x & 63) < 8) {
      float* dst = db_part + (int64_t)(row_w / GEMM_BM) * n_p + col;
      *reinterpret_cast<f32x4*>(dst) = lo;
      *reinterpret_cast<f32x4*>(dst + 4) = hi;
    }
  }
  __device__ void s_end(float*) {}
};

// out[C][R] = in[R][C]^T (bf16; R, C multiples of 64): the weight-gradient GEMM's d-side operands dx_hat and x as [d_p][M_p] (round 5),
// so that its A fragments are K-contiguous.  One 64x64 tile per workgroup through LDS: 16-byte loads along C, 16-byte stores along R.
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int64_t R, int C) {
  __shared__ __attribute__((aligned(16))) unsigned short tile[64][72];      // 144-byte rows (16-byte aligned)
  const int t = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = t + 256 * i, rr = idx >> 3, ch = idx & 7;            // row of the tile, 16-byte chunk of its 64 columns
    *reinterpret_cast<u32x4*>(&tile[rr][8 * ch]) = *reinterpret_cast<const u32x4*>(in + (r0 + rr) * C + c0 + 8 * ch);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = t + 256 * i, cc = idx >> 3, ch = idx & 7;            // column of the tile = output row, 8 consecutive input rows
    unsigned short v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[8 * ch + e][cc];
    u32x4 o = {(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16), (unsigned)v[4] | ((unsigned)v[5] << 16),
               (unsigned)v[6] | ((unsigned)v[7] << 16)};
    *reinterpret_cast<u32x4*>(out + (int64_t)(c0 + cc) * R + r0 + 8 * ch) = o;
  }
}

// split-K partial slab store for dW
struct EpiSlab {
  float* slab;          // [splits][rows][ld]
  int64_t slab_stride;  // rows*ld
  int ld;
  // tail-split launches (GemmArgs::tail_tiles): slab id 0 is the output itself (`slab`, no reduction pass for whole tiles),
  // id 1 + q the q-th piece tile of the compact overflow buffer tail[q][256][256]
  float* tail;
  float* base;
  int ld_cur;
  __device__ void tile_begin(int row0, int col0, int split) {
    if (tail != nullptr && split > 0) {
      ld_cur = 256;
      base = tail + (int64_t)(split - 1) * 65536 - ((int64_t)(row0 & ~255) * 256 + (col0 & ~255));
    } else {
      ld_cur = ld;
      base = slab + slab_stride * split;
    }
  }
  struct Pre {};
  __device__ Pre prefetch(int, int) const { return Pre{}; }
  __device__ void apply(int row, int col, f32x4 v, const Pre&) { *reinterpret_cast<f32x4*>(base + (int64_t)row * ld_cur + col) = v; }
  __device__ void tile_end(float*) {}
};

// out[tail tile] = sum over its K pieces (fixed order) of the overflow buffer tail[piece * tail_tiles + t][256][256]; the tail
// tiles are the last ones of the GEMM's tile walk (tile_coords).  grid (tail_tiles, 16), 256 threads: 16 rows of a tile each.
__global__ __launch_bounds__(256) void reduce_tail_kernel(const float* __restrict__ tail, float* __restrict__ out, int ld, int nbm,
                                                           int nbn, int tail_tiles, int tail_pieces) {
  const int tt = blockIdx.x, id = nbm * nbn - tail_tiles + tt;
  int bm, bn;
  tile_coords(id, nbm, nbn, bm, bn);
  const int t = threadIdx.x, c4 = (t & 63) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = blockIdx.y * 16 + (t >> 6) + 4 * i;
    const float* src = tail + (int64_t)tt * 65536 + r * 256 + c4;
    f32x4 a = *reinterpret_cast<const f32x4*>(src);
    for (int pc = 1; pc < tail_pieces; ++pc) a += *reinterpret_cast<const f32x4*>(src + (int64_t)pc * tail_tiles * 65536);
    *reinterpret_cast<f32x4*>(out + (int64_t)(bm * 256 + r) * ld + bn * 256 + c4) = a;
  }
}

// ------------------------------------------------------------------------------------------
// gradient reduction, loss finalisation, clip + optimizer
// ------------------------------------------------------------------------------------------
// grad[i] = sum_s slab[s][i] (float4 per thread)
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slab, float* __restrict__ grad,
                                                            int64_t n4, int64_t stride4, int splits) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4* s = reinterpret_cast<const f32x4*>(slab);
  f32x4 a = s[i];
  for (int k = 1; k < splits; ++k) a += s[i + k * stride4];
  reinterpret_cast<f32x4*>(grad)[i] = a;
}

// Fused gradient reduction: grad[i] = sum_s slab[s][i] over the flat [dW | db] range (db partials live at
// part_db[s][j]) AND the per-block sum of squares of the result (for clip_grad_norm_ when no data-parallel
// all-reduce follows).  Fixed order everywhere -> deterministic.
// grad_bf16 != null (data parallel with a bf16 payload): the same values are also written rounded to bf16, the copy the
// all-reduce then sums (gnorm_from_bf16_kernel brings the sum back to fp32).
// fin.cnt_part != null: the LAST block also finalises the loss scalars (what finalize_losses_kernel does, for the single-GPU
// fused path where the backward took alpha/count from the forward's counts itself): off the forward -> backward critical path.
struct LossFinalize {
  const float *l1_part, *sq_part, *cnt_part;
  int n_parts;
  float *scal, *metrics;
  int64_t M;
  int d;
  float alpha;
};

__device__ __forceinline__ void finalize_losses_block256(const LossFinalize& f) {
  __shared__ double redf[4][4];
  double a = 0, b = 0, c = 0, m = 0;
  for (int i = threadIdx.x; i < f.n_parts; i += 256) {
    a += (double)f.l1_part[i];
    b += (double)f.sq_part[2 * i];
    c += (double)f.sq_part[2 * i + 1];
    m += (double)f.cnt_part[i];
  }
  a = wave_sum_d(a); b = wave_sum_d(b); c = wave_sum_d(c); m = wave_sum_d(m);
  if ((threadIdx.x & 63) == 0) {
    const int w = threadIdx.x >> 6;
    redf[0][w] = a; redf[1][w] = b; redf[2][w] = c; redf[3][w] = m;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double l1 = (redf[0][0] + redf[0][1]) + (redf[0][2] + redf[0][3]), sq = (redf[1][0] + redf[1][1]) + (redf[1][2] + redf[1][3]);
    const double pl = (redf[2][0] + redf[2][1]) + (redf[2][2] + redf[2][3]), masked = (redf[3][0] + redf[3][1]) + (redf[3][2] + redf[3][3]);
    const double local = (double)f.M * f.d - masked;
    f.scal[0] = (float)local;
    f.scal[1] = f.alpha / (float)local;
    f.scal[2] = 1.0f / (float)f.M;
    f.scal[3] = (float)local;
    const double count = (double)f.scal[0];
    f.metrics[0] = (float)((double)f.alpha * (sq / count));
    f.metrics[1] = (float)(l1 / (double)f.M);
    f.metrics[2] = (float)(pl / ((double)f.M * f.d));
    f.metrics[3] = 0.f;
    f.metrics[4] = f.scal[0];
    f.metrics[5] = f.metrics[6] = f.metrics[7] = 0.f;
  }
}

__global__ __launch_bounds__(256) void reduce_grads_kernel(const float* __restrict__ slab, int64_t stride4, int splits,
                                                            const float* __restrict__ db_part, int db_rows, int n_p,
                                                            float* __restrict__ grad, int64_t nW4, int64_t n4,
                                                            double* __restrict__ gn_part, bf16_t* __restrict__ grad_bf16,
                                                            LossFinalize fin, int bal_m) {
  // bal_m > 0 (balanced fused backward, bwd_fused.h): column tile j of 128 arrives in bal_pieces(j, bal_m) pieces instead of
  // `splits` / `db_rows` everywhere; pieces are added in piece order (deterministic)
  __shared__ double red[4];
  if (fin.cnt_part && blockIdx.x == gridDim.x - 1) finalize_losses_block256(fin);
  double ss = 0;
  const int n4p = n_p / 4;
  // a + piece 1 + piece 2 + ... in piece order, the loads of up to eight pieces in flight together (round 4: with one load per trip
  // of a plain loop every piece was a dependent round trip -- ~11 of them per element at C2, about half of this kernel's 15 us)
  auto add_pieces = [](f32x4 a, const f32x4* p, int64_t stride, int cnt) -> f32x4 {
    int k = 1;
    for (; k + 7 < cnt; k += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(k + u) * stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) a += v[u];
    }
    if (k + 3 < cnt) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = p[(k + u) * stride];
#pragma unroll
      for (int u = 0; u < 4; ++u) a += v[u];
      k += 4;
    }
    if (k + 1 < cnt) {
      const f32x4 v0 = p[k * stride], v1 = p[(k + 1) * stride];
      a += v0;
      a += v1;
      k += 2;
    }
    if (k < cnt) a += p[k * stride];
    return a;
  };
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 a;
    if (i < nW4) {
      const f32x4* s = reinterpret_cast<const f32x4*>(slab);
      const int cnt = bal_m > 0 ? bal_pieces((int)(i % n4p) >> 5, bal_m) : splits;
      a = add_pieces(s[i], s + i, stride4, cnt);
    } else {
      const int64_t j4 = i - nW4;
      const f32x4* s = reinterpret_cast<const f32x4*>(db_part);
      const int cnt = bal_m > 0 ? bal_pieces((int)j4 >> 5, bal_m) : db_rows;
      a = add_pieces(s[j4], s + j4, (int64_t)n4p, cnt);
    }
    reinterpret_cast<f32x4*>(grad)[i] = a;
    if (grad_bf16) reinterpret_cast<bf16x4*>(grad_bf16)[i] = bf16x4{(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3]};
    ss += (double)(a[0] * a[0]) + (double)(a[1] * a[1]) + (double)(a[2] * a[2]) + (double)(a[3] * a[3]);
  }
  ss = wave_sum_d(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x == 0) gn_part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// data parallel, bf16 payload: the all-reduced bf16 gradient back to the fp32 buffer + per-block sum of (scale g)^2
__global__ __launch_bounds__(256) void gnorm_from_bf16_kernel(const bf16_t* __restrict__ gb, float* __restrict__ grad, int64_t n4,
                                                               float scale, double* __restrict__ part) {
  __shared__ double red[4];
  double s = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const bf16x4 v = reinterpret_cast<const bf16x4*>(gb)[i];
    const f32x4 g = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    reinterpret_cast<f32x4*>(grad)[i] = g;
    const f32x4 q = g * scale;
    s += (double)(q[0] * q[0]) + (double)(q[1] * q[1]) + (double)(q[2] * q[2]) + (double)(q[3] * q[3]);
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// db[j] = sum over row tiles of db_part[tile][j]; block = 32 columns x 8 row lanes (fixed order)
__global__ __launch_bounds__(256) void reduce_db_kernel(const float* __restrict__ part, float* __restrict__ db, int nbm,
                                                         int n_p) {
  __shared__ float red[8][32];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + tx;
  float s0 = 0.f, s1 = 0.f;
  int i = ty;
  for (; i + 8 < nbm; i += 16) {
    s0 += part[(int64_t)i * n_p + j];
    s1 += part[(int64_t)(i + 8) * n_p + j];
  }
  if (i < nbm) s0 += part[(int64_t)i * n_p + j];
  red[ty][tx] = s0 + s1;
  __syncthreads();
  if (ty == 0) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += red[k][tx];
    db[j] = s;
  }
}

// One block: loss scalars from the per-tile partials (double accumulation, fixed order -> deterministic).
// metrics: [0]=alpha*sq/count  [1]=l1_sum/M  [2]=plain_sq/(M*d)  [4]=count
// With cnt_part != null (fused forward) the masked-entry count comes from the forward's own partials and the
// scalars scal[0..2] = {count, alpha/count, 1/M} are produced here instead of by finalize_count_kernel.
// gstats != null (data parallel): every scalar is normalised by the GLOBAL count / rows so that the SUM over the ranks of
// metrics[0..2] is the loss of the whole batch; metrics[4] stays this rank's own count (it sums to the global one).
// push.world > 0 (data parallel with the peer exchange, fused forward): the batch statistics are exchanged RIGHT HERE.  The
// fused forward has just counted this rank's unmasked entries, so instead of a separate pass over x plus an all-reduce on a
// second stream, thread r of this one-block kernel writes (count, rows, epoch) into rank r's inbox (uncached device memory
// of the peer, mapped by sae_p2p_init; the epoch last, as a system-scope release) and then polls this rank's own inbox for
// rank r's triple: one one-way flag latency, no extra kernel, nothing on another stream.  The global values are also left in
// gstats_out for the kernels that read them later.  Inbox slots alternate with the epoch's parity; a peer's next write to
// the same slot lies behind two gradient exchanges, i.e. after this read.
struct StatsPush {
  unsigned long long* inbox[8];     // inbox[r]: rank r's inbox, [2 parities][8 sources][4 words]
  int rank, world;
  unsigned long long epoch;
  unsigned long long timeout_ticks; // of s_memrealtime (100 MHz)
  unsigned int* status;             // sticky failure word of the peer exchange (p2p_exchange.h: P2P_ST_*)
  unsigned int* status_host;        // its host-mapped mirror (may be null)
  double* gstats_out;               // [2]: global unmasked count, global rows
};

// The push itself: lane r (< world) of the calling wave writes (v0, v1, epoch) into rank r's inbox and waits for rank r's
// triple in its own.  The inbox is uncached device memory and every access a system-scope atomic: no cache holds these words.
// Nothing of the triple is ever held in a cache (uncached allocation, `sc0 sc1` accesses on both sides), so no write-back or
// invalidate is involved -- what is needed is ORDER: the two payload words, `s_waitcnt vmcnt(0)` (their write acknowledgements;
// inline asm so that the compiler cannot drop or move it), then the epoch word; on the reading side the payload loads are
// issued only after the epoch load has returned the expected value (tests/test_p2p_codeobj.py asserts both in the code object).  Failure handling as in p2p_barrier (p2p_exchange.h): a
// context whose status word is set pushes PUSH_POISON instead of its epoch and does not wait; a poll that times out or reads
// a poisoned epoch sets the status word and poisons both parities of its slots in every peer's inbox.
constexpr unsigned long long PUSH_POISON = 1ull << 63;
__device__ __forceinline__ void stats_push_exchange(const StatsPush& push, double v0, double v1, double (*peer)[2]) {
  if ((int)threadIdx.x < push.world) {
    const int r = threadIdx.x;
    const bool failed = __hip_atomic_load(push.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    const int slot = (int)(push.epoch & 1) * 32;
    unsigned long long* dst = push.inbox[r] + slot + push.rank * 4;
    __hip_atomic_store(dst + 0, (unsigned long long)__double_as_longlong(v0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dst + 1, (unsigned long long)__double_as_longlong(v1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // both payload words acknowledged before the epoch word leaves
    __hip_atomic_store(dst + 2, failed ? (push.epoch | PUSH_POISON) : push.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    peer[r][0] = v0;                // (placeholders of a failed exchange: finite numbers, the run is reported invalid anyway)
    peer[r][1] = v1;
    if (!failed) {
      const unsigned long long* src = push.inbox[push.rank] + slot + r * 4;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      unsigned long long v;
      unsigned int code = 0;
      while ((v = __hip_atomic_load(src + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) < push.epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (__builtin_amdgcn_s_memrealtime() - t0 > push.timeout_ticks) {
          code = 1u;                // P2P_ST_TIMEOUT
          break;
        }
      }
      if (!code && (v & PUSH_POISON)) code = 2u;      // P2P_ST_POISONED
      if (code) {
        atomicOr(push.status, code);
        if (push.status_host) __hip_atomic_store(push.status_host, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(push.inbox[r] + push.rank * 4 + 2, push.epoch | PUSH_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(push.inbox[r] + 32 + push.rank * 4 + 2, push.epoch | PUSH_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the epoch was seen: the payload loads are issued after it returned)
        peer[r][0] = __longlong_as_double((long long)__hip_atomic_load(src + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        peer[r][1] = __longlong_as_double((long long)__hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
      }
    }
  }
}

// self-test of the push (sae_p2p_init): rank r pushes (pattern, 2 pattern + 1); every rank checks all the triples it received
__global__ void p2p_selftest_push_kernel(StatsPush push, int e, unsigned int* bad) {
  __shared__ double peer[8][2];
  auto val = [](int r, int e) { return (double)(1000 * (r + 1) + 37 * e); };
  stats_push_exchange(push, val(push.rank, e), 2.0 * val(push.rank, e) + 1.0, peer);
  __syncthreads();
  if ((int)threadIdx.x < push.world) {
    const int r = threadIdx.x;
    if (peer[r][0] != val(r, e) || peer[r][1] != 2.0 * val(r, e) + 1.0) atomicAdd(bad, 1u);
  }
}

__global__ __launch_bounds__(1024) void finalize_losses_kernel(const float* l1_part, int n_l1, const float* sq_part,
                                                                int n_sq, float* scal, float* metrics, int64_t M,
                                                                int d, float alpha, const float* cnt_part, int n_cnt,
                                                                const double* gstats, StatsPush push) {
  __shared__ double red[3][16];
  __shared__ double redc[16];
  __shared__ double peer_stats[8][2];
  __shared__ double glob[2];
  double a = 0, b = 0, c = 0;
  if (cnt_part) {
    double m = 0;
#pragma unroll 8
    for (int i = threadIdx.x; i < n_cnt; i += 1024) m += (double)cnt_part[i];
    m = wave_sum_d(m);
    if ((threadIdx.x & 63) == 0) redc[threadIdx.x >> 6] = m;
    __syncthreads();
    double tot = 0;
    for (int k = 0; k < 16; ++k) tot += redc[k];
    const double local = (double)M * d - tot;
    if (push.world > 0) {
      stats_push_exchange(push, local, (double)M, peer_stats);
      __syncthreads();
      if (threadIdx.x == 0) {
        double gc = 0, gr = 0;
        for (int r = 0; r < push.world; ++r) {       // rank order: the same bits on every rank
          gc += peer_stats[r][0];
          gr += peer_stats[r][1];
        }
        glob[0] = gc;
        glob[1] = gr;
        push.gstats_out[0] = gc;
        push.gstats_out[1] = gr;
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const double count = push.world > 0 ? glob[0] : (gstats ? gstats[0] : local);
      scal[0] = (float)count;
      scal[1] = alpha / (float)count;
      scal[2] = 1.0f / (float)(push.world > 0 ? glob[1] : (gstats ? gstats[1] : (double)M));
      scal[3] = (float)local;
    }
    __syncthreads();
  }
  const bool global_norm = gstats != nullptr || push.world > 0;
  const double rows_norm = push.world > 0 ? glob[1] : (gstats ? gstats[1] : (double)M);
  // (unrolled: the loads of eight trips are in flight together, the additions keep their order -- with one dependent load per trip
  // this one-workgroup kernel took 78 us at C4's 163 840 tiles)
#pragma unroll 8
  for (int i = threadIdx.x; i < n_l1; i += 1024) a += (double)l1_part[i];
#pragma unroll 8
  for (int i = threadIdx.x; i < n_sq; i += 1024) {
    b += (double)sq_part[2 * i];
    c += (double)sq_part[2 * i + 1];
  }
  a = wave_sum_d(a);
  b = wave_sum_d(b);
  c = wave_sum_d(c);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = a;
    red[1][threadIdx.x >> 6] = b;
    red[2][threadIdx.x >> 6] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double l1 = 0, sq = 0, pl = 0;
    for (int k = 0; k < 16; ++k) {
      l1 += red[0][k];
      sq += red[1][k];
      pl += red[2][k];
    }
    const double count = (double)scal[0], rows = rows_norm;
    metrics[0] = (float)((double)alpha * (sq / count));
    metrics[1] = (float)(l1 / rows);
    metrics[2] = (float)(pl / (rows * d));
    metrics[3] = 0.f;
    metrics[4] = global_norm ? scal[3] : scal[0];
    metrics[5] = metrics[6] = metrics[7] = 0.f;
  }
}

// per-block sum of (scale*g)^2 over the flat gradient
__global__ __launch_bounds__(256) void gnorm_partial_kernel(const float* __restrict__ grad, int64_t n4, float scale,
                                                             double* __restrict__ part) {
  __shared__ double red[4];
  double s = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 g = reinterpret_cast<const f32x4*>(grad)[i] * scale;
    s += (double)(g[0] * g[0]) + (double)(g[1] * g[1]) + (double)(g[2] * g[2]) + (double)(g[3] * g[3]);
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// per-feature max over rows of the (non-negative) latent c[M_p][n_p]  (validate(): train_sae.py:176-178)
__global__ __launch_bounds__(256) void latent_colmax_kernel(const bf16_t* __restrict__ c, int* __restrict__ out_bits,
                                                             int64_t M, int n_p, int rows_per_block) {
  // grid (n_p/256, ceil(M/rows_per_block)); thread -> one column; non-negative floats order like ints
  const int col = blockIdx.x * 256 + threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  float m = 0.f;
  for (int64_t r = r0; r < r1; ++r) m = fmaxf(m, (float)c[r * n_p + col]);
  atomicMax(out_bits + col, __float_as_int(m));
}

// the same into a caller-owned row of exactly n (un-padded) columns
__global__ __launch_bounds__(256) void latent_colmax_bounded_kernel(const bf16_t* __restrict__ c, int* __restrict__ out_bits,
                                                                     int64_t M, int n_p, int n, int rows_per_block) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= n) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  float m = 0.f;
  for (int64_t r = r0; r < r1; ++r) m = fmaxf(m, (float)c[r * n_p + col]);
  atomicMax(out_bits + col, __float_as_int(m));
}

struct OptArgs {
  float lr, grad_scale, clip_thresh, weight_decay;
  float beta1, beta2, eps;
  float one_minus_beta1, one_minus_beta2;
  float bc1, bc2_sqrt;      // 1 - beta1^t, sqrt(1 - beta2^t)
  float step_size;          // Adam: lr / bc1
  float rect;               // RAdam rectification term (valid if rectify)
  int rectify;              // RAdam: rho_t > 5
  int is_radam;
  int scale_metrics;        // 1: the loss scalars in the gradient buffer are a data-parallel SUM that has not been averaged yet
};

// clip_grad_norm_ (coef = min(1, thresh/(norm+1e-6))) fused with the torch single-tensor
// Adam / RAdam update (op order of torch/optim/{adam,radam}.py kept for fp32 agreement).
// cast (TopK): up to two ranges [off4, off4 + len4) (float4 units) of the parameter vector whose UPDATED values are also written
// as bf16 (the GEMM / gather copies of W_enc and W_dec: the next step then needs no cast pass over them).
struct OptCast {
  int64_t off4[2], len4[2];
  bf16_t* dst[2];
};

// One torch single-tensor Adam / RAdam update of four elements (the op order of torch/optim/{adam,radam}.py, one rounding per
// torch op: no fused multiply-adds across them) -- the ONE definition the three optimizer kernels share, so that they agree
// to the bit.
__device__ __forceinline__ void adam_update4(const f32x4 g, f32x4& pv, f32x4& mv, f32x4& vv, const OptArgs& a, const float coef) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma clang fp contract(off)
    float gj = (g[j] * a.grad_scale) * coef;
    if (a.is_radam && a.weight_decay != 0.f) gj = gj + a.weight_decay * pv[j];
    mv[j] = mv[j] + a.one_minus_beta1 * (gj - mv[j]);
    vv[j] = vv[j] * a.beta2;
    vv[j] = vv[j] + (a.one_minus_beta2 * gj) * gj;
    if (a.is_radam) {
      const float mh = mv[j] / a.bc1;
      if (a.rectify) {
        const float adaptive = (1.0f / (sqrtf(vv[j]) + a.eps)) * a.bc2_sqrt;
        pv[j] = pv[j] - ((mh * a.lr) * adaptive) * a.rect;
      } else {
        pv[j] = pv[j] - mh * a.lr;
      }
    } else {
      const float denom = sqrtf(vv[j]) / a.bc2_sqrt + a.eps;
      pv[j] = pv[j] + (-a.step_size * mv[j]) / denom;
    }
  }
}

// sqrt(sum of the gradient-norm partials) -> clip coefficient; block (0, 0) also publishes the norm and, after a data-parallel
// sum, turns the loss scalars into means over ranks (the host clears scale_metrics after the first optimizer step that
// follows a forward_backward: a second call must not rescale; TopK's dead_pct rides in the same summed buffer)
__device__ __forceinline__ float clip_coef_and_metrics(const double* __restrict__ gn_part, int n_part, const OptArgs& a,
                                                       float* __restrict__ metrics, double* redd, bool first_block) {
  double s = 0;
  for (int i = threadIdx.x; i < n_part; i += 256) s += gn_part[i];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) redd[threadIdx.x >> 6] = s;
  __syncthreads();
  const float total = sqrtf((float)(redd[0] + redd[1] + redd[2] + redd[3]));
  if (first_block && threadIdx.x == 0) {
    metrics[3] = total;
    if (a.scale_metrics && a.grad_scale != 1.0f) {
      metrics[0] *= a.grad_scale;
      metrics[1] *= a.grad_scale;
      metrics[2] *= a.grad_scale;
      metrics[5] *= a.grad_scale;
    }
  }
  return fminf(a.clip_thresh / (total + 1e-6f), 1.0f);
}

__global__ __launch_bounds__(256) void optimizer_kernel(float* __restrict__ p, float* __restrict__ m,
                                                         float* __restrict__ v, const float* __restrict__ grad,
                                                         int64_t n4, const double* __restrict__ gn_part, int n_part,
                                                         OptArgs a, float* __restrict__ metrics, OptCast cast) {
  __shared__ double red[4];
  const float coef = clip_coef_and_metrics(gn_part, n_part, a, metrics, red, blockIdx.x == 0);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 g = reinterpret_cast<const f32x4*>(grad)[i];
    f32x4 pv = reinterpret_cast<f32x4*>(p)[i];
    f32x4 mv = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vv = reinterpret_cast<f32x4*>(v)[i];
    adam_update4(g, pv, mv, vv, a, coef);
    reinterpret_cast<f32x4*>(p)[i] = pv;
    reinterpret_cast<f32x4*>(m)[i] = mv;
    reinterpret_cast<f32x4*>(v)[i] = vv;
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (cast.dst[q] && i >= cast.off4[q] && i < cast.off4[q] + cast.len4[q])
        reinterpret_cast<bf16x4*>(cast.dst[q])[i - cast.off4[q]] = bf16x4{(bf16_t)pv[0], (bf16_t)pv[1], (bf16_t)pv[2], (bf16_t)pv[3]};
  }
}

// The same update for the L1 weight matrix W[d_p][n_p], tiled like colnorm_partial_kernel (grid (n_p/128, d_p/32)), which
// ALSO leaves the per-(32-row slab, column) sums of squares of the UPDATED weights in cn_part -- bit-identical to what
// colnorm_partial_kernel would compute from them -- so that the next forward starts with normalize_cast at once (one 5 us
// kernel less per training step).  The bias tail [nW, nW + n_p) is updated by the blocks of an extra grid row, flat.
__global__ __launch_bounds__(256) void optimizer_l1_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                            const float* __restrict__ grad, int d_p, int n_p,
                                                            const double* __restrict__ gn_part, int n_part, OptArgs a,
                                                            float* __restrict__ metrics, float* __restrict__ cn_part) {
  __shared__ double redd[4];
  __shared__ float red[8][128];
  const float coef = clip_coef_and_metrics(gn_part, n_part, a, metrics, redd, blockIdx.x == 0 && blockIdx.y == 0);
  auto update4 = [&](int64_t o) -> f32x4 {
    const f32x4 g = *reinterpret_cast<const f32x4*>(grad + o);
    f32x4 pv = *reinterpret_cast<f32x4*>(p + o), mv = *reinterpret_cast<f32x4*>(m + o), vv = *reinterpret_cast<f32x4*>(v + o);
    adam_update4(g, pv, mv, vv, a, coef);
    *reinterpret_cast<f32x4*>(p + o) = pv;
    *reinterpret_cast<f32x4*>(m + o) = mv;
    *reinterpret_cast<f32x4*>(v + o) = vv;
    return pv;
  };
  const int t = threadIdx.x, tx = t & 31, ty = t >> 5;
  if ((int)blockIdx.y == d_p / 32) {      // the bias: 128 columns per block, threads 0..31
    if (t < 32) update4((int64_t)d_p * n_p + blockIdx.x * 128 + 4 * t);
    return;
  }
  const int col = blockIdx.x * 128 + 4 * tx, r0 = blockIdx.y * 32;
  f32x4 ss = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x4 pv = update4((int64_t)(r0 + ty + 8 * i) * n_p + col);
#pragma unroll
    for (int q = 0; q < 4; ++q) ss[q] = __builtin_fmaf(pv[q], pv[q], ss[q]);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) red[ty][4 * tx + q] = ss[q];
  __syncthreads();
  if (t < 128) cn_part[(int64_t)blockIdx.y * n_p + blockIdx.x * 128 + t] = colnorm_tree8(red, t);
}

// The L1 update for d_p <= 384 with the NEXT forward's weight preparation folded in: a workgroup owns OPTC_COLS columns and ALL d_p
// rows of W, so after the update it holds whole columns -- it forms their norms (the canonical partial sums above, summed
// over the slabs in order: the very value colnorm_partial + normalize_cast would produce), and writes the bf16 copies of the
// NORMALISED weights (Wb [d_p][n_p], Wt [n_p][d_p]) that the next forward / backward read.  The fp32 master keeps the
// un-normalised update -- what the reference holds after optimizer.step() (train_sae.py:450) until its next forward
// normalises in place (l1autoencoder.py:71-73) -- and the denominators go to cnorm[n_p]: the NEXT update divides by them
// while loading (norm_on_load), observers of the master normalise it in place first (normalize_inplace_kernel).  A training
// step then needs neither colnorm_partial nor normalize_cast: 5 kernels -> 4.
//   grid n_p / OPTC_COLS + n_p / 128 (the extra blocks update the bias, flat).  OPTC_COLS = 16 columns and 512 threads per
//   workgroup: a thread updates d_p / 128 <= 3 rows of four columns and has all their loads in flight at once (with 256 threads
//   and 12 dependent load-update-store rounds per thread the kernel took 25 us), and C2's 3072 columns make 192 workgroups
//   (32 columns: 96 workgroups, each CU then moves 400 KB through its own 64 B/clk path: 14.5 us).
//   dynamic LDS (d_p x OPTC_COLS + d_p / 32 x 8 x OPTC_COLS + OPTC_COLS) x 4 B
constexpr int OPTC_COLS = 16, OPTC_THREADS = 512, OPTC_MAX_PASS = 3;
constexpr int OPTC_TPR = OPTC_COLS / 4;                       // threads per row in the update mapping
constexpr int OPTC_ROWS = OPTC_THREADS / OPTC_TPR;            // rows per pass
static_assert(OPTC_ROWS == 128, "rows ty + 128 i: d_p is a multiple of 128");
__host__ __device__ constexpr int optc_lds_bytes(int d_p) { return (d_p * OPTC_COLS + (d_p / 32) * 8 * OPTC_COLS + OPTC_COLS) * 4; }

__global__ __launch_bounds__(OPTC_THREADS) void optimizer_l1_cols_kernel(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v,
                                                                          const float* __restrict__ grad, int d_p, int n_p,
                                                                          const double* __restrict__ gn_part, int n_part, OptArgs a,
                                                                          float* __restrict__ metrics, float* __restrict__ cnorm,
                                                                          int norm_on_load, bf16_t* __restrict__ Wb, bf16_t* __restrict__ Wt) {
  extern __shared__ __attribute__((aligned(16))) float optc_smem[];
  __shared__ double redd[OPTC_THREADS / 64];
  const int t = threadIdx.x, nslab = d_p / 32, npass = d_p / 128;
  const int ncb = n_p / OPTC_COLS;
  const bool bias_block = (int)blockIdx.x >= ncb;
  float* pn = optc_smem;                              // [d_p][OPTC_COLS] updated weights of this workgroup's columns
  float* red = pn + d_p * OPTC_COLS;                  // [nslab][8][OPTC_COLS]
  float* den = red + nslab * 8 * OPTC_COLS;           // [OPTC_COLS]
  // Column block of this workgroup, XCD-major (round 5): a workgroup touches 64-byte pieces of fp32 rows (16 columns) and 32-byte pieces of the
  // bf16 copy, i.e. half / a quarter of a 128-byte line, and workgroup b runs on XCD b % 8 -- with column block = blockIdx the other half of
  // every line was fetched (and written back) by ANOTHER XCD's L2.  Now XCD x owns the contiguous column blocks [x ncb / 8, (x + 1) ncb / 8)
  // and neighbours in a line are dispatched 8 block indices apart, i.e. together.  (-DOPTC_NO_XCD_MAP = the old numbering; n_p is a
  // multiple of 128, so ncb is a multiple of 8.)
#ifndef OPTC_NO_XCD_MAP
  const int cblk = bias_block ? 0 : ((int)blockIdx.x & 7) * (ncb >> 3) + ((int)blockIdx.x >> 3);
#else
  const int cblk = blockIdx.x;
#endif
  const int col0 = cblk * OPTC_COLS;
  const int tx = t % OPTC_TPR, ty = t / OPTC_TPR;      // update / Wb mapping: columns 4 tx .. 4 tx + 3, rows ty + 128 i
  // the loads of the update first: they do not depend on the clip coefficient
  f32x4 dn = {1.f, 1.f, 1.f, 1.f};
  f32x4 g[OPTC_MAX_PASS], pv[OPTC_MAX_PASS], mv[OPTC_MAX_PASS], vv[OPTC_MAX_PASS];
  if (!bias_block) {
    if (norm_on_load) dn = *reinterpret_cast<const f32x4*>(cnorm + col0 + 4 * tx);
#pragma unroll
    for (int i = 0; i < OPTC_MAX_PASS; ++i)
      if (i < npass) {
        const int64_t o = (int64_t)(128 * i + ty) * n_p + col0 + 4 * tx;
        g[i] = *reinterpret_cast<const f32x4*>(grad + o);
        pv[i] = *reinterpret_cast<f32x4*>(p + o);
        mv[i] = *reinterpret_cast<f32x4*>(m + o);
        vv[i] = *reinterpret_cast<f32x4*>(v + o);
      }
  }
  // clip coefficient (every workgroup sums the gradient-norm partials in the same order)
  float coef;
  {
    double sg = 0;
    for (int i = t; i < n_part; i += OPTC_THREADS) sg += gn_part[i];
    sg = wave_sum_d(sg);
    if ((t & 63) == 0) redd[t >> 6] = sg;
    __syncthreads();
    double tot = 0;
#pragma unroll
    for (int w = 0; w < OPTC_THREADS / 64; ++w) tot += redd[w];
    const float total = sqrtf((float)tot);
    if (blockIdx.x == 0 && t == 0) {
      metrics[3] = total;
      if (a.scale_metrics && a.grad_scale != 1.0f) {
        metrics[0] *= a.grad_scale;
        metrics[1] *= a.grad_scale;
        metrics[2] *= a.grad_scale;
        metrics[5] *= a.grad_scale;
      }
    }
    coef = fminf(a.clip_thresh / (total + 1e-6f), 1.0f);
  }
  if (bias_block) {                       // the bias: 128 elements per block, threads 0..31
    if (t < 32) {
      const int64_t o = (int64_t)d_p * n_p + ((int)blockIdx.x - ncb) * 128 + 4 * t;
      const f32x4 gb = *reinterpret_cast<const f32x4*>(grad + o);
      f32x4 pb = *reinterpret_cast<f32x4*>(p + o), mb = *reinterpret_cast<f32x4*>(m + o), vb = *reinterpret_cast<f32x4*>(v + o);
      adam_update4(gb, pb, mb, vb, a, coef);
      *reinterpret_cast<f32x4*>(p + o) = pb;
      *reinterpret_cast<f32x4*>(m + o) = mb;
      *reinterpret_cast<f32x4*>(v + o) = vb;
    }
    return;
  }
  // ---- update
#pragma unroll
  for (int i = 0; i < OPTC_MAX_PASS; ++i)
    if (i < npass) {
      const int row = 128 * i + ty;
      const int64_t o = (int64_t)row * n_p + col0 + 4 * tx;
      if (norm_on_load) {                 // the in-place normalisation the last forward stands for (W / max(||col||, 1e-12))
#pragma unroll
        for (int q = 0; q < 4; ++q) pv[i][q] = pv[i][q] / dn[q];
      }
      adam_update4(g[i], pv[i], mv[i], vv[i], a, coef);
      *reinterpret_cast<f32x4*>(p + o) = pv[i];
      *reinterpret_cast<f32x4*>(m + o) = mv[i];
      *reinterpret_cast<f32x4*>(v + o) = vv[i];
      *reinterpret_cast<f32x4*>(pn + row * OPTC_COLS + 4 * tx) = pv[i];
    }
  __syncthreads();
  // ---- column norms, canonical order: a (slab, y8) pair -> rows y8, y8 + 8, y8 + 16, y8 + 24 of the slab, for column cc
  {
    const int cc = t % OPTC_COLS;
    for (int pr = t / OPTC_COLS; pr < nslab * 8; pr += OPTC_THREADS / OPTC_COLS) {
      const int i = pr >> 3, y8 = pr & 7;
      float ss = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float w = pn[(32 * i + y8 + 8 * k) * OPTC_COLS + cc];
        ss = __builtin_fmaf(w, w, ss);
      }
      red[pr * OPTC_COLS + cc] = ss;
    }
  }
  __syncthreads();
  if (t < OPTC_COLS) {
    float tot = 0.f;
    for (int i = 0; i < nslab; ++i) {
      const float* r = red + i * 8 * OPTC_COLS + t;
      tot += ((r[0] + r[OPTC_COLS]) + (r[2 * OPTC_COLS] + r[3 * OPTC_COLS])) + ((r[4 * OPTC_COLS] + r[5 * OPTC_COLS]) + (r[6 * OPTC_COLS] + r[7 * OPTC_COLS]));
    }
    const float dnm = fmaxf(sqrtf(tot), 1e-12f);
    den[t] = dnm;
    cnorm[col0 + t] = dnm;
  }
  __syncthreads();
  // ---- bf16 copies of the normalised weights: Wb row-major (8 B per thread and row), Wt transposed (16-B pieces of a column)
  {
    const f32x4 dn = *reinterpret_cast<const f32x4*>(den + 4 * tx);
#pragma unroll
    for (int i = 0; i < OPTC_MAX_PASS; ++i)
      if (i < npass) {
        const int row = 128 * i + ty;
        const f32x4 w = *reinterpret_cast<const f32x4*>(pn + row * OPTC_COLS + 4 * tx);
        *reinterpret_cast<bf16x4*>(Wb + (int64_t)row * n_p + col0 + 4 * tx) =
            bf16x4{(bf16_t)(w[0] / dn[0]), (bf16_t)(w[1] / dn[1]), (bf16_t)(w[2] / dn[2]), (bf16_t)(w[3] / dn[3])};
      }
    const int cc = t % OPTC_COLS;                     // column cc, pieces t / OPTC_COLS + k OPTC_THREADS / OPTC_COLS (8 rows = 16 B each)
    const float dc = den[cc];
    for (int pc = t / OPTC_COLS; pc < d_p / 8; pc += OPTC_THREADS / OPTC_COLS) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (bf16_t)(pn[(8 * pc + e) * OPTC_COLS + cc] / dc);
      *reinterpret_cast<bf16x8*>(Wt + (int64_t)(col0 + cc) * d_p + 8 * pc) = o;
    }
  }
}

// W[r][c] /= cnorm[c]: the in-place normalisation a forward stands for, carried out for an observer of the fp32 master
// (sae_get_params, a second forward without an update in between, sae_decode) -- the same division normalize_cast performs.
__global__ __launch_bounds__(256) void normalize_inplace_kernel(float* __restrict__ W, const float* __restrict__ cnorm, int64_t n4, int n_p) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int col = (int)((4 * i) % n_p);
    f32x4 w = reinterpret_cast<f32x4*>(W)[i];
    const f32x4 dn = *reinterpret_cast<const f32x4*>(cnorm + col);
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = w[q] / dn[q];
    reinterpret_cast<f32x4*>(W)[i] = w;
  }
}

// ------------------------------------------------------------------------------------------
// inference helpers (sae_decode)
// ------------------------------------------------------------------------------------------
// latent [M][ld] (fp32 or bf16) -> zero-padded bf16 GEMM operand dst[M_p][n_p]
template <typename T>
__global__ __launch_bounds__(256) void pad_latent_kernel(const T* __restrict__ src, int64_t ld, int64_t M, int n,
                                                          bf16_t* __restrict__ dst, int64_t M_p, int n_p) {
  const int64_t total = M_p * (int64_t)n_p;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / n_p;
    const int c = (int)(i - r * n_p);
    dst[i] = (r < M && c < n) ? (bf16_t)(float)src[r * ld + c] : (bf16_t)0.f;
  }
}

// plain fp32 store of a GEMM tile (+ optional per-column bias), rows >= M and columns >= d dropped
struct EpiStoreF32 {
  float* out;           // [M][d]
  const float* bias;    // [d] or null
  int64_t M;
  int d;
  __device__ void tile_begin(int, int, int) {}
  struct Pre {};
  __device__ Pre prefetch(int, int) const { return Pre{}; }
  __device__ void apply(int row, int col, f32x4 v, const Pre&) {
    if (row >= M) return;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (col + j < d) out[(int64_t)row * d + col + j] = bf16_round(v[j]) + (bias ? bias[col + j] : 0.f);
  }
  __device__ void tile_end(float*) {}
};
