// fp8 (OCP e4m3fn) operand preparation and epilogues for the encoder / decoder GEMMs of the L1 SAE
// (BASELINE configs[4]: "fp8 MFMA enc/dec with bf16 accumulate"; reference arithmetic being replaced:
// src/models/l1autoencoder.py:74,84).
//
// Quantisation is per tensor with POWER-OF-TWO scales (exact to apply and to undo):
//   W8 = e4m3(W * 2^8)            column-normalised |w| <= 1, so 256 |w| <= 256 < 448 (e4m3 max): never saturates;
//   x8 = e4m3(bf16(x) * s_x)      s_x = 2^floor(log2(448 / max|x|)) from an amax pass over the batch;
//   c8 = e4m3(c * s_c)            s_c = 2^floor(log2(448 / bound)), bound = max_row ||x_row||_2 + max(0, max_j b_j) >= every
//                                 c_mj (Cauchy-Schwarz with unit-norm columns): known BEFORE the encoder GEMM runs, so the
//                                 latent is quantised in that GEMM's epilogue, and it cannot saturate either.
// The MFMA accumulates the scaled products in fp32; the epilogue multiplies by 1 / (s_a s_b) and rounds to bf16 where the
// bf16 path rounds (pre-activation before the fp32 bias add; x_hat).  The latent is stored twice: bf16 for the (bf16)
// backward GEMMs, e4m3 for the decoder GEMM.  The backward, the fp32 master weights and the optimizer are unchanged.
#pragma once
#include "l1_kernels.h"

constexpr float FP8_E4M3_MAX = 448.0f;
constexpr float FP8_W_SCALE = 256.0f;

// scal8: [0] s_x  [1] 1/(s_x s_w)  [2] s_c  [3] 1/(s_c s_w)  [4] max|x|  [5] latent bound  [6] s_g  [7] 1/(s_g s_w)
// (s_g: SAE_PREC_FP8_BWD only -- the scale of the quantised dx_hat, the A operand of the fp8 dpre GEMM)
enum { S8_SX = 0, S8_INV_ENC = 1, S8_SC = 2, S8_INV_DEC = 3, S8_AMAX_X = 4, S8_C_BOUND = 5, S8_SG = 6, S8_INV_DPRE = 7, S8_COUNT = 8 };

__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
  unsigned r = 0;
  r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, r, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return r;
}

// largest power of two p with p * v <= 448 (v > 0), i.e. 2^floor(log2(448 / v)); exact (frexp, no libm log)
__device__ __forceinline__ float fp8_pow2_scale(float v) {
  if (!(v > 0.f)) return 1.0f;
  int e;
  (void)frexpf(FP8_E4M3_MAX / v, &e);        // 448 / v = m 2^e, m in [0.5, 1)  ->  floor(log2) = e - 1
  e -= 1;
  if (e > 100) e = 100;
  if (e < -100) e = -100;
  return ldexpf(1.0f, e);
}

// per-block partials of max |x| and max row sum of squares over the bf16 GEMM copy xb[M_p][d_p] (padding is zero);
// one wave per row, 4 rows per block pass
__global__ __launch_bounds__(256) void fp8_x_stats_kernel(const bf16_t* __restrict__ xb, int64_t M_p, int d_p,
                                                          float* __restrict__ part /* [grid][2] */) {
  __shared__ float red[2][4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float amax = 0.f, rmax = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < M_p; row += (int64_t)gridDim.x * 4) {
    const bf16x8* p = reinterpret_cast<const bf16x8*>(xb + row * d_p);
    float ss = 0.f;
    for (int i = lane; i < d_p / 8; i += 64) {
      const bf16x8 v = p[i];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = (float)v[j];
        amax = fmaxf(amax, fabsf(f));
        ss += f * f;
      }
    }
    ss = wave_sum(ss);
    rmax = fmaxf(rmax, ss);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  if (lane == 0) {
    red[0][w] = amax;
    red[1][w] = rmax;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    part[2 * blockIdx.x + 1] = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
  }
}

__global__ __launch_bounds__(256) void fp8_scales_kernel(const float* __restrict__ part, int nparts, const float* __restrict__ bias,
                                                         int n, float* __restrict__ scal8) {
  __shared__ float red[3][4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float amax = 0.f, rmax = 0.f, bmax = 0.f;
  // (one workgroup: unrolled so that eight trips' loads are in flight together -- 320 dependent trips over the bias were 76 us at C5)
#pragma unroll 8
  for (int i = threadIdx.x; i < nparts; i += 256) {
    amax = fmaxf(amax, part[2 * i]);
    rmax = fmaxf(rmax, part[2 * i + 1]);
  }
#pragma unroll 8
  for (int i = threadIdx.x; i < n; i += 256) bmax = fmaxf(bmax, bias[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    amax = fmaxf(amax, __shfl_xor(amax, o, 64));
    rmax = fmaxf(rmax, __shfl_xor(rmax, o, 64));
    bmax = fmaxf(bmax, __shfl_xor(bmax, o, 64));
  }
  if (lane == 0) {
    red[0][w] = amax;
    red[1][w] = rmax;
    red[2][w] = bmax;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    amax = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
    rmax = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    bmax = fmaxf(fmaxf(red[2][0], red[2][1]), fmaxf(red[2][2], red[2][3]));
    // the GEMM sees the QUANTISED operands: every |x8| / s_x and |w8| / s_w may exceed its source by 2^-4 relative, so
    // the product of the two norms by up to 1.0625^2 = 1.13 (plus the bf16 rounding of the result): 15 % headroom
    const float bound = (sqrtf(rmax) + bmax) * 1.15f;
    const float sx = fp8_pow2_scale(amax), sc = fp8_pow2_scale(bound);
    scal8[S8_SX] = sx;
    scal8[S8_INV_ENC] = 1.0f / (sx * FP8_W_SCALE);
    scal8[S8_SC] = sc;
    scal8[S8_INV_DEC] = 1.0f / (sc * FP8_W_SCALE);
    scal8[S8_AMAX_X] = amax;
    scal8[S8_C_BOUND] = bound;
  }
}

// s_g = 2^floor(log2(448 / max|dx_hat|)) from the per-block maxima of fp8_x_stats_kernel run over dx_hat
__global__ __launch_bounds__(256) void fp8_g_scale_kernel(const float* __restrict__ part, int nparts, float* __restrict__ scal8) {
  __shared__ float red[4];
  float amax = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) amax = fmaxf(amax, part[2 * i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float sg = fp8_pow2_scale(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
    scal8[S8_SG] = sg;
    scal8[S8_INV_DPRE] = 1.0f / (sg * FP8_W_SCALE);
  }
}

// x8[M_p][d_p] = e4m3(xb * scal8[sidx])      (sidx = S8_SX: the activations; S8_SG: dx_hat)
__global__ __launch_bounds__(256) void fp8_quant_x_kernel(const bf16_t* __restrict__ xb, unsigned char* __restrict__ x8, int64_t n8,
                                                          const float* __restrict__ scal8, int sidx = S8_SX) {
  const float sx = scal8[sidx];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const bf16x8 v = reinterpret_cast<const bf16x8*>(xb)[i];
    uint2 o;
    o.x = pack4_fp8((float)v[0] * sx, (float)v[1] * sx, (float)v[2] * sx, (float)v[3] * sx);
    o.y = pack4_fp8((float)v[4] * sx, (float)v[5] * sx, (float)v[6] * sx, (float)v[7] * sx);
    reinterpret_cast<uint2*>(x8)[i] = o;
  }
}

// fp8 copies of the (already normalised) fp32 master weights: W8[d_p][n_p] (K = n contiguous, decoder operand) and
// W8t[n_p][d_p] (K = d contiguous, encoder operand), both e4m3(W * 2^8).  Grid (n_p/64, d_p/64), one 64x64 tile.
__global__ __launch_bounds__(256) void fp8_cast_w_kernel(const float* __restrict__ W, unsigned char* __restrict__ W8,
                                                         unsigned char* __restrict__ W8t, int d_p, int n_p) {
  __shared__ __attribute__((aligned(16))) unsigned char tT[64][80];   // [col][row] of the tile
  const int t = threadIdx.x, tx = t & 15, ty = t >> 4;                 // thread -> 4 columns 4 tx.., rows ty, ty+16, ...
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = ty + 16 * i;
    const f32x4 v = *reinterpret_cast<const f32x4*>(W + (int64_t)(r0 + r) * n_p + c0 + 4 * tx);
    const unsigned pk = pack4_fp8(v[0] * FP8_W_SCALE, v[1] * FP8_W_SCALE, v[2] * FP8_W_SCALE, v[3] * FP8_W_SCALE);
    *reinterpret_cast<unsigned*>(W8 + (int64_t)(r0 + r) * n_p + c0 + 4 * tx) = pk;
#pragma unroll
    for (int j = 0; j < 4; ++j) tT[4 * tx + j][r] = (unsigned char)(pk >> (8 * j));
  }
  __syncthreads();
  // 64 columns x 64 rows -> W8t[col][r0 .. r0+64): 4 pieces of 16 B per column, 256 pieces
  const int c = t >> 2, sgm = t & 3;
  *reinterpret_cast<u32x4*>(W8t + (int64_t)(c0 + c) * d_p + r0 + sgm * 16) = *reinterpret_cast<const u32x4*>(&tT[c][sgm * 16]);
}

// encoder epilogue: c = relu(bf16(acc / (s_x s_w)) + b) (l1autoencoder.py:74), rows >= M forced to 0;
// stored as bf16 (backward) and as e4m3(c s_c) (decoder operand); L1 partial sum per tile.
struct EpiEnc8 {
  static constexpr bool ROUNDS_BF16_FIRST = true;     // gemm256_fp8.h: s_x s_w is a power of two, bf16(acc) / (s_x s_w) == bf16(acc / (s_x s_w))
  bf16_t* c;            // [M_p][n_p]
  unsigned char* c8;    // [M_p][n_p]
  const float* bias;    // [n_p]
  const float* scal8;
  float* l1_part;       // [tiles]
  int64_t M;
  int n_p, nbn;
  float l1, inv, sc;
  int tile_id;
  __device__ void tile_begin(int row0, int col0, int) {
    l1 = 0.f;
    inv = scal8[S8_INV_ENC];
    sc = scal8[S8_SC];
    tile_id = (row0 / GEMM_BM) * nbn + col0 / GEMM_BN;
  }
  struct Pre { f32x4 b; };
  __device__ Pre prefetch(int, int col) const { return Pre{*reinterpret_cast<const f32x4*>(bias + col)}; }
  __device__ void apply(int row, int col, f32x4 v, const Pre& pre) {
    const f32x4 b = pre.b;
    bf16x4 o;
    float cv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      cv[j] = fmaxf(bf16_round(v[j] * inv) + b[j], 0.f);
      if (row >= M) cv[j] = 0.f;
      l1 += cv[j];
      o[j] = (bf16_t)cv[j];
    }
    EPI_STORE(reinterpret_cast<bf16x4*>(c + (int64_t)row * n_p + col), o);
    EPI_STORE(reinterpret_cast<unsigned*>(c8 + (int64_t)row * n_p + col), pack4_fp8(cv[0] * sc, cv[1] * sc, cv[2] * sc, cv[3] * sc));
  }
  // gemm256.h (round 4): eight columns per thread -- a 16-byte bf16 and an 8-byte e4m3 store per row instead of two 8-byte and
  // two 4-byte ones (this epilogue writes the latent twice, so its store instructions count double)
  static constexpr bool WIDE8 = true;
  __device__ void apply8(int row, int col, f32x4 v0, f32x4 v1, const Pre& p0, const Pre& p1) {
    bf16x8 o;
    float cv[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      cv[j] = fmaxf(bf16_round(v0[j] * inv) + p0.b[j], 0.f);
      cv[4 + j] = fmaxf(bf16_round(v1[j] * inv) + p1.b[j], 0.f);
      if (row >= M) cv[j] = cv[4 + j] = 0.f;
      l1 += cv[j] + cv[4 + j];
      o[j] = (bf16_t)cv[j];
      o[4 + j] = (bf16_t)cv[4 + j];
    }
    EPI_STORE(reinterpret_cast<bf16x8*>(c + (int64_t)row * n_p + col), o);
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    const u32x2 q = {pack4_fp8(cv[0] * sc, cv[1] * sc, cv[2] * sc, cv[3] * sc), pack4_fp8(cv[4] * sc, cv[5] * sc, cv[6] * sc, cv[7] * sc)};
    EPI_STORE(reinterpret_cast<u32x2*>(c8 + (int64_t)row * n_p + col), q);
  }
  __device__ void tile_end(float* scratch) {
    const float s = block_sum_256_lds(l1, scratch);
    if ((threadIdx.x & 255) == 0) l1_part[tile_id] = s;
  }
  // ---- streaming form (gemm256s.h / gemm256_fp8.h): as EpiEnc's, with the operand scales undone (a power of two: exact on the
  // bf16-rounded accumulator) and the second, e4m3 copy of the latent
  static constexpr bool STREAM = true;
  struct SPre {};
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  f32x2 sbp[4], invp, scp, l1p;      // bias pairs, 1 / (s_x s_w), s_c, the L1 sum as a pair of partial sums
  __device__ void s_begin() { l1 = 0.f; l1p = f32x2{0.f, 0.f}; }
  __device__ int64_t s_rows() const { return M; }
  __device__ void s_tile(int, int col) {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + col), b1 = *reinterpret_cast<const f32x4*>(bias + col + 4);
    sbp[0] = f32x2{b0[0], b0[1]}; sbp[1] = f32x2{b0[2], b0[3]}; sbp[2] = f32x2{b1[0], b1[1]}; sbp[3] = f32x2{b1[2], b1[3]};
    inv = scal8[S8_INV_ENC];
    sc = scal8[S8_SC];
    invp = f32x2{inv, inv};
    scp = f32x2{sc, sc};
  }
  __device__ SPre s_prefetch(int, int) const { return SPre{}; }
  // Round 6: packed arithmetic, 28 vector instructions per 8 latents (the element-wise form, with its bf16 re-rounding of v / (s_x s_w) spelled
  // out as convert + shift, was ~70 -- in a kernel whose K loop is only ten K tiles long, so that the epilogue was a fifth of a tile).  v is the
  // accumulator ALREADY rounded to bf16 and 1 / (s_x s_w) a power of two: v inv is exact, so fma(v, inv, b) IS bf16(v inv) + b, bit for bit.
  template <bool PARTIAL>
  __device__ void s_apply(int row, int col, f32x4 v0, f32x4 v1, const SPre&) {
    f32x2 sv[4] = {__builtin_elementwise_fma(f32x2{v0[0], v0[1]}, invp, sbp[0]), __builtin_elementwise_fma(f32x2{v0[2], v0[3]}, invp, sbp[1]),
                   __builtin_elementwise_fma(f32x2{v1[0], v1[1]}, invp, sbp[2]), __builtin_elementwise_fma(f32x2{v1[2], v1[3]}, invp, sbp[3])};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sv[k] = f32x2{fmaxf(sv[k][0], 0.f), fmaxf(sv[k][1], 0.f)};
      if (PARTIAL && row >= M) sv[k] = f32x2{0.f, 0.f};
      l1p += sv[k];
    }
    // (timing experiment of round 5, `-DFP8_SKIP_BF16`: no bf16 copy of the latent -- the upper bound of what a single latent store buys:
    // encoder 7.38 -> 6.64 ms at C5, profiles/r05_fp8_single_store_bound.txt; as a run-time flag it cost this kernel 6 spill instructions)
#ifndef FP8_SKIP_BF16
    const bf16x8 o = {(bf16_t)sv[0][0], (bf16_t)sv[0][1], (bf16_t)sv[1][0], (bf16_t)sv[1][1],
                      (bf16_t)sv[2][0], (bf16_t)sv[2][1], (bf16_t)sv[3][0], (bf16_t)sv[3][1]};
    EPI_STORE(reinterpret_cast<bf16x8*>(c + (int64_t)row * n_p + col), o);
#endif
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    const f32x2 q0 = sv[0] * scp, q1 = sv[1] * scp, q2 = sv[2] * scp, q3 = sv[3] * scp;
    const u32x2 q = {pack4_fp8(q0[0], q0[1], q1[0], q1[1]), pack4_fp8(q2[0], q2[1], q3[0], q3[1])};
    EPI_STORE(reinterpret_cast<u32x2*>(c8 + (int64_t)row * n_p + col), q);
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

// dpre epilogue of the fp8 dc GEMM (SAE_PREC_FP8_BWD): dpre = (bf16(acc / (s_g s_w)) + 1/M) [c > 0], column sums for db --
// EpiDpre (l1_kernels.h) with the accumulator un-scaled first; the reference arithmetic is the autograd of
// src/models/l1autoencoder.py:74,84, here with e4m3 operands for the dx_hat W product.
struct EpiDpre8 {
  static constexpr bool ROUNDS_BF16_FIRST = true;     // (see EpiEnc8)
  static constexpr int PREFETCH_BATCH = EPI_BATCH_HEAVY;
  const bf16_t* c;      // [M_p][n_p]
  bf16_t* dpre;         // [M_p][n_p]
  float* db_part;       // [M_p / 128][n_p]
  const float* scal;    // scal[2] = 1/M
  const float* scal8;
  int n_p;
  float colsum[4];
  float inv_m, inv;
  int row_tile, col0_;
  __device__ void tile_begin(int row0, int col0, int) {
    colsum[0] = colsum[1] = colsum[2] = colsum[3] = 0.f;
    inv_m = scal[2];
    inv = scal8[S8_INV_DPRE];
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
      const float g = ((float)cv[j] > 0.f) ? (bf16_round(v[j] * inv) + inv_m) : 0.f;
      colsum[j] += g;
      o[j] = (bf16_t)g;
    }
    EPI_STORE(reinterpret_cast<bf16x4*>(dpre + (int64_t)row * n_p + col), o);
  }
  __device__ void tile_end(float* scratch) {
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
  // (no streaming form: next to the fp8 kernel's 64 fragment registers the gate's prefetched latents do not fit without spills;
  // SAE_PREC_FP8_BWD is opt-in and keeps the tile form of gemm256_fp8.h)
};
