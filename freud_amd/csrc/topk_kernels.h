// Kernels of the TopK SAE train step (reference: src/models/topkautoencoder.py:72-151 and the bookkeeping
// in src/scripts/train_sae.py:424-446).  The dense encoder GEMM and the backward GEMMs reuse gemm.h; this
// file holds the epilogues and the non-GEMM stages: activation prep, per-row top-k selection (radix select
// on the bf16 bit patterns in LDS), sparse decode + losses, total variance, dead-latent bookkeeping.
#pragma once
#include "gemm.h"
#include "l1_kernels.h"

// fp32 master -> bf16 GEMM operand copy (both weight matrices, once per step)
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int64_t n8) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    const f32x4 a = reinterpret_cast<const f32x4*>(src)[2 * i], b = reinterpret_cast<const f32x4*>(src)[2 * i + 1];
    bf16x8 o = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
    reinterpret_cast<bf16x8*>(dst)[i] = o;
  }
}

// sae_in = bf16(x - b_dec) (topkautoencoder.py:74), zero padded to [M_p][d_p]; 8 elements per thread, one vector load when
// d % 8 == 0 and x is 16-byte aligned (VEC)
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void topk_prep_x_kernel(const T* __restrict__ x, const float* __restrict__ b_dec,
                                                           bf16_t* __restrict__ xs, int64_t M, int d, int64_t M_p, int d_p) {
  const unsigned int cpr = (unsigned int)d_p >> 3;
  const unsigned int total = (unsigned int)(M_p * cpr);
  for (unsigned int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int64_t row = i / cpr;
    const int c0 = (int)(i - (unsigned int)row * cpr) * 8;
    bf16x8 o;
    if (VEC && row < M && c0 + 8 <= d) {
      const typename Vec8<T>::type v = *reinterpret_cast<const typename Vec8<T>::type*>(x + row * d + c0);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(b_dec + c0), b1 = *reinterpret_cast<const f32x4*>(b_dec + c0 + 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)v[j] - (j < 4 ? b0[j] : b1[j - 4]));
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = 0.f;
        if (row < M && c0 + j < d) v = (float)x[row * d + c0 + j] - b_dec[c0 + j];
        o[j] = (bf16_t)v;
      }
    }
    *reinterpret_cast<bf16x8*>(xs + row * d_p + c0) = o;
  }
}

// total_variance = sum (x - mean over files)^2 (topkautoencoder.py:104-106); x viewed as [B][T*d].
// One thread per (t, feature) column; fixed-order partial sums per block.
template <typename T>
__global__ __launch_bounds__(256) void total_variance_kernel(const T* __restrict__ x, int B, int64_t TD, double* __restrict__ part) {
  __shared__ double red[4];
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double s = 0;
  if (j < TD) {
    float mean = 0.f;
    for (int b = 0; b < B; ++b) mean += (float)x[(int64_t)b * TD + j];
    mean /= (float)B;
    for (int b = 0; b < B; ++b) {
      const float dlt = (float)x[(int64_t)b * TD + j] - mean;
      s += (double)(dlt * dlt);
    }
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// The same for bf16 activations with up to NB files per batch, T*d a multiple of 8 and 16-byte aligned rows: a thread owns COLS (8
// or 4) consecutive columns and keeps all B of their vectors in registers (128 registers either way: NB = 32 x 16 bytes or NB = 64
// x 8 bytes), so the batch is read from HBM once, in wide loads that are all in flight together -- the per-column kernel above
// makes 2 B dependent 2-byte loads per thread and reached 1.2 TB/s (82 us at C3).  Same arithmetic per column (float mean in file
// order, float square, double sum); part[i] still covers columns [256 i, 256 i + 256) (now: a thread's columns in order, then a
// butterfly over the 256 / COLS threads of the part).
constexpr int TV_MAXB = 64;
template <int NB, int COLS>
__global__ __launch_bounds__(256) void total_variance_vec_kernel(const bf16_t* __restrict__ x, int B, int64_t TD, double* __restrict__ part) {
  typedef __attribute__((ext_vector_type(COLS / 2))) unsigned int vec_t;
  const int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * COLS;
  vec_t v[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int q = 0; q < COLS / 2; ++q) v[b][q] = 0u;
    if (b < B && c0 < TD) v[b] = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(x + (int64_t)b * TD + c0));
  }
  auto val = [&](int b, int e) { return __uint_as_float((e & 1) ? (v[b][e >> 1] & 0xFFFF0000u) : (v[b][e >> 1] << 16)); };
  float mean[COLS];
#pragma unroll
  for (int e = 0; e < COLS; ++e) mean[e] = 0.f;
#pragma unroll
  for (int b = 0; b < NB; ++b)
    if (b < B) {
#pragma unroll
      for (int e = 0; e < COLS; ++e) mean[e] += val(b, e);
    }
#pragma unroll
  for (int e = 0; e < COLS; ++e) mean[e] /= (float)B;
  double sc[COLS];
#pragma unroll
  for (int e = 0; e < COLS; ++e) sc[e] = 0.0;
#pragma unroll
  for (int b = 0; b < NB; ++b)
    if (b < B) {
#pragma unroll
      for (int e = 0; e < COLS; ++e) {
        const float dlt = val(b, e) - mean[e];
        sc[e] += (double)(dlt * dlt);
      }
    }
  double s = 0.0;
#pragma unroll
  for (int e = 0; e < COLS; ++e) s += sc[e];
  constexpr int GROUP = 256 / COLS;                 // threads per part (32 or 64: inside one wave)
#pragma unroll
  for (int o = GROUP / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & (GROUP - 1)) == 0 && c0 < TD) part[c0 >> 8] = s;
}

// dead_mask = num_frames_since_fired > threshold (train_sae.py:436-439); tk[0] = num_dead, tk[1] = k_aux,
// tkf[0] = aux scale = min(num_dead / (d/2), 1) (topkautoencoder.py:111-115).  One block.
__global__ __launch_bounds__(1024) void dead_mask_kernel(const long long* __restrict__ nfsf, unsigned char* __restrict__ dead,
                                                          float* __restrict__ did_fire, int n, int n_p, double threshold,
                                                          int d, int* __restrict__ tk, float* __restrict__ tkf) {
  __shared__ int cnt;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  int c = 0;
  for (int i = threadIdx.x; i < n_p; i += 1024) {
    const bool dd = i < n && (double)nfsf[i] > threshold;
    dead[i] = dd;
    did_fire[i] = 0.f;
    c += dd;
  }
  atomicAdd(&cnt, c);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int k_aux_full = d / 2;
    tk[0] = cnt;
    tk[1] = cnt < k_aux_full ? cnt : k_aux_full;
    tkf[0] = fminf((float)cnt / (float)k_aux_full, 1.0f);
  }
}

// encoder epilogue: pre = relu(bf16(acc + bias)) (Linear under autocast: bf16 addmm, one rounding), rows >= M zero
// tmax != null: also the maximum of every (row, 64-column tile) for the tile-driven select (topk_select_tiles_kernel), which
// then reads only the tiles that can hold one of the row's k largest values.
// The arithmetic is done on PACKED bf16 pairs: v_cvt_pk_bf16_f32 rounds two sums at once, ReLU is a signed 16-bit max with 0
// (negative floats are negative as int16), the maximum is an unsigned 16-bit max (values >= 0: the pattern orders like the
// value) -- 12 vector instructions per 4 columns.  The per-row maximum over the 32 threads that share a row is NOT reduced
// per call (five DPP steps + hazard nops each: that was 40 % of this epilogue, itself a third of the GEMM at K = 768): each
// thread keeps its 16 row maxima and tile_end() reduces them through the (by then free) LDS tile, one row per thread.
struct EpiTopkEnc {
  bf16_t* pre;          // [M_p][n_p]
  const float* bias;    // [n_p] fp32 master ALREADY ROUNDED to bf16 (round_bias_kernel: autocast casts it)
  int64_t M;
  int n_p;
  unsigned short* tmax; // [M_p][n_p / 64] bf16 bit patterns (maxima of 64-column tiles), or null
  typedef __attribute__((ext_vector_type(2))) short s16x2;
  typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;
  typedef __attribute__((ext_vector_type(2))) float f32x2;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  unsigned int rmax[16];
  int row0_, col0_;
  bool partial_;        // the tile holds rows >= M (block-uniform)
  __device__ void tile_begin(int row0, int col0, int) {
    row0_ = row0;
    col0_ = col0;
    partial_ = row0 + GEMM_BM > M;
  }
  struct Pre { f32x4 b; };
  __device__ Pre prefetch(int, int col) const { return Pre{*reinterpret_cast<const f32x4*>(bias + col)}; }
  // `it` = index of the call within the tile (0..15), a compile-time constant at the call sites (epi_apply)
  __device__ void apply_it(int it, int row, int col, f32x4 v, const Pre& pf) {
    const f32x4 b = pf.b;
    const s16x2 zero = {0, 0};
    s16x2 p0 = __builtin_bit_cast(s16x2, __builtin_convertvector(f32x2{v[0] + b[0], v[1] + b[1]}, bf16x2));
    s16x2 p1 = __builtin_bit_cast(s16x2, __builtin_convertvector(f32x2{v[2] + b[2], v[3] + b[3]}, bf16x2));
    p0 = __builtin_elementwise_max(p0, zero);
    p1 = __builtin_elementwise_max(p1, zero);
    if (partial_ && row >= M) p0 = p1 = zero;
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    EPI_STORE(reinterpret_cast<u32x2*>(pre + (int64_t)row * n_p + col),
              (u32x2{__builtin_bit_cast(unsigned int, p0), __builtin_bit_cast(unsigned int, p1)}));
    if (tmax) {
      const unsigned int mm = __builtin_bit_cast(unsigned int, __builtin_elementwise_max(__builtin_bit_cast(u16x2, p0), __builtin_bit_cast(u16x2, p1)));
      rmax[it] = max(mm & 0xFFFFu, mm >> 16);
    }
  }
  // scratch: the 128 x 132-float LDS tile of this 256-thread group, free once the apply() loop has read it
  __device__ void tile_end(float* scratch) {
    if (!tmax) return;                                  // (uniform over the launch)
    constexpr int PITCH = 36;                           // 144-byte rows: the 16-byte row reads below are conflict-free
    unsigned int* sc = reinterpret_cast<unsigned int*>(scratch);
    const int t = threadIdx.x & 255;
#pragma unroll
    for (int it = 0; it < 16; ++it) sc[((t >> 5) + 8 * it) * PITCH + (t & 31)] = rmax[it];
    lds_barrier();
    if (t < 128) {
      const u32x4* r = reinterpret_cast<const u32x4*>(sc + t * PITCH);
      unsigned int m[2] = {0u, 0u};                     // entry e of the row = the maximum of columns 4 e .. 4 e + 3
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const u32x4 w = r[q];
        m[q >> 2] = max(max(m[q >> 2], w[0]), max(max(w[1], w[2]), w[3]));
      }
      // two 64-column maxima per row and 128-column sub-tile, one 4-byte store (the row pitch n_p / 64 is even)
      *reinterpret_cast<unsigned int*>(tmax + (int64_t)(row0_ + t) * (n_p >> 6) + (col0_ >> 6)) = m[0] | (m[1] << 16);
    }
  }
  // ---- streaming form (gemm256s.h, fp32 variant: the bias joins the fp32 accumulator before the ONE rounding).  Lane (rq = lane / 4,
  // cp = lane % 4) owns columns 8 cp .. + 7 of each 32-column half j of its wave's 64 columns -- exactly one 64-column tile of
  // tmax -- and rows rq + 16 q + 32 i.  Per row: max over the lane's 8 values of half 0, then of half 1, then over the 4 lanes
  // of the row (two DPP quad swaps); lane cp == 0 stores the 16-bit maximum.
  static constexpr bool STREAM = true;
  static constexpr bool STREAM_F32 = true;
  struct SPre {};
  f32x4 sb[2][2];         // bias of the lane's columns: [half j][first / second four]
  unsigned int smax[8];   // running maximum of row (i, q) over the two halves
  __device__ void s_begin() {}
  __device__ int64_t s_rows() const { return M; }
  __device__ void s_tile(int row0, int col) {      // col = first column of the lane's 8 in half 0
    partial_ = row0 + GEMM_BM > M;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      sb[j][0] = *reinterpret_cast<const f32x4*>(bias + col + 32 * j);
      sb[j][1] = *reinterpret_cast<const f32x4*>(bias + col + 32 * j + 4);
    }
  }
  __device__ void s_apply(int e, int row, int col, f32x4 v0, f32x4 v1) {     // e = 4 i + 2 j + q (compile-time constant)
    const int j = (e >> 1) & 1, iq = 2 * (e >> 2) + (e & 1);
    const s16x2 zero = {0, 0};
    const f32x4 b0 = sb[j][0], b1 = sb[j][1];
    s16x2 p[4];
    p[0] = __builtin_bit_cast(s16x2, __builtin_convertvector(f32x2{v0[0] + b0[0], v0[1] + b0[1]}, bf16x2));
    p[1] = __builtin_bit_cast(s16x2, __builtin_convertvector(f32x2{v0[2] + b0[2], v0[3] + b0[3]}, bf16x2));
    p[2] = __builtin_bit_cast(s16x2, __builtin_convertvector(f32x2{v1[0] + b1[0], v1[1] + b1[1]}, bf16x2));
    p[3] = __builtin_bit_cast(s16x2, __builtin_convertvector(f32x2{v1[2] + b1[2], v1[3] + b1[3]}, bf16x2));
#pragma unroll
    for (int k = 0; k < 4; ++k) p[k] = __builtin_elementwise_max(p[k], zero);
    if (partial_ && row >= M) p[0] = p[1] = p[2] = p[3] = zero;
    EPI_STORE(reinterpret_cast<u32x4*>(pre + (int64_t)row * n_p + col),
              (u32x4{__builtin_bit_cast(unsigned int, p[0]), __builtin_bit_cast(unsigned int, p[1]),
                     __builtin_bit_cast(unsigned int, p[2]), __builtin_bit_cast(unsigned int, p[3])}));
    {   // (no `if (tmax)` and no lane predicate on the store below: either is a branch per call, i.e. a basic-block boundary per 8 latents;
        //  the engine always passes the tile-maximum buffer, and the four lanes of a quad store the same value to the same address)
      const u16x2 a = __builtin_elementwise_max(__builtin_bit_cast(u16x2, p[0]), __builtin_bit_cast(u16x2, p[1]));
      const u16x2 b = __builtin_elementwise_max(__builtin_bit_cast(u16x2, p[2]), __builtin_bit_cast(u16x2, p[3]));
      const unsigned int mm = __builtin_bit_cast(unsigned int, __builtin_elementwise_max(a, b));
      unsigned int m1 = max(mm & 0xFFFFu, mm >> 16);
      if (j == 0) {
        smax[iq] = m1;
      } else {
        m1 = max(m1, smax[iq]);
        m1 = max(m1, (unsigned int)__builtin_amdgcn_update_dpp((int)m1, (int)m1, 0xB1, 0xF, 0xF, false));     // quad_perm [1,0,3,2]
        m1 = max(m1, (unsigned int)__builtin_amdgcn_update_dpp((int)m1, (int)m1, 0x4E, 0xF, 0xF, false));     // quad_perm [2,3,0,1]
        tmax[(int64_t)row * (n_p >> 6) + (col >> 6)] = (unsigned short)m1;
      }
    }
  }
  __device__ void s_tile_end(int, int) {}
  __device__ void s_end(float*) {}
};

// bias rounded to bf16 and kept as float (what the autocast addmm adds): once per step, n_p values
__global__ __launch_bounds__(256) void round_bias_kernel(const float* __restrict__ b, float* __restrict__ out, int n_p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_p) out[i] = bf16_round(b[i]);
}

// ------------------------------------------------------------------------------------------
// per-row top-k: one workgroup per row.  Values are non-negative bf16 (post-ReLU), so their 16-bit patterns
// order like the values: two-level radix select (high byte, low byte) on LDS histograms finds the k-th largest
// pattern `thr`; every element > thr is selected, ties at thr are taken in increasing column order.
// Writes the masked dense row (selected activations, zeros elsewhere) and marks did_fire.
// With `dead` != null only dead latents compete (where(dead, pre, -inf), topkautoencoder.py:118-121).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void topk_select_kernel(const bf16_t* __restrict__ pre, bf16_t* __restrict__ dense,
                                                           int* __restrict__ top_idx, float* __restrict__ did_fire,
                                                           const unsigned char* __restrict__ dead, const int* __restrict__ k_ptr,
                                                           int k_fixed, int kcap, int n, int n_p, int64_t M,
                                                           unsigned short* __restrict__ vals, int write_dense) {
  if (k_ptr && *k_ptr <= 0) return;     // AuxK pass without dead latents: nothing downstream reads its outputs
  __shared__ int hist[256];
  __shared__ int sel_hi, need, sel_lo, ntie;
  const int t = threadIdx.x;
  const int64_t row = blockIdx.x;
  const int k = k_ptr ? *k_ptr : k_fixed;
  const unsigned short* p = reinterpret_cast<const unsigned short*>(pre + row * n_p);
  unsigned short* o = reinterpret_cast<unsigned short*>(dense + row * n_p);
  int* ti = top_idx + row * kcap;
  unsigned short* tv = vals + row * kcap;     // the selected activations, compact: tv[p] belongs to ti[p]
  if (k <= 0 || row >= M) {             // padding rows select nothing (and must not mark any latent as fired)
    if (write_dense)
      for (int i = t; i < n_p; i += 256) o[i] = 0;
    for (int j = t; j < kcap; j += 256) ti[j] = -1;
    return;
  }
  auto key = [&](int i) -> int {       // 16-bit key; -1 = not a candidate
    if (i >= n) return -1;
    if (dead && !dead[i]) return -1;
    return (int)p[i];
  };
  // ---- level 1: histogram of the high byte
  hist[t] = 0;
  __syncthreads();
  for (int i = t; i < n_p; i += 256) {
    const int kk = key(i);
    if (kk >= 0) atomicAdd(&hist[kk >> 8], 1);
  }
  __syncthreads();
  if (t == 0) {
    int acc = 0, b = 255;
    for (; b >= 0; --b) {
      if (acc + hist[b] >= k) break;
      acc += hist[b];
    }
    if (b < 0) b = 0;
    sel_hi = b;
    need = k - acc;      // how many to take from bin b
  }
  __syncthreads();
  const int hi = sel_hi;
  __syncthreads();
  hist[t] = 0;
  __syncthreads();
  for (int i = t; i < n_p; i += 256) {
    const int kk = key(i);
    if (kk >= 0 && (kk >> 8) == hi) atomicAdd(&hist[kk & 255], 1);
  }
  __syncthreads();
  if (t == 0) {
    int acc = 0, b = 255;
    for (; b >= 0; --b) {
      if (acc + hist[b] >= need) break;
      acc += hist[b];
    }
    if (b < 0) b = 0;
    sel_lo = b;
    ntie = need - acc;   // how many elements equal to thr to take
  }
  __syncthreads();
  const int thr = (hi << 8) | sel_lo;
  const int take_ties = ntie;
  // ---- emit: elements > thr unconditionally; ties in increasing column order (wave/block ordered scan)
  __shared__ int wave_cnt[4];
  __shared__ int tie_base, out_pos;
  if (t == 0) {
    tie_base = 0;
    out_pos = 0;
  }
  __syncthreads();
  for (int i0 = 0; i0 < n_p; i0 += 256) {
    const int i = i0 + t;
    const int kk = key(i);
    const bool gt = kk > thr;
    const bool tie = kk == thr;
    // ordered rank among ties in this chunk
    const unsigned long long tb = __ballot(tie);
    const int lane = t & 63, w = t >> 6;
    const int before_in_wave = __popcll(tb & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[w] = __popcll(tb);
    __syncthreads();
    int before = tie_base + before_in_wave;
    for (int ww = 0; ww < w; ++ww) before += wave_cnt[ww];
    const bool take = gt || (tie && before < take_ties);
    unsigned short val = 0;
    if (take) {
      val = (unsigned short)kk;
      const int pos = atomicAdd(&out_pos, 1);
      if (pos < kcap) {
        ti[pos] = i;
        tv[pos] = val;
      }
      if (did_fire) did_fire[i] = 1.0f;
    }
    if (write_dense && i < n_p) o[i] = val;
    __syncthreads();
    if (t == 0) tie_base += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  // pad the index list (fewer than kcap candidates, e.g. k_aux < kcap)
  __syncthreads();
  for (int j = out_pos + t; j < kcap; j += 256) ti[j] = -1;
}

// ------------------------------------------------------------------------------------------
// per-row top-k, register-resident version: the whole row (n_p <= 2048 * MAXV bf16) is loaded once into
// registers as packed 15-bit keys (post-ReLU values are non-negative, so the bf16 bit pattern orders like the
// value; non-candidates -- padding columns, or living latents in the AuxK pass -- become key 0).  The k-th largest
// key is found by a binary search on the key value; each probe counts "key >= T" on two packed keys per
// instruction: ((w | 0x80008000) - T*0x00010001) has bit 15 / 31 set exactly where the half is >= T.
// (That is the general path; nearly every row takes the fast path in front of it -- see there.)
// Elements > T are selected; ties at T are taken in increasing column order up to k (like the radix kernel above: the
// engine's tie rule is "lowest column first", what a stable descending sort takes); exact zeros only enter the selection
// when fewer than k positive candidates exist (then in the same order).  Writes the masked dense row, the index list
// (in (thread, register) order -- the list is a set) and did_fire.  Deterministic.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int block_excl_scan_256(int v, int* wave_tot /* >= 4 ints LDS */, int* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  __syncthreads();
  if (lane == 63) wave_tot[w] = inc;
  __syncthreads();
  int base = 0;
  for (int ww = 0; ww < w; ++ww) base += wave_tot[ww];
  *total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
  return base + inc - v;
}

// COMPACT (the AuxK selection on the compacted dead set, topk_aux.h): instead of the masked dense row over all n_p columns
// the kernel writes the masked row over the dead columns only -- dense[row * n_p + r] for r < ND_p, r = rank of the column
// among the dead ones (vec_rank / vec_bits: dead_compact_kernel): the compact row is zero-filled first (the block scans'
// barriers drain those stores before anything is emitted), then every selected POSITIVE value is stored at its rank.
// Selected zeros (ties at 0 when a row has fewer positive dead latents than k_aux) carry neither activation nor
// gradient, so they are not ranked at all here; `write_dense` is implied.
// Candidates above `my` in the four per-wave candidate segments (c0..c3 entries, each segment zero-padded to a multiple of
// 8 by its wave: packed candidates are > 0).  Every lane reads the same addresses (LDS broadcast), 8 candidates per pair
// of 16-byte reads: one candidate per dependent 4-byte read made this loop the longest phase of the select kernels
// (~64 cycles of LDS latency per candidate, 150-1000 candidates per row).
template <int SEGSZ>
__device__ __forceinline__ int topk_rank_among(const unsigned int* cand_pk, int c0, int c1, int c2, int c3, unsigned int my) {
  int rank = 0;
  const int cnt[4] = {c0, c1, c2, c3};
#pragma unroll
  for (int sg = 0; sg < 4; ++sg) {
    const u32x4* p = reinterpret_cast<const u32x4*>(cand_pk + sg * SEGSZ);
    const int n8 = (cnt[sg] + 7) >> 3;
#pragma unroll 4
    for (int j = 0; j < n8; ++j) {
      const u32x4 a = p[2 * j], b = p[2 * j + 1];
      rank += (a[0] > my) + (a[1] > my) + (a[2] > my) + (a[3] > my) + (b[0] > my) + (b[1] > my) + (b[2] > my) + (b[3] > my);
    }
  }
  return rank;
}
// the pad: lanes 0..7 of the wave zero the 8 entries behind its last candidate (no-op past the segment's end)
template <int SEGSZ>
__device__ __forceinline__ void topk_pad_segment(unsigned int* seg, int wcount, int lane) {
  if (lane < 8 && wcount + lane < SEGSZ) seg[wcount + lane] = 0u;
}

// block-wide number of 16-bit keys >= T (T in 1..0x8000) among the threads' packed key vectors
template <int MAXV>
__device__ __forceinline__ int topk_count_ge(const u32x4 (&keys)[MAXV], unsigned T, int t, int* red) {
  const unsigned tp = T * 0x00010001u;
  unsigned c2 = 0;
  static_for<0, MAXV>([&](auto v_tag_) {
    constexpr int v = decltype(v_tag_)::value;
#pragma unroll
    for (int q = 0; q < 4; ++q) c2 += (((keys[v][q] | 0x80008000u) - tp) >> 15) & 0x00010001u;
  });
  int c = (int)((c2 & 0xFFFFu) + (c2 >> 16));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  __syncthreads();
  if ((t & 63) == 0) red[t >> 6] = c;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// byte offset of the transposed copy of vec_bits behind the table (dead_compact_kernel; n_p <= 32768)
__host__ __device__ __forceinline__ int vec_bits_t_offset(int n_p) { return ((n_p / 8 + 15) / 16) * 16; }
constexpr int VEC_BITS_T_BYTES = 256 * 16;
#ifndef SEL_OCC
#define SEL_OCC 3        // waves per SIMD the compact AuxK select is compiled for
#endif
template <int MAXV, bool COMPACT = false>
__global__ __launch_bounds__(256, (COMPACT && MAXV <= 12) ? SEL_OCC : 1) void topk_select_reg_kernel(const bf16_t* pre, bf16_t* dense,      // (NOT restrict: the copy + select form of the compact AuxK selection runs in place, pre == dense)
                                                               int* __restrict__ top_idx, float* __restrict__ did_fire,
                                                               const unsigned char* __restrict__ dead,
                                                               const int* __restrict__ k_ptr, int k_fixed, int kcap, int n,
                                                               int n_p, int64_t M, unsigned short* __restrict__ vals,
                                                               int write_dense, const unsigned char* __restrict__ only_flagged = nullptr,
                                                               const int* __restrict__ vec_rank = nullptr,
                                                               const unsigned char* __restrict__ vec_bits = nullptr,
                                                               const int* __restrict__ tkd = nullptr,
                                                               const int* __restrict__ dead_cols = nullptr,
                                                               int compact_mode = 0) {
  // compact_mode (round 5; dictionaries above 24 576 latents, where the COMPACT instantiation would need 44 key vectors per thread
  // -- 176 registers + candidate words, the last kernel of the code object that spilled, and 7.8 ms per step at n = 40 960):
  //   1  (COMPACT instantiation) only COPY the row's values at the dead columns into the compact row and return;
  //   2  (general instantiation) select IN PLACE on such compact rows: the row is tkd[TKD_NDP] entries long (pitch n_p), its first
  //      tkd[TKD_ND] entries are candidates; write_dense leaves the masked compact row the AuxK GEMMs read.  An instantiation whose
  //      MAXV does not fit the row -- or is larger than it needs -- returns at once: the host launches <12> and <44> back to back
  //      and the device-side count decides (no host synchronisation).
  // The selection is the one the COMPACT path makes: top k_aux of the dead latents by value, ties by lowest column (the compact
  // index keeps the column order); zero-valued fill-ins are written as the zeros they are.
  __shared__ int red[4];
  __shared__ int sc[4];
  const int t = threadIdx.x;
  const int64_t row = blockIdx.x;
#ifdef SEL_STAMP
  // diagnostic build (tools/build_variant.sh selstamp -DSEL_STAMP; bench.py --dbg 67): s_memtime at the phase boundaries of the
  // compact select, thread 0 of every row, into the last 64 bytes of the row's (otherwise unused) tail in `dense`
  unsigned long long stamp_[8];
  int nst_ = 0;
#define SEL_MARK() do { if (COMPACT && t == 0 && nst_ < 8) stamp_[nst_++] = __builtin_amdgcn_s_memtime(); } while (0)
#define SEL_FLUSH() do { if (COMPACT && t == 0) { unsigned long long* o_ = reinterpret_cast<unsigned long long*>(dense + (row + 1) * n_p) - 8; \
    for (int i_ = 0; i_ < 8; ++i_) o_[i_] = i_ < nst_ ? stamp_[i_] - stamp_[0] : 0ull; } } while (0)
#else
#define SEL_MARK() do {} while (0)
#define SEL_FLUSH() do {} while (0)
#endif
  SEL_MARK();
  if (only_flagged && !only_flagged[row]) return;     // the tile-driven kernel already selected this row
  const int k_req = k_ptr ? *k_ptr : k_fixed;
  if (k_ptr && k_req <= 0) return;      // AuxK pass without dead latents: nothing downstream reads its outputs (block-uniform)
  int nvec = n_p >> 3;                             // 16-byte vectors in the row
  if (!COMPACT && compact_mode == 2) {
    nvec = tkd[4] >> 3;
    n = tkd[0];
    if (nvec > MAXV * 256 || (MAXV > 12 && nvec <= 12 * 256)) return;      // (block-uniform) the other instantiation's row
    if (k_ptr && *k_ptr >= n) return;      // no more dead latents than k_aux: the copy IS the selection (every dead latent is taken)
  }
  const u32x4* src = reinterpret_cast<const u32x4*>(pre + row * n_p);
  u32x4* dst = reinterpret_cast<u32x4*>(dense + row * n_p);
  int* ti = top_idx + row * kcap;
  unsigned short* tv = vals + row * kcap;     // the selected activations, compact: tv[p] belongs to ti[p]
  const int cvec = COMPACT ? tkd[4] >> 3 : 0;      // 16-byte vectors of the compact row (ND_p / 8)
  if (COMPACT) write_dense = 0;
  auto cpos = [&](int col) { return vec_rank[col >> 3] + __popc((unsigned)vec_bits[col >> 3] & ((1u << (col & 7)) - 1u)); };
  if (row >= M) {                       // padding rows (all-zero pre) select nothing and must not mark any latent as fired
    if (write_dense)
      for (int g = t; g < nvec; g += 256) dst[g] = u32x4{0u, 0u, 0u, 0u};
    if (COMPACT)
      for (int g = t; g < cvec; g += 256) dst[g] = u32x4{0u, 0u, 0u, 0u};
    for (int j = t; j < kcap; j += 256) ti[j] = -1;
    return;
  }
  unsigned short* crow = reinterpret_cast<unsigned short*>(dense + row * n_p);
  if (COMPACT && (k_req >= tkd[0] || compact_mode == 1)) {
    // no more dead latents than k_aux = d/2 (the usual state of a healthy run): the AuxK selection takes EVERY dead latent,
    // so the compact row is just the row's values at the dead columns (zeros stay zeros) -- no selection, no barrier
    // (compact_mode 1: the copy is all this launch is for; the general instantiation selects on the compact rows afterwards)
    const unsigned short* prow = reinterpret_cast<const unsigned short*>(pre + row * n_p);
    for (int r = t; r < 8 * cvec; r += 256) {
      const int j = dead_cols[r];
      crow[r] = j >= 0 ? prow[j] : (unsigned short)0;
    }
    return;
  }
  if (COMPACT) {                        // (__syncthreads = s_waitcnt vmcnt(0) + barrier: these stores have landed before any emission)
    for (int g = t; g < cvec; g += 256) dst[g] = u32x4{0u, 0u, 0u, 0u};
  }

  // ---- load + candidate filter.  keys[v][q] packs columns 8 g + 2 q (low half) and 8 g + 2 q + 1 (high half)
  u32x4 keys[MAXV];
  unsigned cand[MAXV];                             // 8 candidate bits per vector
  // the dead bits of each vector's 8 columns, one byte (dead implies col < n).  MAXV <= 12: fetched up front, so that the row's
  // loads do not wait for them one by one; larger rows read them in the load loop -- a third per-vector array on top of keys and
  // cand (264 dwords at MAXV = 44) made hipcc leave all three on the stack (1072 bytes of scratch per lane)
  constexpr bool CBITS_AHEAD = COMPACT && MAXV <= 12;
  unsigned cbits[CBITS_AHEAD ? MAXV : 1];
  if (CBITS_AHEAD) {
    // the thread's bytes of vec_bits (vectors t, t + 256, ...) are the same for every row: dead_compact_kernel keeps them transposed
    // behind the table ([256 threads][16 bytes], zeros past the row's end) -- one 16-byte load instead of twelve predicated byte
    // loads (with their branches ~180 of the ~800 instructions this phase issued per thread; the phase is issue-bound)
    const u32x4 cbw = *reinterpret_cast<const u32x4*>(vec_bits + vec_bits_t_offset(n_p) + 16 * t);
    static_for<0, MAXV>([&](auto v_tag_) {
      constexpr int v = decltype(v_tag_)::value; cbits[v] = (cbw[v >> 2] >> (8 * (v & 3))) & 0xFFu;
    });
  }
  auto cbits_of = [&](int v, int g) -> unsigned { return CBITS_AHEAD ? cbits[CBITS_AHEAD ? v : 0] : (g < nvec ? (unsigned)vec_bits[g] : 0u); };
  // Round 4: ALL of the thread's row loads are issued before the first one is used (MAXV <= 12).  With the load and its masking in
  // one loop body hipcc waited for every vector in turn -- twelve dependent HBM latencies, 24 k of a row's 35 k cycles in the
  // compact select (stamps: `-DSEL_STAMP`, bench.py --dbg 67).
  constexpr bool LOADS_AHEAD = MAXV <= 12;
  if (LOADS_AHEAD) {
    static_for<0, MAXV>([&](auto v_tag_) {
      constexpr int v = decltype(v_tag_)::value;
      const int g = v * 256 + t;
      u32x4 w = {0u, 0u, 0u, 0u};
      // (COMPACT: a non-zero byte implies g < nvec)
      if (CBITS_AHEAD ? cbits_of(v, g) != 0u : (g < nvec && (!COMPACT || cbits_of(v, g) != 0u))) w = __builtin_nontemporal_load(src + g);
      keys[v] = w;
    });
  }
  // Compact select with the dead bits up front: a vector past the row's end has no dead bits and was not loaded, so the bound is
  // the block-uniform "any vector of this slot in the row" (a scalar branch) instead of a per-lane one (exec-mask save / restore
  // around every vector's masking).  (With NO branch at all hipcc spilled 186 registers instead of 40 and the kernel ran 60 % slower.)
  {
  static_for<0, MAXV>([&](auto v_tag_) {
    constexpr int v = decltype(v_tag_)::value;
    const int g = v * 256 + t;
    u32x4 w = {0u, 0u, 0u, 0u};
    unsigned cb = 0;
    if (CBITS_AHEAD ? v * 256 < nvec : g < nvec) {
      // read exactly once (3.2 GB at C3): keep it out of the caches' way.  COMPACT: vectors without a dead column are not
      // read at all (with few dead latents that is most of the row)
      const unsigned cbv = COMPACT ? cbits_of(v, g) : 0u;
      if (LOADS_AHEAD) w = keys[v];
      else if (!COMPACT || cbv != 0u) w = __builtin_nontemporal_load(src + g);
      if (!COMPACT && dead == nullptr && n == n_p) {   // every column is a candidate (block-uniform)
        cb = 0xFFu;
      } else {
        if (COMPACT) {
          cb = cbv;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int col = 8 * g + e;
            const bool ok = col < n && (!dead || dead[col]);
            cb |= ok ? (1u << e) : 0u;
          }
        }
        // zero the keys of non-candidates
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned m = ((cb >> (2 * q)) & 1u ? 0x0000FFFFu : 0u) | ((cb >> (2 * q + 1)) & 1u ? 0xFFFF0000u : 0u);
          w[q] &= m;
        }
      }
    }
    keys[v] = w;
    cand[v] = cb;
  });
  }
  int ncand_l = 0;
  static_for<0, MAXV>([&](auto v_tag_) {
    constexpr int v = decltype(v_tag_)::value; ncand_l += __popc(cand[v]);
  });
  SEL_MARK();                                          // [1] row loaded and masked
  int ncand;
  (void)block_excl_scan_256(ncand_l, sc, &ncand);
  SEL_MARK();                                          // [2] candidate count
  const int k = k_req < ncand ? k_req : ncand;     // where(dead, pre, -inf).topk(k_aux): k_aux <= num_dead by construction
  if (k <= 0) {
    if (write_dense) {
      static_for<0, MAXV>([&](auto v_tag_) {
        constexpr int v = decltype(v_tag_)::value;
        if (v * 256 + t < nvec) dst[v * 256 + t] = u32x4{0u, 0u, 0u, 0u};
      });
    }
    if (COMPACT)
      for (int g = t; g < cvec; g += 256) dst[g] = u32x4{0u, 0u, 0u, 0u};
    for (int j = t; j < kcap; j += 256) ti[j] = -1;
    return;
  }
  // ---- fast path (rows with plenty of positive candidates, i.e. nearly all): a PROVABLE lower bound L of the k-th
  // largest value from the threads' own maxima, then an exact ranking of the few keys >= L.
  //   Each wave takes the kw-th largest of its 64 lane maxima, kw = ceil(k / 4) (binary search on the value with ballots,
  //   no barrier): at least kw of its keys are >= that value, so at least 4 kw >= k keys of the row are >= the smallest
  //   of the four -- L <= k-th largest.  The keys >= L (typically 1-3 k of them) go to LDS as ONE sortable word
  //   (key << 17 | (0x1FFFF - column): larger = larger value, then lower column, the tie rule), every candidate counts
  //   the candidates above it -- its rank IS its output slot (sorted by value), ranks below k are the selection.
  //   This replaces ~18 block-wide counting probes over all n keys (the old path below, kept for rows it cannot take:
  //   L == 0, k > 1024, or a wave with more than TOPK_SEG_CAP candidates).  For 256 < k <= 1024 (the AuxK selection, k_aux =
  //   d / 2) the bound comes from FOUR values per lane instead of one.
  constexpr int TOPK_SEG_CAP = 256;                    // candidates per wave
  __shared__ __attribute__((aligned(16))) unsigned int cand_pk[4 * TOPK_SEG_CAP];
  __shared__ unsigned int wave_lb[4];
  __shared__ int wave_cnt[4];
  __shared__ unsigned int thr_pk;
  bool done = false;
  if (k <= 1024) {
    // packed largest (m1) and second largest (m2) key of the thread's even columns (low halves) and odd columns (high halves)
    typedef __attribute__((ext_vector_type(2))) unsigned short us2;
    us2 m1 = {0, 0}, m2 = {0, 0};
    static_for<0, MAXV>([&](auto v_tag_) {
      constexpr int v = decltype(v_tag_)::value;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const us2 b = __builtin_bit_cast(us2, keys[v][q]);
        m2 = __builtin_elementwise_max(m2, __builtin_elementwise_min(m1, b));
        m1 = __builtin_elementwise_max(m1, b);
      }
    });
    // k <= 256: one value per lane (its maximum), the kw-th largest of 64 with kw = ceil(k / 4) <= 64;
    // k  > 256: four values per lane (two largest of each half: four different elements of the row), kw-th largest of 256
    const bool four = k > 256;
    unsigned int tv4[4] = {max((unsigned int)m1[0], (unsigned int)m1[1]), 0u, 0u, 0u};
    if (four) {
      tv4[0] = m1[0];
      tv4[1] = m1[1];
      tv4[2] = m2[0];
      tv4[3] = m2[1];
    }
    const int kw = (k + 3) >> 2;
    unsigned int lo = 0, hi = 0x8000u;                 // largest lw with #(values >= lw) >= kw (lw = 0 always qualifies)
    while (lo < hi) {
      const unsigned int mid = (lo + hi + 1) >> 1;
      int cnt = __popcll(__ballot(tv4[0] >= mid));
      if (four) cnt += __popcll(__ballot(tv4[1] >= mid)) + __popcll(__ballot(tv4[2] >= mid)) + __popcll(__ballot(tv4[3] >= mid));
      if (cnt >= kw) lo = mid; else hi = mid - 1;
    }
    if ((t & 63) == 0) wave_lb[t >> 6] = lo;
    __syncthreads();
    const unsigned int L = min(min(wave_lb[0], wave_lb[1]), min(wave_lb[2], wave_lb[3]));
    SEL_MARK();                                        // [3] lower bound L
    // L == 0: one of the waves holds fewer than kw positive values.  The main selection then goes to the counting probes below;
    // the COMPACT (AuxK) select takes the candidate path all the same with EVERY positive key as a candidate -- dead latents are
    // dead because their pre-activations are mostly negative, so a row has only a few hundred positive dead values, about k_aux
    // of them: round 3 sent every such row (ALL rows of the C3 bench with 31 % dead, stamps of round 4) through ~17 block-wide
    // probes over all 96 keys per thread, 106 k of the row's 136 k cycles.
    const unsigned int Lc = (COMPACT && L == 0u) ? 1u : L;
    if (Lc > 0) {                                      // block-uniform
      // Append the keys >= L to this WAVE's segment of the candidate list: positions come from a ballot + lane prefix and a
      // wave-uniform running count -- no LDS atomics (an add-with-return per hit cost a ~100-cycle round trip each; this
      // loop was 60 % of the kernel), and the scalar branch skips the pairs in which no lane of the wave has a hit.
      const unsigned int lp = Lc * 0x00010001u;
      const int wv = t >> 6, lane = t & 63;
      unsigned int* seg = cand_pk + wv * TOPK_SEG_CAP;
      int wcount = 0;                                  // wave-uniform
      static_for<0, MAXV>([&](auto v_tag_) {
        constexpr int v = decltype(v_tag_)::value;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned int w = keys[v][q];
          const unsigned int hit = ((w | 0x80008000u) - lp) & 0x80008000u;       // bit 15 / 31: low / high key >= L
          const unsigned long long any = __ballot(hit != 0u);
          if (any != 0ull) {                           // wave-uniform
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const bool mine = (hit >> (15 + 16 * h)) & 1u;
              const unsigned long long bm = __ballot(mine);
              if (mine) {
                const int pos = wcount + (int)__popcll(bm & ((1ull << lane) - 1ull));
                const unsigned int key = (w >> (16 * h)) & 0xFFFFu;
                const int col = 8 * (v * 256 + t) + 2 * q + h;
                if (pos < TOPK_SEG_CAP) seg[pos] = (key << 17) | (0x1FFFFu - (unsigned int)col);
              }
              wcount += (int)__popcll(bm);
            }
          }
        }
      });
      if (lane == 0) wave_cnt[wv] = wcount;
      topk_pad_segment<TOPK_SEG_CAP>(seg, wcount, lane);
      __syncthreads();
      SEL_MARK();                                      // [4] candidates appended
      const int c0 = wave_cnt[0], c1 = wave_cnt[1], c2 = wave_cnt[2], c3 = wave_cnt[3];
      const int C = c0 + c1 + c2 + c3;
      const bool fits = c0 <= TOPK_SEG_CAP && c1 <= TOPK_SEG_CAP && c2 <= TOPK_SEG_CAP && c3 <= TOPK_SEG_CAP;
      if (fits && COMPACT) {
        // The compact row needs MEMBERSHIP only (a selected value goes to the rank of its column among the dead ones, not to
        // the rank of its value): the k-th largest packed candidate by binary search on the 32-bit word -- ballots per wave,
        // one LDS exchange and ONE barrier per probe (double-buffered counters) -- instead of C comparisons per candidate
        // (that ranking was 60 % of this kernel's instructions at k_aux = 384, C ~ 450).  The packed words are distinct
        // (they carry the column), so the probe that counts exactly k ends the search.
        __shared__ int bs_cnt[2][4];
        unsigned int mine[4] = {0u, 0u, 0u, 0u};           // C <= 4 * TOPK_SEG_CAP = 1024: at most 4 candidates per thread
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = t + 256 * u;
          if (i < C) {
            const int sg = i < c0 ? 0 : (i < c0 + c1 ? 1 : (i < c0 + c1 + c2 ? 2 : 3));
            const int li = i - (sg == 0 ? 0 : (sg == 1 ? c0 : (sg == 2 ? c0 + c1 : c0 + c1 + c2)));
            mine[u] = cand_pk[sg * TOPK_SEG_CAP + li];
          }
        }
        if (C <= k) {      // (only with Lc = 1: no more positive dead latents than k_aux in this row) every candidate is selected; the
                           // zeros that fill the selection up carry neither activation nor gradient
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (mine[u] != 0u) crow[cpos((int)(0x1FFFFu - (mine[u] & 0x1FFFFu)))] = (unsigned short)(mine[u] >> 17);
          SEL_MARK();
          SEL_FLUSH();
          return;
        }
#ifndef SEL_BINSEARCH
        // Round 4: the k-th largest candidate by a two-level RADIX select instead of a binary search over the 32-bit word (up to 32
        // probes, a block barrier each).  Level 1: histogram over the key's upper 8 bits (sign-free bf16: the exponent) -- an LDS add
        // per candidate, one block scan over the reversed bins finds the bin B1 that holds the k-th largest and how many of its
        // members are needed; level 2: the same over the lower 7 bits (the mantissa) of B1's members; what is left ties on the
        // VALUE, and the rule "lowest column first" ranks it: the members go to a short list and count the members above them.
        // An oversized tie (> 256) falls back to the binary search below (block-uniform).
        __shared__ int hist[256];
        __shared__ int sel_bin, sel_need, tie_n;
        __shared__ unsigned int tie_list[256];
        int B1 = 0, need1 = 0, B2 = -1, need2 = 0, inB2 = 0;
        bool all1 = false, all2 = false;
#pragma unroll 1
        for (int level = 0; level < 2; ++level) {
          hist[t] = 0;
          if (t == 0) tie_n = 0;
          __syncthreads();
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (mine[u] != 0u) {
              const int b1 = (int)(mine[u] >> 24), b2 = (int)((mine[u] >> 17) & 127u);
              if (level == 0) atomicAdd(&hist[b1], 1);
              else if (b1 == B1) atomicAdd(&hist[b2], 1);
            }
          __syncthreads();
          const int hr = hist[255 - t];                       // reversed: the prefix over t is the suffix over the bins
          const int want = level == 0 ? k : need1;
          int tot;
          const int excl = block_excl_scan_256(hr, sc, &tot);
          if (excl < want && want <= excl + hr) {             // exactly one thread (the bins hold >= want candidates, hr > 0 there)
            sel_bin = 255 - t;
            sel_need = want - excl;
          }
          __syncthreads();
          if (level == 0) {
            B1 = sel_bin; need1 = sel_need;
            all1 = need1 == hist[B1];
            if (all1) break;                                  // (block-uniform) the whole bin is needed: no second level
          } else {
            B2 = sel_bin; need2 = sel_need; inB2 = hist[B2];
            all2 = need2 == inB2;
          }
          __syncthreads();                                    // hist is cleared again at the top of level 1
        }
        if (all1 || all2 || inB2 <= 256) {                    // block-uniform
          const bool ties = !all1 && !all2;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int b1 = (int)(mine[u] >> 24), b2 = (int)((mine[u] >> 17) & 127u);
            if (ties && mine[u] != 0u && b1 == B1 && b2 == B2) tie_list[atomicAdd(&tie_n, 1)] = mine[u];
          }
          if (ties) __syncthreads();
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (mine[u] == 0u) continue;
            const int b1 = (int)(mine[u] >> 24), b2 = (int)((mine[u] >> 17) & 127u);
            bool sel = b1 > B1 || (b1 == B1 && (all1 || b2 > B2 || (b2 == B2 && all2)));
            if (ties && b1 == B1 && b2 == B2) {
              int above = 0;
              for (int i = 0; i < inB2; ++i) above += tie_list[i] > mine[u] ? 1 : 0;
              sel = above < need2;
            }
            if (sel) crow[cpos((int)(0x1FFFFu - (mine[u] & 0x1FFFFu)))] = (unsigned short)(mine[u] >> 17);
          }
          SEL_MARK();                                  // [5] selected and stored
          SEL_FLUSH();
          return;
        }
#endif
        unsigned int lo = 1u, hi = 0xFFFFFFFFu;             // largest T with #(candidates >= T) >= k; C >= k, candidates > 0
        int probe = 0;
        while (lo < hi) {
          const unsigned int mid = lo + ((hi - lo) >> 1) + 1u;
          int cw = 0;
#pragma unroll
          for (int u = 0; u < 4; ++u) cw += (int)__popcll(__ballot(mine[u] >= mid));
          if (lane == 0) bs_cnt[probe & 1][wv] = cw;
          __syncthreads();
          const int cnt = bs_cnt[probe & 1][0] + bs_cnt[probe & 1][1] + bs_cnt[probe & 1][2] + bs_cnt[probe & 1][3];
          ++probe;
          if (cnt >= k) {
            lo = mid;
            if (cnt == k) break;
          } else {
            hi = mid - 1u;
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (mine[u] >= lo && mine[u] != 0u) crow[cpos((int)(0x1FFFFu - (mine[u] & 0x1FFFFu)))] = (unsigned short)(mine[u] >> 17);
        done = true;
      } else if (fits) {                               // block-uniform; C >= k by construction
        for (int i = t; i < C; i += 256) {
          // candidate i of the concatenated segments
          const int sg = i < c0 ? 0 : (i < c0 + c1 ? 1 : (i < c0 + c1 + c2 ? 2 : 3));
          const int li = i - (sg == 0 ? 0 : (sg == 1 ? c0 : (sg == 2 ? c0 + c1 : c0 + c1 + c2)));
          const unsigned int my = cand_pk[sg * TOPK_SEG_CAP + li];
          const int rank = topk_rank_among<TOPK_SEG_CAP>(cand_pk, c0, c1, c2, c3, my);
          if (rank < k) {
            const int col = (int)(0x1FFFFu - (my & 0x1FFFFu));
            if (rank < kcap) {
              ti[rank] = col;
              tv[rank] = (unsigned short)(my >> 17);
            }
            if (did_fire) did_fire[col] = 1.0f;
            if (rank == k - 1) thr_pk = my;
            if (COMPACT && (my >> 17) != 0u) crow[cpos(col)] = (unsigned short)(my >> 17);
          }
        }
        for (int j = k + t; j < kcap; j += 256) ti[j] = -1;
        if (write_dense) {                             // the masked dense row for those who read it (validation, fallbacks)
          __syncthreads();
          const unsigned int thr = thr_pk;
          static_for<0, MAXV>([&](auto v_tag_) {
            constexpr int v = decltype(v_tag_)::value;
            const int g = v * 256 + t;
            if (g < nvec) {
              u32x4 o;
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                unsigned int w = keys[v][q], keep = 0u;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                  const unsigned int key = (w >> (16 * h)) & 0xFFFFu;
                  const unsigned int pk = (key << 17) | (0x1FFFFu - (unsigned int)(8 * g + 2 * q + h));
                  keep |= (key >= L && pk >= thr) ? (0xFFFFu << (16 * h)) : 0u;
                }
                o[q] = w & keep;
              }
              dst[g] = o;
            }
          });
        }
        done = true;
      }
    }
  }
  if (done) return;
  __syncthreads();
  SEL_MARK();                                          // (slow path) [k] fast path not taken

  // ---- largest T in [1, 0x8000] with count(key >= T) >= k; T = 0 if fewer than k positive keys
  const int npos = topk_count_ge<MAXV>(keys, 1u, t, red);
  if (COMPACT && npos <= k) {
    // no more positive dead latents than k_aux in this row: all of them are selected (and the zeros that fill the selection up
    // carry nothing) -- each thread stores its own positive candidates at their compact positions, nothing to rank or scan
    static_for<0, MAXV>([&](auto v_tag_) {
      constexpr int v = decltype(v_tag_)::value;
      const int g = v * 256 + t;
      if (g < nvec && cand[v] != 0u) {
        const int base = vec_rank[g];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned short key = (unsigned short)((keys[v][e >> 1] >> (16 * (e & 1))) & 0xFFFFu);
          if (((cand[v] >> e) & 1u) && key != 0) crow[base + __popc(cand[v] & ((1u << e) - 1u))] = key;
        }
      }
    });
    return;
  }
  unsigned T = 0;
  if (npos >= k) {
    unsigned lo = 1, hi = 0x8000u;                 // invariant: count(>= lo) >= k, count(>= hi + 1) < k
    while (lo < hi) {
      const unsigned mid = (lo + hi + 1) >> 1;
      if (topk_count_ge<MAXV>(keys, mid, t, red) >= k) lo = mid;
      else hi = mid - 1;
    }
    T = lo;
  }
  const int n_gt = T >= 0x8000u ? 0 : (T == 0 ? npos : topk_count_ge<MAXV>(keys, T + 1, t, red));
  const int need_ties = k - n_gt;                  // elements equal to T to take (for T == 0: candidate zeros)
  // ---- ties at T in INCREASING COLUMN order (the rule of all three select kernels: what a stable descending sort
  // takes).  Column = 8 (256 v + t) + e, i.e. vector-row v first, then thread, then element.  Most rows take every
  // element equal to T (no boundary tie): only a row with more ties than it needs ranks them, one block scan per
  // vector-row (block-uniform branch).
  int tie_l = 0;
  static_for<0, MAXV>([&](auto v_tag_) {
    constexpr int v = decltype(v_tag_)::value;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned key = (keys[v][e >> 1] >> (16 * (e & 1))) & 0xFFFFu;
      tie_l += (key == T && ((cand[v] >> e) & 1u)) ? 1 : 0;
    }
  });
  int tie_tot;
  (void)block_excl_scan_256(tie_l, sc, &tie_tot);
  const bool skip_zero = COMPACT && T == 0;          // zero-valued ties: nothing downstream of the compact row sees them
  const bool rank_ties = !skip_zero && tie_tot > need_ties;
  // ---- emit
  int sel_l = 0, tie_base = 0;
  // (the selection mask of a vector rides in bits 8-15 of its candidate word: a third per-vector array -- 44 registers at MAXV = 44 --
  // was what pushed that instantiation over the register file)
  static_for<0, MAXV>([&](auto v_tag_) {
    constexpr int v = decltype(v_tag_)::value;
    unsigned gm = 0, tm = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned key = (keys[v][e >> 1] >> (16 * (e & 1))) & 0xFFFFu;
      const bool c = (cand[v] >> e) & 1u;
      gm |= (c && key > T) ? (1u << e) : 0u;
      tm |= (c && key == T) ? (1u << e) : 0u;
    }
    if (skip_zero) tm = 0;
    if (rank_ties) {
      int row_tot;
      const int before = block_excl_scan_256(__popc(tm), sc, &row_tot);
      int budget = need_ties - tie_base - before;      // ties of this vector this thread may still take
      tie_base += row_tot;
      unsigned keep = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if ((tm >> e) & 1u) {
          keep |= budget > 0 ? (1u << e) : 0u;
          budget -= 1;
        }
      tm = keep;
    }
    cand[v] = (cand[v] & 0xFFu) | ((gm | tm) << 8);
    sel_l += __popc(gm | tm);
  });
  int sel_tot;
  int pos = block_excl_scan_256(sel_l, sc, &sel_tot);
  static_for<0, MAXV>([&](auto v_tag_) {
    constexpr int v = decltype(v_tag_)::value;
    const int g = v * 256 + t;
    if (g < nvec) {
      if (write_dense) {          // the masked dense row: only the inference / validation calls and the dense fallbacks read it
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned m = (((cand[v] >> 8) >> (2 * q)) & 1u ? 0x0000FFFFu : 0u) | (((cand[v] >> 8) >> (2 * q + 1)) & 1u ? 0xFFFF0000u : 0u);
          o[q] = keys[v][q] & m;
        }
        dst[g] = o;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (((cand[v] >> 8) >> e) & 1u) {
          const int col = 8 * g + e;
          const unsigned short key = (unsigned short)((keys[v][e >> 1] >> (16 * (e & 1))) & 0xFFFFu);
          if (pos < kcap) {
            ti[pos] = col;
            tv[pos] = key;
          }
          ++pos;
          if (did_fire) did_fire[col] = 1.0f;
          if (COMPACT && key != 0) crow[vec_rank[g] + __popc(cand[v] & ((1u << e) - 1u))] = key;
        }
    }
  });
  for (int j = sel_tot + t; j < kcap; j += 256) ti[j] = -1;
#ifdef SEL_STAMP
  SEL_MARK();                                          // slow path done
  if (COMPACT && t == 0) {                              // slot 7 = 1000 + marks: tells the slow path from the histogram path (slot 7 = 0)
    unsigned long long* o_ = reinterpret_cast<unsigned long long*>(dense + (row + 1) * n_p) - 8;
    for (int i_ = 0; i_ < 7; ++i_) o_[i_] = i_ < nst_ ? stamp_[i_] - stamp_[0] : 0ull;
    o_[7] = 1000ull + (unsigned long long)nst_;
  }
#endif
}

// ------------------------------------------------------------------------------------------
// Tile-driven select (training, main selection: no dead filter, no dense row): the encoder GEMM's epilogue left the
// maximum of every 64-column tile of the row (EpiTopkEnc::tmax).  The k-th largest tile maximum L is a provable lower
// bound of the k-th largest value (k tiles hold a value >= L each), so only the tiles whose maximum reaches L can hold a
// selected value: k tiles unless maxima tie -- a sixth of the row at C3 (64 of 384 tiles) instead of all of it.  The keys >= L of those tiles are ranked
// exactly like in topk_select_reg_kernel's fast path (one sortable word per candidate: value, then lowest column).
// Returns 0 work for rows it cannot take (flag[row] = 1: more than the candidate capacity, or L == 0) -- the caller runs
// topk_select_reg_kernel afterwards with `only_flagged`, which skips the rows done here.
// ------------------------------------------------------------------------------------------
constexpr int TSEL_TILE = 64;            // columns per tile maximum (round 3: 128 -- twice the candidate bytes and compaction work per row)
constexpr int TSEL_MAX_TILES = 2048;     // n_p <= 131072
constexpr int TSEL_VEC = 8;              // 16-byte vectors per thread at most (2048 vectors = 256 tiles; rows with more are left to the general kernel)
#ifdef TSEL_STAMP
// diagnostic build (tools/build_variant.sh tselstamp -DTSEL_STAMP; bench.py --dbg 68): s_memtime at the phase boundaries of the first
// 4096 rows, wave 0: [row][8] = cycles since the row's start
__device__ unsigned long long tsel_stamp_buf[4096 * 8];
#define TSEL_MARK() do { if (t == 64 * lead && nst_ < 8) stamp_[nst_++] = __builtin_amdgcn_s_memtime(); } while (0)
#define TSEL_FLUSH() do { if (t == 64 * lead && row < 4096) { for (int i_ = 0; i_ < 8; ++i_) tsel_stamp_buf[row * 8 + i_] = i_ < nst_ ? stamp_[i_] - stamp_[0] : 0ull; } } while (0)
#else
#define TSEL_MARK() do {} while (0)
#define TSEL_FLUSH() do {} while (0)
#endif
template <int NPER>                     // tile maxima per lane: n_p <= 4096 NPER (8: C3's 384 tiles; 32: any supported n_p)
__global__ __launch_bounds__(256) void topk_select_tiles_kernel(const bf16_t* __restrict__ pre, const unsigned short* __restrict__ tmax,
                                                                int* __restrict__ top_idx, unsigned short* __restrict__ vals,
                                                                float* __restrict__ did_fire, int k, int kcap, int n_p, int64_t M,
                                                                unsigned char* __restrict__ flag) {
  constexpr int SEG = 256;
  __shared__ unsigned short tlist[TSEL_MAX_TILES];
  __shared__ __attribute__((aligned(16))) unsigned int cand_pk[4 * SEG];
  __shared__ int wave_cnt[4];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int64_t row = blockIdx.x;
  const int ntiles = n_p / TSEL_TILE;
  // The row's leader wave runs the descent below and ranks the first 64 candidates at the end.  (Rotating the leader with a hash
  // of the row -- in case wave 0 of every workgroup shared a SIMD -- was 4 % SLOWER: 358 against 344 us, -DTSEL_ROTATE_LEADER.)
#ifdef TSEL_ROTATE_LEADER
  const int lead = (int)(((blockIdx.x >> 3) * 0x9E3779B1u) >> 30);
#else
  constexpr int lead = 0;
#endif
  int* ti = top_idx + row * kcap;
  unsigned short* tv = vals + row * kcap;
  if (t == 0) flag[row] = 0;
  if (row >= M) {
    for (int j = t; j < kcap; j += 256) ti[j] = -1;
    return;
  }
#ifdef TSEL_STAMP
  unsigned long long stamp_[8];
  int nst_ = 0;
#endif
  TSEL_MARK();                                      // [0] start
  // Round 4: WAVE 0 finds L and the candidate tiles in registers -- lane l holds the maxima of tiles l, l + 64, ...; L = the k-th
  // largest maximum by a bitwise descent over the 15 value bits (bf16 patterns of non-negative numbers): "at least k maxima >= T"
  // is one ballot + population count per 64 tiles, all on the scalar unit; the candidate list is the ballot's prefix order.  The
  // other three waves wait at the barrier: a waiting wave takes no issue slots, and the kernel is issue-bound (DESIGN section 4
  // "TopK": every wave running the descent for itself, without the barrier, was 6 % slower).
  __shared__ unsigned int L_s;
  __shared__ int nq_s;
  if (wv == lead) {
    static_assert(NPER * 64 <= TSEL_MAX_TILES, "tlist");
    unsigned int tmv[NPER];
#pragma unroll
    for (int q = 0; q < NPER; ++q) {
      tmv[q] = 0u;
      if (64 * q < ntiles) {                            // (uniform)
        const int i = 64 * q + lane;
        if (i < ntiles) tmv[q] = tmax[row * ntiles + i];
      }
    }
#ifdef TSEL_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    TSEL_MARK();                                      // [1] tile maxima in registers
    unsigned int Lw = 0u;
#pragma unroll 1
    for (int bit = 14; bit >= 0; --bit) {
      const unsigned int T = Lw | (1u << bit);
      int cnt = 0;
#pragma unroll
      for (int q = 0; q < NPER; ++q)
        if (64 * q < ntiles) cnt += (int)__popcll(__ballot(tmv[q] >= T));
      if (cnt >= k) Lw = T;
    }
    int nqw = 0;
    if (Lw != 0u) {
#pragma unroll
      for (int q = 0; q < NPER; ++q)
        if (64 * q < ntiles) {
          const bool mine = tmv[q] >= Lw;
          const unsigned long long bm = __ballot(mine);
          if (mine) tlist[nqw + (int)__popcll(bm & ((1ull << lane) - 1ull))] = (unsigned short)(64 * q + lane);
          nqw += (int)__popcll(bm);
        }
    }
    if (lane == 0) {
      L_s = Lw;
      nq_s = nqw;
    }
  }
  __syncthreads();
  const unsigned int L = L_s;
  if (L == 0u) {                                    // fewer than k tiles with a positive value: the general kernel takes the row
    if (t == 0) flag[row] = 1;
    return;
  }
  const int nq = nq_s;
  constexpr int VPT = TSEL_TILE / 8;                 // 16-byte vectors per tile
  const int nv = nq * VPT;                           // 16-byte vectors to read
  TSEL_MARK();                                      // [2] L and the candidate tile list
  const unsigned int lp = L * 0x00010001u;
  unsigned int* seg = cand_pk + wv * SEG;
  // Round 4: every load of the thread is issued before the first is used (with the load inside the compaction loop each of the
  // up to twelve vectors cost a full dependent memory latency); vector slots past the row's candidate tiles are skipped
  // block-uniformly (two of the eight slots are in use at C3).
  u32x4 wv_[TSEL_VEC];
#pragma unroll
  for (int v = 0; v < TSEL_VEC; ++v) {
    wv_[v] = u32x4{0u, 0u, 0u, 0u};
    if (v * 256 < nv) {
      const int vi = v * 256 + t;
      if (vi < nv)
        wv_[v] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(pre + row * n_p + (int)tlist[vi / VPT] * TSEL_TILE + (vi % VPT) * 8));
    }
  }
#ifdef TSEL_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  TSEL_MARK();                                      // [3] candidate tiles in registers
#ifndef TSEL_SCAN_COMPACT
  // compaction: one ballot per dword and one per 16-bit half with a hit (about one value in a hundred of the candidate tiles
  // reaches L, but a wave's 512 values per vector nearly always hold some, so nearly every branch below is taken)
  int wcount = 0;
#pragma unroll
  for (int v = 0; v < TSEL_VEC; ++v) {
    if (v * 256 >= nv) continue;
    const u32x4 w = wv_[v];
    const int vi = v * 256 + t;
    const int col0 = vi < nv ? (int)tlist[vi / VPT] * TSEL_TILE + (vi % VPT) * 8 : 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned int hit = ((w[q] | 0x80008000u) - lp) & 0x80008000u;
      const unsigned long long any = __ballot(hit != 0u);
      if (any != 0ull) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bool mine = (hit >> (15 + 16 * h)) & 1u;
          const unsigned long long bm = __ballot(mine);
          if (mine) {
            const int pos = wcount + (int)__popcll(bm & ((1ull << lane) - 1ull));
            const unsigned int key = (w[q] >> (16 * h)) & 0xFFFFu;
            if (pos < SEG) seg[pos] = (key << 17) | (0x1FFFFu - (unsigned int)(col0 + 2 * q + h));
          }
          wcount += (int)__popcll(bm);
        }
      }
    }
  }
#else
  // Tried in round 4 (-DTSEL_SCAN_COMPACT), NOT faster (kernel 391 against 379 us, profiles/r04_tile_select_builds.txt) although
  // its stamps show 27 % fewer cycles in this phase: compaction in two passes over the registers: (1) every lane counts its own values >= L (packed compare: bit 15 / 31 of
  // (w | 0x8000'8000) - L'L per 16-bit half), ONE wave-wide exclusive scan of the counts (DPP) gives the lane its first slot in
  // the wave's segment; (2) the few lanes with hits (a lane's eight values of a vector hold one with probability ~8 %) walk the
  // set bits of their hit mask and store the keys.
  // hit mask of a vector: bit q = the low half of dword q (column 2 q), bit 16 + q = its high half (column 2 q + 1)
  auto hit_mask = [&](const u32x4& w) -> unsigned int {
    const unsigned int h0 = ((w[0] | 0x80008000u) - lp) & 0x80008000u, h1 = ((w[1] | 0x80008000u) - lp) & 0x80008000u;
    const unsigned int h2 = ((w[2] | 0x80008000u) - lp) & 0x80008000u, h3 = ((w[3] | 0x80008000u) - lp) & 0x80008000u;
    return (h0 >> 15) | (h1 >> 14) | (h2 >> 13) | (h3 >> 12);
  };
  int cnt = 0;
#pragma unroll
  for (int v = 0; v < TSEL_VEC; ++v)
    if (v * 256 < nv) cnt += __popc(hit_mask(wv_[v]));
  int incl = cnt;
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false);      // row_shr:1 (lanes without a source add 0)
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false);      // row_shr:2
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, false);      // row_shr:4
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, false);      // row_shr:8: inclusive within the row of 16
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, false);      // row_bcast15: rows 1, 3 += lane 15 of the row before
  incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, false);      // row_bcast31: rows 2, 3 += lane 31
  const int wcount = __builtin_amdgcn_readlane(incl, 63);
  int pos = incl - cnt;
#pragma unroll
  for (int v = 0; v < TSEL_VEC; ++v) {
    if (v * 256 >= nv) continue;
    const u32x4 w = wv_[v];
    unsigned int u = hit_mask(w);
    if (u != 0u) {
      const int vi = v * 256 + t;
      const int col0 = (int)tlist[vi / VPT] * TSEL_TILE + (vi % VPT) * 8;
      while (u != 0u) {
        const int b = __ffs(u) - 1;
        u &= u - 1u;
        const int q = b & 3, h = b >> 4;
        const unsigned int dw = q == 0 ? w[0] : (q == 1 ? w[1] : (q == 2 ? w[2] : w[3]));
        const unsigned int key = (dw >> (16 * h)) & 0xFFFFu;
        if (pos < SEG) seg[pos] = (key << 17) | (0x1FFFFu - (unsigned int)(col0 + 2 * q + h));
        ++pos;
      }
    }
  }
#endif
  TSEL_MARK();                                      // [4] candidates appended
  if (lane == 0) wave_cnt[wv] = wcount;
  topk_pad_segment<SEG>(seg, wcount, lane);
  __syncthreads();
  TSEL_MARK();                                      // [5] block barrier
  const int c0 = wave_cnt[0], c1 = wave_cnt[1], c2 = wave_cnt[2], c3 = wave_cnt[3];
  const int C = c0 + c1 + c2 + c3;
  if (nv > TSEL_VEC * 256 || c0 > SEG || c1 > SEG || c2 > SEG || c3 > SEG) {     // block-uniform
    if (t == 0) flag[row] = 1;
    return;
  }
  for (int i = ((wv - lead) & 3) * 64 + lane; i < C; i += 256) {      // candidates 0..63 on the leader, 64..127 on the wave after it, ...
    const int sg = i < c0 ? 0 : (i < c0 + c1 ? 1 : (i < c0 + c1 + c2 ? 2 : 3));
    const int li = i - (sg == 0 ? 0 : (sg == 1 ? c0 : (sg == 2 ? c0 + c1 : c0 + c1 + c2)));
    const unsigned int my = cand_pk[sg * SEG + li];
    const int rank = topk_rank_among<SEG>(cand_pk, c0, c1, c2, c3, my);
    if (rank < k) {
      const int col = (int)(0x1FFFFu - (my & 0x1FFFFu));
      if (rank < kcap) {
        ti[rank] = col;
        tv[rank] = (unsigned short)(my >> 17);
      }
      if (did_fire) did_fire[col] = 1.0f;
    }
  }
  for (int j = k + t; j < kcap; j += 256) ti[j] = -1;
#ifdef TSEL_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TSEL_MARK();                                      // [6] ranked and stored
  if (t == 64 * lead && nst_ < 8) stamp_[nst_++] = stamp_[0] + (unsigned long long)(C * 1000 + nq);     // [7] candidates * 1000 + candidate tiles
  TSEL_FLUSH();
#endif
}

// masked dense row from the compact selection (only when somebody asks for it: sae_latent_buffer, sae_latent_colmax,
// sae_debug_read after a training forward, which does not write the dense rows): one block per row
__global__ __launch_bounds__(256) void topk_densify_kernel(const bf16_t* __restrict__ vals, const int* __restrict__ idx, int kcap,
                                                            bf16_t* __restrict__ dense, int n_p) {
  const int64_t row = blockIdx.x;
  u32x4* dst = reinterpret_cast<u32x4*>(dense + row * n_p);
  for (int g = threadIdx.x; g < n_p / 8; g += 256) dst[g] = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  for (int p = threadIdx.x; p < kcap; p += 256) {
    const int j = idx[row * kcap + p];
    if (j >= 0) dense[row * n_p + j] = vals[row * kcap + p];
  }
}

// ------------------------------------------------------------------------------------------
// sparse decode + losses: x_hat = bf16(sum_j act_j W_dec[idx_j]) + b_dec ; e = x_hat - x
// (topkautoencoder.py:15-18,87-91,101-102).  One workgroup per activation row, threads over d.
// pass 0 (aux == 0): writes e (fp32) and sum e^2, sum over valid elements.
// pass 1 (aux == 1): e_hat from the aux selection, accumulates sum (e_hat - e)^2 and writes dh = e_hat - e.
// ------------------------------------------------------------------------------------------
// One WAVE per activation row (4 rows per workgroup): lane l owns the d_p/64 contiguous columns
// [l*cpl, (l+1)*cpl), so every gathered W_dec row is read as one contiguous, fully coalesced line by the wave.
// NPAIR > 0: d_p == 128 * NPAIR is a compile-time constant, so a lane's 4 * NPAIR bytes of a W_dec row are fetched with
// unconditional (mergeable into dwordx2/x4) loads and two gathered rows are kept in flight; NPAIR == 0: any d_p <= 1536.
#ifndef TKD_ROWS
#define TKD_ROWS 2        // gathered W_dec rows in flight per wave and trip of topk_decode_kernel (C3, one box: 2 -> 801 us, 4 -> 821,
                          // 8 -> 903: the gather is bound by what misses the L2s, not by its round trips)
#endif
template <typename T, int NPAIR = 0>
__global__ __launch_bounds__(256) void topk_decode_kernel(const T* __restrict__ x, const bf16_t* __restrict__ vals,
                                                           const int* __restrict__ idx, int kcap, const bf16_t* __restrict__ Wd,
                                                           const float* __restrict__ b_dec, float* __restrict__ e,
                                                           float* __restrict__ dh, float* __restrict__ part, int64_t M, int d,
                                                           int d_p, int n_p, int aux, const int* __restrict__ gate) {
  if (gate && *gate <= 0) return;                // AuxK pass while no latent is dead (decided on the device: no host sync)
  constexpr int MAXP = NPAIR > 0 ? NPAIR : 12;   // column pairs per lane: d_p <= 64 * 2 * MAXP
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + w;
  const int npair = NPAIR > 0 ? NPAIR : d_p >> 7;   // (d_p / 64) / 2 column pairs per lane
  const int c0 = lane * 2 * npair;
  float acc[2 * MAXP];
#pragma unroll
  for (int i = 0; i < 2 * MAXP; ++i) acc[i] = 0.f;
  const int* ri = idx + row * kcap;
  const bf16_t* rv = vals + row * kcap;      // compact selected activations, rv[p] <-> ri[p]
  for (int j0 = 0; j0 < kcap; j0 += 64) {
    // 64 (index, activation) pairs at a time, one per lane, broadcast with readlane-style shuffles
    const int jj = j0 + lane;
    const int my_i = jj < kcap ? ri[jj] : -1;
    const float my_a = my_i >= 0 ? (float)rv[jj] : 0.f;
    const int cnt = kcap - j0 < 64 ? kcap - j0 : 64;
    if constexpr (NPAIR > 0) {
      // TKD_ROWS gathered rows per trip, all of their loads in flight together (negative = padding index: row 0 is read and
      // weighted by 0); the products are added in index order whatever the trip length
      for (int j = 0; j < cnt; j += TKD_ROWS) {
        float av[TKD_ROWS];
        const unsigned* wp[TKD_ROWS];
#pragma unroll
        for (int r = 0; r < TKD_ROWS; ++r) {
          const int jr = j + r < 64 ? j + r : 63;              // (wave-uniform)
          const int ir = __shfl(my_i, jr, 64);
          const bool ok = j + r < cnt && ir >= 0;
          av[r] = ok ? __shfl(my_a, jr, 64) : 0.f;
          wp[r] = reinterpret_cast<const unsigned*>(Wd + (int64_t)(ok ? ir : 0) * d_p + c0);
        }
        unsigned u[TKD_ROWS][NPAIR];
#pragma unroll
        for (int r = 0; r < TKD_ROWS; ++r)
#pragma unroll
          for (int p = 0; p < NPAIR; ++p) u[r][p] = wp[r][p];
#pragma unroll
        for (int r = 0; r < TKD_ROWS; ++r)
#pragma unroll
          for (int p = 0; p < NPAIR; ++p) {
            acc[2 * p] += av[r] * __uint_as_float(u[r][p] << 16);
            acc[2 * p + 1] += av[r] * __uint_as_float(u[r][p] & 0xFFFF0000u);
          }
      }
    } else {
      for (int j = 0; j < cnt; ++j) {
        const int ii = __shfl(my_i, j, 64);
        const float av = __shfl(my_a, j, 64);
        if (ii < 0) continue;               // wave-uniform
        const unsigned* wr = reinterpret_cast<const unsigned*>(Wd + (int64_t)ii * d_p + c0);
#pragma unroll
        for (int p = 0; p < MAXP; ++p)
          if (p < npair) {
            const unsigned u = wr[p];
            acc[2 * p] += av * __uint_as_float(u << 16);
            acc[2 * p + 1] += av * __uint_as_float(u & 0xFFFF0000u);
          }
      }
    }
  }
  float sq = 0.f;
  // (Round 5: the bias and target values of the lane's columns are loaded TOGETHER, from clamped addresses, before any is used.  Written as
  // `if (row < M && c < d) { ... b_dec[c] ... x[row * d + c] ... }` per column every column was two loads and two `s_waitcnt vmcnt(0)`: up to
  // 24 memory latencies in a row at the end of every wave -- next to the 32 trips of the gather loop above.)
  float bdv[2 * MAXP], tgt[2 * MAXP];
  {
    const int64_t rr = row < M ? row : M - 1;
#pragma unroll
    for (int p = 0; p < 2 * MAXP; ++p)
      if (p < 2 * npair) bdv[p] = b_dec[c0 + p < d ? c0 + p : d - 1];
    if (aux) {                 // (the switch outside the loops, raw values first, conversions after: no wait between the loads)
#pragma unroll
      for (int p = 0; p < 2 * MAXP; ++p)
        if (p < 2 * npair) tgt[p] = e[rr * d_p + (c0 + p < d ? c0 + p : d - 1)];
    } else {
      T raw[2 * MAXP];
#pragma unroll
      for (int p = 0; p < 2 * MAXP; ++p)
        if (p < 2 * npair) raw[p] = x[rr * d + (c0 + p < d ? c0 + p : d - 1)];
#pragma unroll
      for (int p = 0; p < 2 * MAXP; ++p)
        if (p < 2 * npair) tgt[p] = (float)raw[p];
    }
  }
#pragma unroll
  for (int p = 0; p < 2 * MAXP; ++p)
    if (p < 2 * npair) {
      const int c = c0 + p;
      float out = 0.f;
      if (row < M && c < d) {
        const float xh = bf16_round(acc[p]) + bdv[p];
        out = xh - tgt[p];                                    // (aux: e_hat - e, the aux decode predicts the residual)
        sq += out * out;
      }
      if (!aux) e[row * d_p + c] = out;
      else dh[row * d_p + c] = out;
    }
  sq = wave_sum(sq);
  if (lane == 0) part[row] = sq;
}

// tkf: [0] aux scale, [1] total_variance, [5] coef = alpha*scale*2/tv
// metrics: [0] fvu, [1] auxk*alpha, [2] mse, [5] dead fraction, [6] multi_topk_fvu
__global__ __launch_bounds__(1024) void topk_finalize_kernel(const double* __restrict__ tv_part, int n_tv,
                                                             const float* __restrict__ e2_part, const float* __restrict__ a2_part,
                                                             const float* __restrict__ m2_part,
                                                             int64_t Mp, int64_t M, int d, float auxk_alpha, const int* tk,
                                                             float* __restrict__ tkf, float* __restrict__ metrics,
                                                             float dead_frac_n, const double* __restrict__ gstats, int world) {
  __shared__ double red[4][16];
  double a = 0, b = 0, c = 0, m = 0;
  const bool have_aux = a2_part != nullptr && tk[0] > 0;     // with no dead latent the AuxK kernels did not run
#pragma unroll 8      // (one workgroup: eight trips' loads in flight, same order of the sums)
  for (int i = threadIdx.x; i < n_tv; i += 1024) a += tv_part[i];
  // (the two run-time switches as template tags, eight trips' loads in flight: tested per trip they were a branch and a wait per load --
  // 64 trips per thread at M = 65 536, 38 us with AuxK active in this one-workgroup kernel.  Same order of the sums.)
  auto rows = [&](auto aux_tag, auto multi_tag) {
    constexpr bool AUX = decltype(aux_tag)::value, MULTI = decltype(multi_tag)::value;
    int64_t i = threadIdx.x;
    for (; i + 7 * 1024 < Mp; i += 8 * 1024) {
      float ev[8], av[8], mv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        ev[u] = e2_part[i + u * 1024];
        av[u] = AUX ? a2_part[i + u * 1024] : 0.f;
        mv[u] = MULTI ? m2_part[i + u * 1024] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        b += (double)ev[u];
        if (AUX) c += (double)av[u];
        if (MULTI) m += (double)mv[u];
      }
    }
    for (; i < Mp; i += 1024) {
      b += (double)e2_part[i];
      if (AUX) c += (double)a2_part[i];
      if (MULTI) m += (double)m2_part[i];
    }
  };
  if (have_aux) { if (m2_part) rows(std::true_type{}, std::true_type{}); else rows(std::true_type{}, std::false_type{}); }
  else          { if (m2_part) rows(std::false_type{}, std::true_type{}); else rows(std::false_type{}, std::false_type{}); }
  a = wave_sum_d(a);
  b = wave_sum_d(b);
  c = wave_sum_d(c);
  m = wave_sum_d(m);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = a;
    red[1][threadIdx.x >> 6] = b;
    red[2][threadIdx.x >> 6] = c;
    red[3][threadIdx.x >> 6] = m;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double tv = 0, e2 = 0, a2 = 0, m2 = 0;
    for (int q = 0; q < 16; ++q) {
      tv += red[0][q];
      e2 += red[1][q];
      a2 += red[2][q];
      m2 += red[3][q];
    }
    if (tv == 0) tv = 1.0;
    const float tvf = (float)tv;
    const float scale = tkf[0];
    tkf[1] = tvf;
    const float fvu = (float)e2 / tvf;
    const float auxk = (tk[0] > 0 && a2_part) ? (scale * (float)a2 / tvf) * auxk_alpha : 0.f;
    tkf[5] = (tk[0] > 0 && a2_part) ? auxk_alpha * scale * 2.0f / tvf : 0.f;
    metrics[0] = fvu;
    metrics[1] = auxk;
    // data parallel (gstats != null): tv above is the GLOBAL total variance, so fvu / auxk / multi are this rank's share
    // of the global losses (they SUM over the ranks); mse likewise over the global rows; dead_pct is the same on every
    // rank, so each contributes 1 / world of it
    metrics[2] = (float)(e2 / ((gstats ? gstats[0] : (double)M) * d));
    metrics[3] = 0.f;
    metrics[4] = 0.f;
    metrics[5] = (float)tk[0] / dead_frac_n / (float)(gstats ? world : 1);   // dead_pct (train/dead_pct, train_sae.py:481-485)
    metrics[6] = m2_part ? (float)m2 / tvf : 0.f;   // multi_topk_fvu (topkautoencoder.py:134-140)
    metrics[7] = 0.f;
  }
}

// de = 2 e / tv - de_hat,  de_hat = coef (e_hat - e),  dm = (2/8) e_multi / tv (the multi_topk_fvu / 8 loss term,
// train_sae.py:442)  -> bf16 GEMM operands; column sums for d b_dec
__global__ __launch_bounds__(256) void topk_de_kernel(const float* __restrict__ e, const float* __restrict__ dh,
                                                       const float* __restrict__ tkf, bf16_t* __restrict__ de_b,
                                                       bf16_t* __restrict__ dh_b, float* __restrict__ dbd_part, int64_t Mp,
                                                       int d_p, int rows_per_block, int aux_possible,
                                                       const int* __restrict__ tk, const float* __restrict__ em,
                                                       bf16_t* __restrict__ dm_b) {
  const int use_aux = aux_possible && tk[0] > 0;
  // grid (d_p / 256, ceil(Mp / rows_per_block)); thread = one column, fixed row order
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < Mp ? r0 + rows_per_block : Mp;
  const float two_over_tv = 2.0f / tkf[1], coef = tkf[5];
  float s = 0.f;
  // Eight rows per trip with all their loads in flight, the two run-time switches (AuxK active, multi-TopK) as template tags (round 5).  As a
  // plain loop with the switches tested per row this was: load, s_waitcnt vmcnt(0), branch, store -- one memory latency per row and thread,
  // 99 us for 300 MB at C3.  The sums keep their order.
  auto run = [&](auto aux_tag, auto em_tag) {
    constexpr bool AUX = decltype(aux_tag)::value, EM = decltype(em_tag)::value;
    auto one = [&](int64_t r, float ev, float dv, float mv) {
      float g = ev * two_over_tv, gh = 0.f;
      if (AUX) {
        gh = coef * dv;
        g -= gh;
        dh_b[r * d_p + c] = (bf16_t)gh;
      }
      de_b[r * d_p + c] = (bf16_t)g;
      s += g + gh;                        // d b_dec gets de + de_hat (both decoders add b_dec)
      if (EM) {                           // ... and the multi-TopK decode's gradient
        const float gm = mv * (0.125f * two_over_tv);
        dm_b[r * d_p + c] = (bf16_t)gm;
        s += gm;
      }
    };
    int64_t r = r0;
    for (; r + 8 <= r1; r += 8) {
      float ev[8], dv[8], mv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        ev[u] = e[(r + u) * d_p + c];
        dv[u] = AUX ? dh[(r + u) * d_p + c] : 0.f;
        mv[u] = EM ? em[(r + u) * d_p + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) one(r + u, ev[u], dv[u], mv[u]);
    }
    for (; r < r1; ++r) one(r, e[r * d_p + c], AUX ? dh[r * d_p + c] : 0.f, EM ? em[r * d_p + c] : 0.f);
  };
  if (c < d_p) {
    if (use_aux) { if (em) run(std::true_type{}, std::true_type{}); else run(std::true_type{}, std::false_type{}); }
    else         { if (em) run(std::false_type{}, std::true_type{}); else run(std::false_type{}, std::false_type{}); }
  }
  if (c < d_p) dbd_part[(int64_t)blockIdx.y * d_p + c] = s;
}

// ------------------------------------------------------------------------------------------
// Sparse d pre-activation (replaces the dense [M x n] "ddense" GEMM + mask when d_p == 128 * NPAIR): only the selected
// latents of a row carry gradient (topkautoencoder.py:79-91), so for every selected latent j of row m
//     da = de[m] . W_dec[idx_j]      (bf16 operands, fp32 dot, one rounding to bf16 like the GEMM it replaces)
//     dpre[m][idx_j] (+)= da  if its activation is > 0 (ReLU gate),   d b_enc[idx_j] += da
// with the same gather as the sparse decode (k rows of W_dec per activation row instead of all n).  The aux selection
// (dead latents, de_hat) is added into the same dense row afterwards by the same lane, so the result does not depend on
// scheduling.  d b_enc is accumulated with 64-bit integer atomics in 2^-40 fixed point: exact, hence order-independent
// (run-to-run deterministic), unlike float atomics.  dpre must be zero-filled before the launch.
// One wave per activation row; lane l owns the 2 * NPAIR contiguous columns starting at l * 2 * NPAIR.
// ------------------------------------------------------------------------------------------
constexpr double TOPK_FX_SCALE = 1099511627776.0;   // 2^40

// The decodes that feed `pre` (in autograd's execution order: multi-TopK, AuxK, main -- each adds into the bf16 gradient
// of `pre` with one rounding, which the read-add-write below reproduces)
struct DactsPasses {
  const bf16_t* g[3];       // d output of the decode [M_p][d_p] (null = pass absent)
  const bf16_t* vals[3];    // its selected activations, compact [M_p][kcap]
  const int* idx[3];        // its index list [M_p][kcap]
  int kcap[3];
  int gated[3];             // 1: runs only while tk[0] > 0 (AuxK)
};

template <int NPAIR>
__global__ __launch_bounds__(256) void topk_dacts_kernel(DactsPasses ps, const bf16_t* __restrict__ Wd, bf16_t* __restrict__ dpre,
                                                          long long* __restrict__ dbe_fx, int64_t M, int n_p,
                                                          const int* __restrict__ tk) {
  constexpr int d_p = 128 * NPAIR;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + w;
  if (row >= M) return;                                // wave-uniform
  const int c0 = lane * 2 * NPAIR;
  for (int pass = 0; pass < 3; ++pass) {
    const bf16_t* gsrc = ps.g[pass];
    if (!gsrc || (ps.gated[pass] && tk[0] <= 0)) continue;      // the AuxK pass needs dead latents (device-side decision)
    const int kcap = ps.kcap[pass];
    const bf16_t* rv = ps.vals[pass] + row * kcap;
    const int* ri = ps.idx[pass] + row * kcap;
    float g[2 * NPAIR];
    {
      const unsigned* gp = reinterpret_cast<const unsigned*>(gsrc + row * d_p + c0);
#pragma unroll
      for (int p = 0; p < NPAIR; ++p) {
        const unsigned u = gp[p];
        g[2 * p] = __uint_as_float(u << 16);
        g[2 * p + 1] = __uint_as_float(u & 0xFFFF0000u);
      }
    }
    bf16_t* out = dpre + row * n_p;
    for (int j0 = 0; j0 < kcap; j0 += 64) {
      const int jj = j0 + lane;
      const int my_i = jj < kcap ? ri[jj] : -1;
      const float my_a = my_i >= 0 ? (float)rv[jj] : 0.f;
      const int cnt = kcap - j0 < 64 ? kcap - j0 : 64;
      for (int j = 0; j < cnt; j += 2) {
        const int j1 = j + 1 < cnt ? j + 1 : j;
        const int i0 = __shfl(my_i, j, 64), i1 = __shfl(my_i, j1, 64);
        const float a0 = __shfl(my_a, j, 64), a1 = j + 1 < cnt ? __shfl(my_a, j1, 64) : 0.f;
        const unsigned* w0 = reinterpret_cast<const unsigned*>(Wd + (int64_t)(i0 >= 0 ? i0 : 0) * d_p + c0);
        const unsigned* w1 = reinterpret_cast<const unsigned*>(Wd + (int64_t)(i1 >= 0 ? i1 : 0) * d_p + c0);
        unsigned u0[NPAIR], u1[NPAIR];
#pragma unroll
        for (int p = 0; p < NPAIR; ++p) u0[p] = w0[p];
#pragma unroll
        for (int p = 0; p < NPAIR; ++p) u1[p] = w1[p];
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int p = 0; p < NPAIR; ++p) {
          s0 += g[2 * p] * __uint_as_float(u0[p] << 16) + g[2 * p + 1] * __uint_as_float(u0[p] & 0xFFFF0000u);
          s1 += g[2 * p] * __uint_as_float(u1[p] << 16) + g[2 * p + 1] * __uint_as_float(u1[p] & 0xFFFF0000u);
        }
        s0 = wave_sum(s0);
        s1 = wave_sum(s1);
        if (lane == 0) {      // dpre was zero-filled before the launch: every pass adds (bf16, one rounding per addition)
          if (i0 >= 0 && a0 > 0.f) {
            out[i0] = (bf16_t)(bf16_round(s0) + (float)out[i0]);
            atomicAdd(reinterpret_cast<unsigned long long*>(dbe_fx + i0),
                      (unsigned long long)__double2ll_rn((double)bf16_round(s0) * TOPK_FX_SCALE));
          }
          if (j + 1 < cnt && i1 >= 0 && a1 > 0.f) {
            out[i1] = (bf16_t)(bf16_round(s1) + (float)out[i1]);
            atomicAdd(reinterpret_cast<unsigned long long*>(dbe_fx + i1),
                      (unsigned long long)__double2ll_rn((double)bf16_round(s1) * TOPK_FX_SCALE));
          }
        }
      }
    }
  }
}

// d b_enc: the encoder bias enters the bf16 addmm as a bf16 cast, so its gradient is a bf16 value (one rounding of the
// column sum); `exact` keeps the unrounded sums for the d b_dec GEMV (topk_dsae_colsum_kernel).
__global__ __launch_bounds__(256) void topk_dbe_from_fx_kernel(const long long* __restrict__ fx, float* __restrict__ gbe,
                                                                float* __restrict__ exact, int n_p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n_p) {
    const float v = (float)((double)fx[i] / TOPK_FX_SCALE);
    exact[i] = v;
    gbe[i] = bf16_round(v);
  }
}

// ddense epilogue: dpre = [selected] * bf16(de . W_dec^T) (+ aux part), gated by pre > 0; column sums -> d b_enc
struct EpiTopkDpre {
  static constexpr bool ROUNDS_BF16_FIRST = true;     // gemm256.h: the tile goes through LDS as bf16
  static constexpr int PREFETCH_BATCH = EPI_BATCH_HEAVY;
  const bf16_t* sel;    // masked dense activations of this pass (selection mask = value > 0)
  bf16_t* dpre;         // [M_p][n_p]
  float* dbe_part;      // [nbm][n_p] (only written when `last`)
  int n_p, accumulate, last;
  float colsum[4];
  int row_tile, col0_;
  __device__ void tile_begin(int row0, int col0, int) {
    colsum[0] = colsum[1] = colsum[2] = colsum[3] = 0.f;
    row_tile = row0 / GEMM_BM;
    col0_ = col0;
  }
  struct Pre { bf16x4 sv, prev; };
  __device__ Pre prefetch(int row, int col) const {
    const int64_t o = (int64_t)row * n_p + col;
    Pre p;
    p.sv = *reinterpret_cast<const bf16x4*>(sel + o);
    p.prev = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    if (accumulate) p.prev = *reinterpret_cast<const bf16x4*>(dpre + o);
    return p;
  }
  __device__ void apply(int row, int col, f32x4 v, const Pre& pre) {
    const int64_t o = (int64_t)row * n_p + col;
    const bf16x4 sv = pre.sv, prev = pre.prev;
    bf16x4 out;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float g = ((float)sv[j] > 0.f) ? bf16_round(v[j]) : 0.f;
      g += (float)prev[j];
      colsum[j] += g;
      out[j] = (bf16_t)g;
    }
    *reinterpret_cast<bf16x4*>(dpre + o) = out;
  }
  __device__ void tile_end(float* scratch) {
    if (!last) return;
    const int t = threadIdx.x & 255;
    f32x4 cs = {colsum[0], colsum[1], colsum[2], colsum[3]};
    *reinterpret_cast<f32x4*>(scratch + (t >> 5) * 128 + (t & 31) * 4) = cs;
    lds_barrier();
    if (t < 128) {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) s += scratch[g * 128 + t];
      dbe_part[(int64_t)row_tile * n_p + col0_ + t] = s;
    }
  }
};

// dsae_in epilogue: only its column sums are needed (d b_dec -= sum_rows dsae_in, topkautoencoder.py:74)
struct EpiTopkDsaeIn {
  float* part;          // [nbm][d_p]
  int d_p;
  float colsum[4];
  int row_tile, col0_;
  __device__ void tile_begin(int row0, int col0, int) {
    colsum[0] = colsum[1] = colsum[2] = colsum[3] = 0.f;
    row_tile = row0 / GEMM_BM;
    col0_ = col0;
  }
  struct Pre {};
  __device__ Pre prefetch(int, int) const { return Pre{}; }
  __device__ void apply(int, int, f32x4 v, const Pre&) {
#pragma unroll
    for (int j = 0; j < 4; ++j) colsum[j] += bf16_round(v[j]);
  }
  __device__ void tile_end(float* scratch) {
    const int t = threadIdx.x & 255;
    f32x4 cs = {colsum[0], colsum[1], colsum[2], colsum[3]};
    *reinterpret_cast<f32x4*>(scratch + (t >> 5) * 128 + (t & 31) * 4) = cs;
    lds_barrier();
    if (t < 128) {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) s += scratch[g * 128 + t];
      part[(int64_t)row_tile * d_p + col0_ + t] = s;
    }
  }
};

// sum over rows of dsae_in = dpre W_enc without the [M][d] GEMM: sum_m sum_n dpre[m][n] We[n][c] = sum_n dbe[n] We[n][c]
// (dbe = column sums of dpre = the encoder-bias gradient).  Partials over 64-latent chunks, fixed order.
__global__ __launch_bounds__(256) void topk_dsae_colsum_kernel(const float* __restrict__ dbe, const bf16_t* __restrict__ We_b,
                                                                float* __restrict__ part, int n_p, int d_p) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int n0 = blockIdx.y * 64;
  if (c >= d_p) return;
  float s = 0.f;
  const int cnt = n_p - n0 < 64 ? n_p - n0 : 64;
#pragma unroll 8      // (eight trips' loads in flight, same order of the sum)
  for (int i = 0; i < cnt; ++i) s += dbe[n0 + i] * (float)We_b[(int64_t)(n0 + i) * d_p + c];
  part[(int64_t)blockIdx.y * d_p + c] = s;
}

// d b_dec[c] = sum_blocks dbd_part[.][c] - sum_tiles dsae_part[.][c]: 64 columns x 16 waves per workgroup, each wave a
// slice of the partial rows, combined through LDS in wave order (fixed order -> deterministic)
__global__ __launch_bounds__(1024) void topk_dbd_kernel(const float* __restrict__ dbd_part, int nb, const float* __restrict__ ds_part,
                                                         int nt, float* __restrict__ out, int d_p) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (c < d_p) {
#pragma unroll 8
    for (int i = w; i < nb; i += 16) s += dbd_part[(int64_t)i * d_p + c];
#pragma unroll 8
    for (int i = w; i < nt; i += 16) s -= ds_part[(int64_t)i * d_p + c];
  }
  red[w][lane] = s;
  __syncthreads();
  if (w == 0 && c < d_p) {
    float t = 0.f;
    for (int ww = 0; ww < 16; ++ww) t += red[ww][lane];
    out[c] = t;
  }
}

// num_frames_since_fired += rows; [did_fire] = 0 (train_sae.py:443-446).  did_fire may be a data-parallel SUM.
__global__ __launch_bounds__(256) void nfsf_update_kernel(long long* __restrict__ nfsf, const float* __restrict__ did_fire, int n,
                                                           long long rows, const double* __restrict__ gstats) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (gstats) rows = (long long)gstats[0];            // data parallel: the rows of ALL ranks (activations.shape[0] * shape[1])
  if (i < n) nfsf[i] = did_fire[i] > 0.f ? 0 : nfsf[i] + rows;
}
