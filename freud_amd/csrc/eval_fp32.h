// fp32 evaluation forward: the arithmetic of the reference's validate() on device='cpu' (src/scripts/train_sae.py:162-166:
// `nullcontext()` instead of autocast, so L1AutoEncoder.forward / TopKAutoEncoder.forward run in fp32 end to end).  The training
// kernels (and the default evaluation) compute in bf16 with fp32 accumulation -- CPU autocast's arithmetic -- which agrees with that
// path to ~1e-2 only; `bestval.pth` selection (train_sae.py:585-595) rests on these numbers, so sae_set_eval_precision(SAE_PREC_FP32)
// makes sae_eval / sae_eval_into run THIS path: fp32 operands on the fp32 matrix instruction (v_mfma_f32_32x32x2_f32, fp32
// accumulate), fp32 bias / ReLU / selection, losses summed in double.  A validation file is 1500 rows: speed is not the point here
// (a plain 64 x 64 LDS-tiled GEMM, ~20-40 TFLOP/s), faithfulness is.
//
//   L1   (l1autoencoder.py:69-95):  c = relu(x W + b);  x_hat = c W^T;  l1 = mean_rows sum_j |c|;  recon = alpha * masked mse;  mse
//   TopK (topkautoencoder.py:72-151): pre = relu((x - b_dec) W_enc^T + b_enc);  top-k per row (ties: lowest column first, the engine's
//        rule);  x_hat = dense W_dec + b_dec;  fvu = sum e^2 / total_variance;  multi-TopK with 4k;  no AuxK term (validate() passes no
//        dead mask: :171)
#pragma once
#include "common.h"

constexpr int E32_BM = 64, E32_BN = 64, E32_BK = 16, E32_LD = 68;
constexpr int E32_RES_BLOCKS = 512;                                  // workgroups (and partial sums) of the residual / variance passes
constexpr int E32_L1_PARTS = 16384;                                   // (latent blocks of 256 columns x 64 rows: 1500 rows x 163 840 latents)
constexpr int E32_PART_DOUBLES = E32_L1_PARTS + 2 * 4 * E32_RES_BLOCKS + E32_RES_BLOCKS;

// C[M x N] = A[M x K] . B,  A row-major (K contiguous, lda);  B as [K][N] (BT = false, ldb = row pitch of a k-row) or as [N][K]
// (BT = true: the operand is stored transposed, K contiguous).  K a multiple of 16, N a multiple of 64 (padded shapes); M arbitrary.
template <bool BT>
__global__ __launch_bounds__(256) void e32_gemm_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                       float* __restrict__ C, int64_t ldc, int M, int N, int K) {
  __shared__ __attribute__((aligned(16))) float As[E32_BK][E32_LD];      // k-major: lane l of an MFMA reads 32 consecutive rows
  __shared__ __attribute__((aligned(16))) float Bs[E32_BK][E32_LD];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w & 1, wn = w >> 1;
  const int m0 = blockIdx.y * E32_BM, n0 = blockIdx.x * E32_BN;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int lr = t >> 2, lk = (t & 3) * 4;               // A (and a transposed B): row lr, k offset lk
  const int bk = t >> 4, bc = (t & 15) * 4;              // B as [K][N]: k-row bk, column offset bc
  const bool a_ok = m0 + lr < M;
  const float* ap = A + (int64_t)(a_ok ? m0 + lr : 0) * lda + lk;
  const float* bp = BT ? B + (int64_t)(n0 + lr) * ldb + lk : B + (int64_t)bk * ldb + n0 + bc;
  for (int k0 = 0; k0 < K; k0 += E32_BK) {
    f32x4 av = a_ok ? *reinterpret_cast<const f32x4*>(ap + k0) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bv = BT ? *reinterpret_cast<const f32x4*>(bp + k0) : *reinterpret_cast<const f32x4*>(bp + (int64_t)k0 * ldb);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) As[lk + i][lr] = av[i];
    if (BT) {
#pragma unroll
      for (int i = 0; i < 4; ++i) Bs[lk + i][lr] = bv[i];
    } else {
      *reinterpret_cast<f32x4*>(&Bs[bk][bc]) = bv;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < E32_BK / 2; ++kk) {
      const float a = As[2 * kk + (lane >> 5)][32 * wm + (lane & 31)];
      const float b = Bs[2 * kk + (lane >> 5)][32 * wn + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }
  const int col = n0 + 32 * wn + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + 32 * wm + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
    if (row < M && col < N) C[(int64_t)row * ldc + col] = acc[r];
  }
}

// x (fp32 / fp16 / bf16, [M][d]) -> fp32 [M][d_p], zero padded; `sub` (TopK: b_dec) is subtracted where given
template <typename T>
__global__ __launch_bounds__(256) void e32_load_x_kernel(const T* __restrict__ x, float* __restrict__ out, const float* __restrict__ sub,
                                                         int64_t M, int d, int d_p) {
  const int64_t total = M * d_p;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / d_p;
    const int j = (int)(i - r * d_p);
    float v = 0.f;
    if (j < d) {
      v = (float)x[r * d + j];
      if (sub) v -= sub[j];
    }
    out[i] = v;
  }
}

// pre[M][n_p] -> relu(pre + b) in place (columns >= n: zero); per-block sums of the latent (the L1 norm: it is non-negative) in
// double; per-feature maxima by integer atomics on the bits (the values are >= 0: their bit patterns order like the values)
__global__ __launch_bounds__(256) void e32_bias_relu_kernel(float* __restrict__ pre, const float* __restrict__ b, int64_t M, int n, int n_p,
                                                            double* __restrict__ l1_part, int* __restrict__ colmax_bits) {
  __shared__ double red[4];
  const int col = blockIdx.x * 256 + threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.y * 64, r1 = r0 + 64 < M ? r0 + 64 : M;
  double s = 0;
  if (col < n_p) {
    const float bj = col < n ? b[col] : 0.f;
    float mx = 0.f;
    for (int64_t r = r0; r < r1; ++r) {
      float v = col < n ? fmaxf(pre[r * n_p + col] + bj, 0.f) : 0.f;
      pre[r * n_p + col] = v;
      s += (double)v;
      mx = fmaxf(mx, v);
    }
    if (colmax_bits && col < n && mx > 0.f) atomicMax(colmax_bits + col, __float_as_int(mx));
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) l1_part[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// residual sums of one evaluation forward: e = x_hat (+ add[j]) - x over [M][d];  part[4 b + {0,1,2}] = sum e^2 over the entries with
// x != -1.0 (mse_loss, l1autoencoder.py:29-36), sum e^2 over all entries, number of entries with x == -1.0
template <typename T>
__global__ __launch_bounds__(256) void e32_residual_kernel(const float* __restrict__ xhat, int d_p, const float* __restrict__ add,
                                                           const T* __restrict__ x, int64_t M, int d, double* __restrict__ part) {
  __shared__ double red[3][4];
  double sq = 0, pl = 0, nm = 0;
  const int64_t total = M * d;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / d;
    const int j = (int)(i - r * d);
    const float xv = (float)x[i];
    float xh = xhat[r * d_p + j];
    if (add) xh += add[j];
    const float e = xh - xv;
    const double e2 = (double)e * (double)e;
    pl += e2;
    if (xv != -1.0f) sq += e2; else nm += 1.0;
  }
  sq = wave_sum_d(sq); pl = wave_sum_d(pl); nm = wave_sum_d(nm);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sq; red[1][threadIdx.x >> 6] = pl; red[2][threadIdx.x >> 6] = nm; }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[4 * blockIdx.x + 0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    part[4 * blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    part[4 * blockIdx.x + 2] = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
  }
}

// TopK: total_variance = sum (x - x.mean(0))^2 with the mean over the FILES of a [B][T][d] batch (topkautoencoder.py:104-106):
// per (t, feature) position s2 - s1^2 / B in double; validate() runs one file per forward, where this is exactly 0 (and the
// caller then takes 1.0, as the reference does)
template <typename T>
__global__ __launch_bounds__(256) void e32_total_variance_kernel(const T* __restrict__ x, int64_t B, int64_t TD, double* __restrict__ part) {
  __shared__ double red[4];
  double tv = 0;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < TD; p += (int64_t)gridDim.x * 256) {
    double s1 = 0;
    for (int64_t f = 0; f < B; ++f) s1 += (double)(float)x[f * TD + p];
    const float mean = (float)(s1 / (double)B);               // (the reference's mean is an fp32 tensor)
    for (int64_t f = 0; f < B; ++f) {
      const float dv = (float)x[f * TD + p] - mean;
      tv += (double)dv * (double)dv;
    }
  }
  tv = wave_sum_d(tv);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = tv;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// TopK selection on fp32 pre-activations (>= 0 after the ReLU): one workgroup per row.  The k-th largest value by a descent over
// the 31 value bits (non-negative floats order like their bit patterns): "at least k entries >= T".  Kept: every entry above it
// and the first (k - those) entries equal to it in column order -- the engine's tie rule; if the k-th largest is 0 only the
// positive entries (the zeros contribute nothing to the decode).  out = the masked dense row.
__global__ __launch_bounds__(256) void e32_topk_select_kernel(const float* __restrict__ pre, float* __restrict__ out, int n, int n_p, int k,
                                                              int* __restrict__ idx_out /* [M][k] or null */) {
  __shared__ int cnt[4];
  __shared__ int run;
  const int64_t row = blockIdx.x;
  const float* p = pre + row * n_p;
  float* o = out + row * n_p;
  const int t = threadIdx.x;
  auto count_ge = [&](unsigned thr) -> int {
    int c = 0;
    for (int j = t; j < n; j += 256) c += (__float_as_uint(p[j]) >= thr) ? 1 : 0;
    c = (int)wave_sum((float)c);          // (< 2^24: exact in fp32)
    __syncthreads();
    if ((t & 63) == 0) cnt[t >> 6] = c;
    __syncthreads();
    return cnt[0] + cnt[1] + cnt[2] + cnt[3];
  };
  unsigned vk = 0u;
  for (int bit = 30; bit >= 0; --bit) {
    const unsigned trial = vk | (1u << bit);
    if (count_ge(trial) >= k) vk = trial;
  }
  // vk = bits of the k-th largest value (0 if fewer than k entries are positive)
  const int above = vk == 0u ? 0 : count_ge(vk + 1u);
  int need = vk == 0u ? 0 : k - above;               // entries equal to the k-th largest value still to take, lowest column first
  if (t == 0) run = 0;
  __syncthreads();
  for (int j0 = 0; j0 < n_p; j0 += 256) {
    const int j = j0 + t;
    const unsigned u = j < n ? __float_as_uint(p[j]) : 0u;
    const bool tie = vk != 0u && u == vk;
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(tie);
    const int before_in_wave = __builtin_popcountll(bal & ((1ull << (t & 63)) - 1ull));
    if ((t & 63) == 0) cnt[t >> 6] = __builtin_popcountll(bal);
    __syncthreads();
    int before = run + before_in_wave;
    for (int ww = 0; ww < (t >> 6); ++ww) before += cnt[ww];
    const bool keep = (u > vk && u != 0u) || (tie && before < need);
    if (j < n_p) o[j] = keep ? __uint_as_float(u) : 0.f;
    __syncthreads();
    if (t == 0) run += cnt[0] + cnt[1] + cnt[2] + cnt[3];
    __syncthreads();
  }
  (void)idx_out;
}

// per-feature maxima of a non-negative fp32 [M][n_p] array (the selected TopK activations) into integer bit patterns
__global__ __launch_bounds__(256) void e32_colmax_kernel(const float* __restrict__ a, int64_t M, int n, int n_p, int* __restrict__ colmax_bits) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= n) return;
  const int64_t r0 = (int64_t)blockIdx.y * 64, r1 = r0 + 64 < M ? r0 + 64 : M;
  float mx = 0.f;
  for (int64_t r = r0; r < r1; ++r) mx = fmaxf(mx, fabsf(a[r * n_p + col]));
  if (mx > 0.f) atomicMax(colmax_bits + col, __float_as_int(mx));
}

struct E32Final {
  const double *l1_part, *res_part, *res2_part, *tv_part;
  int n_l1, n_res, n_res2, n_tv;
  int64_t M;
  int d;
  float alpha;
  int topk;
  float dead_frac;
};

// the 8 loss scalars of the evaluation forward, in the slots the training path uses (include/freud_sae.h, sae_read_metrics)
__global__ __launch_bounds__(256) void e32_finalize_kernel(E32Final f, float* __restrict__ metrics) {
  __shared__ double red[6][4];
  double l1 = 0, sq = 0, pl = 0, nm = 0, pl2 = 0, tv = 0;
  for (int i = threadIdx.x; i < f.n_l1; i += 256) l1 += f.l1_part[i];
  for (int i = threadIdx.x; i < f.n_res; i += 256) { sq += f.res_part[4 * i]; pl += f.res_part[4 * i + 1]; nm += f.res_part[4 * i + 2]; }
  for (int i = threadIdx.x; i < f.n_res2; i += 256) pl2 += f.res2_part[4 * i + 1];
  for (int i = threadIdx.x; i < f.n_tv; i += 256) tv += f.tv_part[i];
  l1 = wave_sum_d(l1); sq = wave_sum_d(sq); pl = wave_sum_d(pl); nm = wave_sum_d(nm); pl2 = wave_sum_d(pl2); tv = wave_sum_d(tv);
  if ((threadIdx.x & 63) == 0) {
    const int w = threadIdx.x >> 6;
    red[0][w] = l1; red[1][w] = sq; red[2][w] = pl; red[3][w] = nm; red[4][w] = pl2; red[5][w] = tv;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double v[6];
    for (int q = 0; q < 6; ++q) v[q] = (red[q][0] + red[q][1]) + (red[q][2] + red[q][3]);
    const double count = (double)f.M * f.d - v[3];
    if (!f.topk) {
      metrics[0] = (float)((double)f.alpha * (v[1] / count));                // reconstruction_loss = recon_alpha * masked mse
      metrics[1] = (float)(v[0] / (double)f.M);                              // l1_loss
      metrics[2] = (float)(v[2] / ((double)f.M * f.d));                      // return_mse
      metrics[3] = 0.f;
      metrics[4] = (float)count;
      metrics[5] = metrics[6] = metrics[7] = 0.f;
    } else {
      const double tvar = v[5] == 0.0 ? 1.0 : v[5];                          // topkautoencoder.py:105-106
      metrics[0] = (float)(v[2] / tvar);                                     // fvu
      metrics[1] = 0.f;                                                      // auxk_loss: validate() passes no dead mask
      metrics[2] = (float)(v[2] / ((double)f.M * f.d));                      // return_mse
      metrics[3] = 0.f;
      metrics[4] = 0.f;
      metrics[5] = f.dead_frac;
      metrics[6] = f.n_res2 > 0 ? (float)(v[4] / tvar) : 0.f;                // multi_topk_fvu
      metrics[7] = 0.f;
    }
  }
}
