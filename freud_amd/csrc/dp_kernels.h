// Data-parallel exactness: the statistics of a batch that the losses NORMALISE by, as doubles in one device buffer that
// is summed over the ranks before the step's loss / backward kernels read it, so that R ranks x batch B compute exactly
// the gradient of one rank x batch R B (the reference itself is single-process):
//   L1   : stats[0] = number of unmasked entries (x != -1: the masked-MSE denominator, l1autoencoder.py:29-36),
//          stats[1] = rows (the mean over rows of the L1 term, l1autoencoder.py:85);
//   TopK : stats[0] = rows, stats[1] = files B, stats[2 + j] = sum_b x[b][j], stats[2 + TD + j] = sum_b x[b][j]^2 for every
//          (t, feature) column j < TD = T d: total_variance = sum_j (sum x^2 - (sum x)^2 / B) with the mean over ALL files
//          (x.mean(0), topkautoencoder.py:104-106).
// These depend on the input batch only, never on the model: a rank can compute and all-reduce them while its forward runs.
#pragma once
#include "common.h"

constexpr int DP_STATS_HEAD = 2;

// per-block partial counts of x == -1 over M*d elements (fixed order, no atomics); 16-byte loads when x is aligned
template <typename T>
__global__ __launch_bounds__(256) void dp_count_masked_kernel(const T* __restrict__ x, int64_t total, unsigned int* __restrict__ part) {
  __shared__ unsigned int red[4];
  unsigned int m = 0;
  constexpr int V = 16 / (int)sizeof(T);
  typedef __attribute__((ext_vector_type(V))) T TV;
  const int64_t nvec = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? total / V : 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    const TV v = reinterpret_cast<const TV*>(x)[i];
#pragma unroll
    for (int j = 0; j < V; ++j) m += ((float)v[j] == -1.0f);
  }
  for (int64_t i = nvec * V + (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) m += ((float)x[i] == -1.0f);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m += (unsigned int)__shfl_xor((int)m, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void dp_l1_stats_kernel(const unsigned int* __restrict__ part, int nparts, int64_t M, int d,
                                                          double* __restrict__ stats) {
  __shared__ unsigned long long red[4];
  unsigned long long m = 0;
  for (int i = threadIdx.x; i < nparts; i += 256) m += part[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m += (unsigned long long)__shfl_xor((long long)m, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    stats[0] = (double)M * d - (double)(red[0] + red[1] + red[2] + red[3]);
    stats[1] = (double)M;
  }
}

// TopK: column sums and sums of squares over the B files of x viewed as [B][TD]; one thread per column, fixed order
template <typename T>
__global__ __launch_bounds__(256) void dp_topk_stats_kernel(const T* __restrict__ x, int B, int64_t TD, int64_t M,
                                                            double* __restrict__ stats) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j == 0) {
    stats[0] = (double)M;
    stats[1] = (double)B;
  }
  if (j >= TD) return;
  double s1 = 0, s2 = 0;
  for (int b = 0; b < B; ++b) {
    const double v = (double)(float)x[(int64_t)b * TD + j];
    s1 += v;
    s2 += v * v;
  }
  stats[DP_STATS_HEAD + j] = s1;
  stats[DP_STATS_HEAD + TD + j] = s2;
}

// total_variance partials from the (summed) column statistics: sum_j (s2_j - s1_j^2 / B)
__global__ __launch_bounds__(256) void dp_topk_tv_kernel(const double* __restrict__ stats, int64_t TD, double* __restrict__ part) {
  __shared__ double red[4];
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  double s = 0;
  if (j < TD) {
    const double B = stats[1], s1 = stats[DP_STATS_HEAD + j], s2 = stats[DP_STATS_HEAD + TD + j];
    s = s2 - s1 * s1 / B;
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// Order-independent 64-bit checksum of an fp32 buffer's BITS (sae_param_checksum): sum over i of mix(bits[i] ^ i * K1) -- integer
// adds commute, so the atomics give the same word whatever the order.  Replicas of a data-parallel run hold bit-identical
// parameters and optimizer moments by construction; train() compares these words across the ranks.
__global__ __launch_bounds__(256) void checksum_kernel(const float* __restrict__ p, int64_t n, unsigned long long* __restrict__ out) {
  unsigned long long h = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    unsigned long long v = (unsigned long long)__float_as_uint(p[i]) ^ ((unsigned long long)i * 0x9E3779B97F4A7C15ull);
    v ^= v >> 29;
    v *= 0xD6E8FEB86659FD93ull;
    v ^= v >> 32;
    h += v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) h += __shfl_xor(h, off, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(out, h);
}

