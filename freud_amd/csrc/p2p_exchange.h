// Peer exchange: all-reduce (sum) of gradient ranges and batch statistics over hipIpc peer mappings -- the "direct
// reduce-scatter + all-gather over the full xGMI mesh" of SURVEY.md section 5 / 8e.  The reference has no counterpart
// (train_sae.py:448-450 is single-device); this sits between loss.backward() and clip_grad_norm_ of a data-parallel run.
//
// Why not a ring: xGMI is a full mesh of point-to-point links, so every rank can pull 1/R of the payload from each of its
// R - 1 peers at once (all links busy, one hop), where a ring moves 2 (R - 1) / R of it over ONE link per hop.  For the
// 4.7 MB gradient of the d = 384 model the exchange is latency-bound: three flag round trips and two short bursts.
//
// One kernel, `nblocks` workgroups, the same grid on every rank.  The flattened element space of the call's segments is cut
// into R shards (rank q owns shard q) and every shard into `nblocks` slices; workgroup b of every rank works on slice b of
// every shard, so workgroup b only ever depends on workgroup b of its peers and the ranks synchronise per workgroup:
//   barrier 0  every rank's inputs are final (the producing kernels precede this one in stream order; a kernel boundary is a
//              system-scope release) and nobody still reads the previous call's results;
//   phase 1    reduce-scatter: rank q sums slice (q, b) over the peers' buffers IN RANK ORDER (every rank would compute the
//              same bits; only the owner does) and writes the sum over its own copy;
//   barrier 1
//   phase 2    all-gather: every rank copies slice (p, b) from its owner p -- the replicas end up bit-identical;
//   barrier 2  nobody overwrites its buffer (next step's reduce_grads) while a peer still reads it.
// (A world of one rank -- the single-GPU test hook -- has nothing to fetch and nobody to wait for after phase 1.)
// Flags live in device memory allocated uncached (hipDeviceMallocUncached) on every rank; a rank writes its arrival into the
// PEERS' flag blocks with a system-scope release store and polls its OWN block with system-scope acquire loads.  Flag values
// are the call's epoch (monotonic, 64 bit): no resets, no ABA.  A poll that outlasts `timeout_ticks` of the 100 MHz
// s_memrealtime clock sets the status word and gives up instead of hanging the GPU (a dead peer must not take the node down).
//
// Payloads: P2P_F32 in place on fp32; P2P_F64 in place on doubles (statistics); P2P_BF16 reads the peers' bf16 COPY of the
// segment (written by reduce_grads), sums in fp32, rounds ONCE, publishes the rounded sum through its own bf16 copy and writes
// fp32 results: half the link bytes of fp32 in both phases, one rounding (a bf16 ring all-reduce rounds at every hop).
// The reference's CPU autocast rounds exactly these weight gradients to bf16 as well.
#pragma once
#include "common.h"

constexpr int P2P_MAX_WORLD = 8;
constexpr int P2P_MAX_BLOCKS = 64;
constexpr int P2P_MAX_SEGS = 4;
constexpr int P2P_THREADS = 512;
constexpr int P2P_CHANNELS = 2;                        // 0: statistics, 1: gradients (each its own flags and epoch)
constexpr int P2P_SIG_WORDS = P2P_CHANNELS * 3 * P2P_MAX_BLOCKS * P2P_MAX_WORLD;    // 64-bit words per rank
enum { P2P_F32 = 0, P2P_F64 = 1, P2P_BF16 = 2 };

// A 2-D block of a row-major buffer: `rows` rows of `cols` elements, `pitch` elements apart, first element `off`.
// cols and off are multiples of 4 (two for doubles), so a 16-byte vector never straddles a row.
struct P2PSeg {
  int64_t off, pitch;
  int rows, cols;
  int kind;             // P2P_F32 / P2P_F64 / P2P_BF16
  int in_norm;          // 1: the reduced values enter gn_part (parameter gradients; not the loss scalars / did_fire flags)
};

struct P2PArgs {
  void* buf[P2P_MAX_WORLD];                  // the fp32 (or fp64) buffer of every rank; buf[rank] is local
  bf16_t* bbuf[P2P_MAX_WORLD];               // bf16 copies (P2P_BF16 segments only)
  unsigned long long* sig[P2P_MAX_WORLD];    // flag blocks; sig[rank] is local
  P2PSeg seg[P2P_MAX_SEGS];
  int nseg, rank, world, channel;
  unsigned long long epoch;
  unsigned long long timeout_ticks;          // of s_memrealtime (100 MHz)
  unsigned int* status;                      // device word: bit 0 = a barrier timed out
  double* gn_part;                           // optional: per-workgroup sum of squares of the REDUCED values (clip norm)
};

__device__ __forceinline__ void p2p_barrier(const P2PArgs& a, int slot) {
  __syncthreads();                           // the workgroup's loads / stores of the phase are complete (vmcnt(0) + barrier)
  if ((int)threadIdx.x < a.world) {
    const int64_t base = (((int64_t)a.channel * 3 + slot) * P2P_MAX_BLOCKS + blockIdx.x) * P2P_MAX_WORLD;
    __hip_atomic_store(a.sig[threadIdx.x] + base + a.rank, a.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long* mine = a.sig[a.rank] + base + threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(mine, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < a.epoch) {
      __builtin_amdgcn_s_sleep(2);
      if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
        atomicOr(a.status, 1u);
        break;
      }
    }
  }
  __syncthreads();
}

__device__ __forceinline__ double p2p_sq(const f32x4& o) {
  return (double)(o[0] * o[0]) + (double)(o[1] * o[1]) + (double)(o[2] * o[2]) + (double)(o[3] * o[3]);
}

// The exchange kernel.  EPV = elements per 16-byte vector: 4 = fp32 and bf16-payload segments (mixed per segment),
// 2 = fp64 segments.  Every SEGMENT is cut into `world` shards of `gridDim.x` slices on its own (a 16-byte vector never
// straddles two segments, and the address of a vector needs no search for its segment); vector indices are 32 bit.
template <int EPV>
__global__ __launch_bounds__(P2P_THREADS) void p2p_allreduce_kernel(P2PArgs a) {
  // sums of squares of the reduced values, kept PER SHARD: they are added in shard order at the end, not in the order the
  // phases ran (own shard first -- different on every rank), so every replica gets the same clip coefficient bit for bit
  double ss_own = 0, ss[P2P_MAX_WORLD];
#pragma unroll
  for (int q = 0; q < P2P_MAX_WORLD; ++q) ss[q] = 0;

  // vectors [v0, v1) of slice (q, blockIdx.x) of segment g; address of vector v of segment g
  auto slice_of = [&](const P2PSeg& g, int q, unsigned& v0, unsigned& v1) {
    const unsigned total = (unsigned)g.rows * (unsigned)(g.cols / EPV);
    const unsigned shard = (total + a.world - 1) / a.world, slice = (shard + gridDim.x - 1) / gridDim.x;
    v0 = q * shard + blockIdx.x * slice;
    v1 = v0 + slice;
    if (v1 > (q + 1) * shard) v1 = (q + 1) * shard;
    if (v1 > total) v1 = total;
  };
  auto elem_of = [&](const P2PSeg& g, unsigned v) -> int64_t {
    if (g.rows == 1) return g.off + (int64_t)v * EPV;
    const unsigned vpr = (unsigned)(g.cols / EPV), row = v / vpr;
    return g.off + (int64_t)row * g.pitch + (int64_t)(v - row * vpr) * EPV;
  };

  p2p_barrier(a, 0);

  // ---- phase 1: the owner's sum of slice (rank, b) over every rank's copy, in rank order
  for (int sgi = 0; sgi < a.nseg; ++sgi) {
    const P2PSeg g = a.seg[sgi];
    unsigned v0, v1;
    slice_of(g, a.rank, v0, v1);
    for (unsigned v = v0 + threadIdx.x; v < v1; v += P2P_THREADS) {
      const int64_t e = elem_of(g, v);
      if constexpr (EPV == 2) {
        typedef __attribute__((ext_vector_type(2))) double f64x2;
        f64x2 acc = *reinterpret_cast<const f64x2*>(reinterpret_cast<const double*>(a.buf[0]) + e);
        for (int p = 1; p < a.world; ++p) acc += *reinterpret_cast<const f64x2*>(reinterpret_cast<const double*>(a.buf[p]) + e);
        *reinterpret_cast<f64x2*>(reinterpret_cast<double*>(a.buf[a.rank]) + e) = acc;
      } else {
        f32x4 o;
        if (g.kind == P2P_BF16) {
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          for (int p = 0; p < a.world; ++p) {
            const bf16x4 t = *reinterpret_cast<const bf16x4*>(a.bbuf[p] + e);
            acc += f32x4{(float)t[0], (float)t[1], (float)t[2], (float)t[3]};
          }
          const bf16x4 r = {(bf16_t)acc[0], (bf16_t)acc[1], (bf16_t)acc[2], (bf16_t)acc[3]};
          *reinterpret_cast<bf16x4*>(a.bbuf[a.rank] + e) = r;
          o = f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
        } else {
          o = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.buf[0]) + e);
          for (int p = 1; p < a.world; ++p) o += *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.buf[p]) + e);
        }
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.buf[a.rank]) + e) = o;
        if (g.in_norm) ss_own += p2p_sq(o);
      }
    }
  }

  if (a.world > 1) {
    p2p_barrier(a, 1);

    // ---- phase 2: fetch every other shard's slice b from its owner
#pragma unroll
    for (int q = 0; q < P2P_MAX_WORLD; ++q) {
      if (q >= a.world || q == a.rank) continue;
      for (int sgi = 0; sgi < a.nseg; ++sgi) {
        const P2PSeg g = a.seg[sgi];
        unsigned v0, v1;
        slice_of(g, q, v0, v1);
        for (unsigned v = v0 + threadIdx.x; v < v1; v += P2P_THREADS) {
          const int64_t e = elem_of(g, v);
          if constexpr (EPV == 2) {
            typedef __attribute__((ext_vector_type(2))) double f64x2;
            *reinterpret_cast<f64x2*>(reinterpret_cast<double*>(a.buf[a.rank]) + e) =
                *reinterpret_cast<const f64x2*>(reinterpret_cast<const double*>(a.buf[q]) + e);
          } else {
            f32x4 o;
            if (g.kind == P2P_BF16) {
              const bf16x4 r = *reinterpret_cast<const bf16x4*>(a.bbuf[q] + e);
              o = f32x4{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
            } else {
              o = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.buf[q]) + e);
            }
            *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.buf[a.rank]) + e) = o;
            if (g.in_norm) ss[q] += p2p_sq(o);
          }
        }
      }
    }

    p2p_barrier(a, 2);
  }

  if (EPV == 4 && a.gn_part) {
    __shared__ double red[P2P_THREADS / 64];
    double s = 0;
#pragma unroll
    for (int q = 0; q < P2P_MAX_WORLD; ++q) s += (q == a.rank) ? ss_own : ss[q];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0;
#pragma unroll
      for (int w = 0; w < P2P_THREADS / 64; ++w) t += red[w];
      a.gn_part[blockIdx.x] = t;
    }
  }
}

// Column range [c0, c0 + cols) of the fused backward's partial slabs -> gradient buffer: dW rows [0, d_p) of that range plus
// the bias entries of the range (row d_p of the index space), summed over the row-range partials in fixed order; optional bf16
// copy for the exchange and per-workgroup sum of squares (the clip norm of a single-GPU step).  The whole-buffer form is
// reduce_grads_kernel (l1_kernels.h); this one lets the exchange of one column range run under the backward of the next.
__global__ __launch_bounds__(256) void reduce_grads_range_kernel(const float* __restrict__ slab, int64_t slab_stride, int splits,
                                                                  const float* __restrict__ db_part, int db_rows, int n_p, int d_p,
                                                                  int c0, int cols, float* __restrict__ grad,
                                                                  double* __restrict__ gn_part, bf16_t* __restrict__ grad_bf16) {
  __shared__ double red[4];
  double ss = 0;
  const int vpr = cols / 4;
  const int64_t nv = (int64_t)(d_p + 1) * vpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / vpr;
    const int col = c0 + (int)(i - row * vpr) * 4;
    f32x4 a;
    int64_t o;
    if (row < d_p) {
      o = row * n_p + col;
      a = *reinterpret_cast<const f32x4*>(slab + o);
      for (int k = 1; k < splits; ++k) a += *reinterpret_cast<const f32x4*>(slab + (int64_t)k * slab_stride + o);
    } else {
      o = (int64_t)d_p * n_p + col;
      a = *reinterpret_cast<const f32x4*>(db_part + col);
      for (int k = 1; k < db_rows; ++k) a += *reinterpret_cast<const f32x4*>(db_part + (int64_t)k * n_p + col);
    }
    *reinterpret_cast<f32x4*>(grad + o) = a;
    if (grad_bf16) *reinterpret_cast<bf16x4*>(grad_bf16 + o) = bf16x4{(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3]};
    ss += (double)(a[0] * a[0]) + (double)(a[1] * a[1]) + (double)(a[2] * a[2]) + (double)(a[3] * a[3]);
  }
  ss = wave_sum_d(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x == 0) gn_part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// dW rows [0, d_p) x columns [c0, c0 + cols) of the generic weight-gradient GEMM's split-K slabs -> gradient buffer (the
// column chunks of the d >= 1024 path under the peer exchange)
__global__ __launch_bounds__(256) void reduce_slabs_range_kernel(const float* __restrict__ slab, int64_t slab_stride, int splits, int n_p,
                                                                  int d_p, int c0, int cols, float* __restrict__ grad) {
  const int vpr = cols / 4;
  const int64_t nv = (int64_t)d_p * vpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / vpr;
    const int64_t o = row * n_p + c0 + (i - row * vpr) * 4;
    f32x4 a = *reinterpret_cast<const f32x4*>(slab + o);
    for (int k = 1; k < splits; ++k) a += *reinterpret_cast<const f32x4*>(slab + (int64_t)k * slab_stride + o);
    *reinterpret_cast<f32x4*>(grad + o) = a;
  }
}

// Self-test pattern of sae_p2p_init: element i of rank r = pattern(r, i); after the exchange every element must be the sum
// over the ranks.  Small integers: exact in fp32 and in bf16 (|value| < 256).
__global__ void p2p_selftest_fill_kernel(float* buf, bf16_t* bbuf, int64_t n, int rank) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = (float)((i * 7 + rank * 3) % 29 - 14);
    buf[i] = v;
    if (bbuf) bbuf[i] = (bf16_t)v;
  }
}
__global__ void p2p_selftest_check_kernel(const float* buf, int64_t n, int world, unsigned int* bad) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float want = 0.f;
    for (int r = 0; r < world; ++r) want += (float)((i * 7 + r * 3) % 29 - 14);
    if (buf[i] != want) atomicAdd(bad, 1u);
  }
}
