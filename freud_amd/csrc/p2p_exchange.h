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
// PEERS' flag blocks and polls its OWN block.  Flag values are the call's epoch (monotonic, 64 bit): no resets, no ABA.
//
// Visibility ("memory model"; tests/test_p2p_codeobj.py asserts these instructions in the built code
// object).  G / Gb / stats are ordinary (coarse-grained) device memory: local lines live in this XCD's L2 (write-back), a
// PEER's lines mapped through hipIpc are cached non-coherently (MTYPE NC) in the reader's L2.  Every barrier therefore is
//   every storing wave: s_waitcnt vmcnt(0) -> workgroup barrier -> (the signalling lanes) system-scope RELEASE fence =
//   `buffer_wbl2 sc0 sc1` (dirty L2 lines, i.e. the phase's sums, reach memory) + `s_waitcnt vmcnt(0)` (inline asm: hipcc drops
//   the wait when it can prove its scoreboard empty, MI355X_MICROARCH.md "Compiler hazard") -> relaxed system-scope flag stores
//   into the peers' blocks -> relaxed system-scope polls of the own block (uncached memory: every poll reaches memory) ->
//   system-scope ACQUIRE fence = `buffer_inv sc0 sc1` (this CU's L1 and the NC lines of this XCD's L2 are dropped: the peers'
//   lines read in an earlier step are re-fetched) + `s_waitcnt vmcnt(0)` (the invalidate completes asynchronously) ->
//   workgroup barrier -> the phase's plain loads.
//
// Failure: a poll that outlasts `timeout_ticks` of the 100 MHz s_memrealtime clock gives up instead of hanging the GPU (a dead
// peer must not take the node down), sets the status word AND poisons the protocol: the flags it owes the peers for this call
// and for every later one carry P2P_POISON, a poller that reads a poisoned flag fails too (and poisons in turn), a failed
// workgroup skips both phases (its buffer stays as it was), and the status word is sticky -- every later exchange of a failed
// context starts failed.  So no rank can sail through on flags a peer published before it gave up.
//
// Payloads: P2P_F32 in place on fp32; P2P_F64 in place on doubles (statistics); P2P_BF16 reads the peers' bf16 COPY of the
// segment (written by reduce_grads), sums in fp32, rounds ONCE, publishes the rounded sum through its own bf16 copy and writes
// fp32 results: half the link bytes of fp32 in both phases, one rounding (a bf16 ring all-reduce rounds at every hop).
// The reference's CPU autocast rounds exactly these weight gradients to bf16 as well.
#pragma once
#include "common.h"
#include "p2p_index.h"

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
  unsigned int* status;                      // device word, sticky: P2P_ST_TIMEOUT | P2P_ST_POISONED
  unsigned int* status_host;                 // the same word in host-mapped memory (may be null): the step loop polls it for free
  double* gn_part;                           // optional: per-workgroup sum of squares of the REDUCED values (clip norm)
  int fault;                                 // test hook (FREUD_P2P_FAULT): 1 = skip the phase-2 copy of shard (rank + 1) % world
};

constexpr unsigned long long P2P_POISON = 1ull << 63;      // a flag value with this bit: the writer has left the protocol
enum { P2P_ST_TIMEOUT = 1u, P2P_ST_POISONED = 2u };

__device__ __forceinline__ void p2p_mark_failed(unsigned int* status, unsigned int* status_host, unsigned int code) {
  atomicOr(status, code);
  if (status_host) __hip_atomic_store(status_host, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One barrier of the exchange (see the header comment for the instruction sequence and why).  `fail_s`: the workgroup's
// failure word in LDS (set at kernel start from the sticky status word).  Returns false when the protocol has failed.
// FENCES = false (barrier 2): nothing was written for the peers since barrier 1 and nothing of theirs is read afterwards -- the
// barrier only says "I have finished reading your buffer" -- so neither the write-back nor the invalidate is needed.
template <bool FENCES = true>
__device__ __forceinline__ bool p2p_barrier(const P2PArgs& a, int slot, volatile int* fail_s) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave: its stores of the phase have left the wave
  __syncthreads();
  if ((int)threadIdx.x < a.world) {                      // (all in wave 0: world <= 8)
    const bool failed = *fail_s != 0;
    const int64_t blk = (int64_t)blockIdx.x * P2P_MAX_WORLD;
    const int64_t base = ((int64_t)a.channel * 3 + slot) * P2P_MAX_BLOCKS * P2P_MAX_WORLD + blk;
    if (FENCES) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");      // system scope: buffer_wbl2 sc0 sc1
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ... completed BEFORE the flag leaves
    }
    __hip_atomic_store(a.sig[threadIdx.x] + base + a.rank, failed ? (a.epoch | P2P_POISON) : a.epoch, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
    if (!failed) {
      const unsigned long long* mine = a.sig[a.rank] + base + threadIdx.x;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      unsigned long long v;
      unsigned int code = 0;
      while ((v = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) < a.epoch) {
        __builtin_amdgcn_s_sleep(2);
        if (__builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) {
          code = P2P_ST_TIMEOUT;
          break;
        }
      }
      if (!code && (v & P2P_POISON)) code = P2P_ST_POISONED;
      if (code) {
        // leave the protocol for good and tell everybody: the flags this workgroup owes peer `threadIdx.x` for this call
        // (all three barriers -- the peer may wait at any of them) carry the poison; later calls start failed (sticky status)
        *fail_s = 1;
        p2p_mark_failed(a.status, a.status_host, code);
#pragma unroll
        for (int sl = 0; sl < 3; ++sl)
          __hip_atomic_store(a.sig[threadIdx.x] + ((int64_t)a.channel * 3 + sl) * P2P_MAX_BLOCKS * P2P_MAX_WORLD + blk + a.rank,
                             a.epoch | P2P_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    if (FENCES) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");      // system scope: buffer_inv sc0 sc1
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the invalidate has completed before the workgroup is released
    }
  }
  __syncthreads();
  return *fail_s == 0;
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
  // (p2p_index.h: plain index arithmetic, also compiled for the host by tests/test_p2p_index.py, which checks for worlds 1-8 and
  // every grid that the slices of a segment cover it exactly once)
  auto slice_of = [&](const P2PSeg& g, int q, unsigned& v0, unsigned& v1) {
    p2p_slice_of((unsigned)g.rows * (unsigned)(g.cols / EPV), (unsigned)a.world, gridDim.x, (unsigned)q, blockIdx.x, v0, v1);
  };
  auto elem_of = [&](const P2PSeg& g, unsigned v) -> int64_t { return p2p_elem_of(g.off, g.pitch, g.rows, g.cols, EPV, v); };

  __shared__ int fail_s;
  if (threadIdx.x == 0) fail_s = __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;   // sticky
  bool ok = p2p_barrier<true>(a, 0, &fail_s);     // (its first workgroup barrier publishes fail_s)

  // ---- phase 1: the owner's sum of slice (rank, b) over every rank's copy, in rank order
  for (int sgi = 0; ok && sgi < a.nseg; ++sgi) {
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
    ok = p2p_barrier<true>(a, 1, &fail_s);

    // ---- phase 2: fetch every other shard's slice b from its owner
#pragma unroll
    for (int q = 0; q < P2P_MAX_WORLD; ++q) {
      if (!ok || q >= a.world || q == a.rank) continue;
      if (a.fault == 1 && q == (a.rank + 1) % a.world) continue;      // injected fault: this shard keeps the local partials
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

    p2p_barrier<false>(a, 2, &fail_s);
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
// `out` / `out_pitch` / `out_c0`: where the summed block goes -- the gradient buffer itself (pitch n_p, column c0: the peer exchange
// takes the strided block as it is) or a CONTIGUOUS staging block (pitch cols, column 0: what RCCL can sum in one call)
__global__ __launch_bounds__(256) void reduce_slabs_range_kernel(const float* __restrict__ slab, int64_t slab_stride, int splits, int n_p,
                                                                  int d_p, int c0, int cols, float* __restrict__ out, int64_t out_pitch,
                                                                  int out_c0) {
  const int vpr = cols / 4;
  const int64_t nv = (int64_t)d_p * vpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / vpr, cv = (i - row * vpr) * 4;
    const int64_t o = row * n_p + c0 + cv;
    f32x4 a = *reinterpret_cast<const f32x4*>(slab + o);
    for (int k = 1; k < splits; ++k) a += *reinterpret_cast<const f32x4*>(slab + (int64_t)k * slab_stride + o);
    *reinterpret_cast<f32x4*>(out + row * out_pitch + out_c0 + cv) = a;
  }
}

// Self-test of sae_p2p_init (engine.hip: p2p_selftest).  Element i of rank r in exchange number e of payload kind k holds
// pattern(i, r, e, k); after the exchange every element of the exchanged segment must be the sum over the ranks and every
// element outside it must still hold this rank's own pattern.  The pattern CHANGES WITH e while the addresses do not: a peer
// line that stayed in a cache from exchange e - 1 gives a wrong sum in exchange e (round 3 filled the same values twice, so a
// stale line held the right value).  Small integers: exact in fp32, bf16 and fp64 (|value| <= 14, sums < 256).
__device__ __forceinline__ int p2p_pattern(int64_t i, int rank, int e, int kind) {
  return (int)((i * 7 + rank * 3 + e * 5 + kind * 11) % 29) - 14;
}
__global__ void p2p_selftest_fill_kernel(float* buf, bf16_t* bbuf, int64_t n, int rank, int e, int kind) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = (float)p2p_pattern(i, rank, e, kind);
    buf[i] = v;
    if (bbuf) bbuf[i] = (bf16_t)v;
  }
}
__global__ void p2p_selftest_fill64_kernel(double* buf, int64_t n, int rank, int e, int kind) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    buf[i] = (double)p2p_pattern(i, rank, e, kind);
}
// seg: the exchanged block (rows x cols, pitch, off) of the buffer of n elements
__global__ void p2p_selftest_check_kernel(const float* buf, int64_t n, P2PSeg seg, int rank, int world, int e, int kind, unsigned int* bad) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t rel = i - seg.off;
    const bool inside = rel >= 0 && rel / seg.pitch < seg.rows && rel % seg.pitch < seg.cols;
    float want = 0.f;
    if (inside) for (int r = 0; r < world; ++r) want += (float)p2p_pattern(i, r, e, kind);
    else want = (float)p2p_pattern(i, rank, e, kind);
    if (buf[i] != want) atomicAdd(bad, 1u);
  }
}
__global__ void p2p_selftest_check64_kernel(const double* buf, int64_t n, int world, int e, int kind, unsigned int* bad) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double want = 0;
    for (int r = 0; r < world; ++r) want += (double)p2p_pattern(i, r, e, kind);
    if (buf[i] != want) atomicAdd(bad, 1u);
  }
}
