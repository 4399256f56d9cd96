// Sparse backward of the TopK SAE (reference: autograd of src/models/topkautoencoder.py:72-91, where the decoder input is
// the top-k activations SCATTERED into a dense [M x n] row and every gradient GEMM runs densely on 99.7 % zeros).
//
// Only the selected (row m, latent j) pairs carry gradient.  With g = d loss / d (decoder output) of a decode pass
// (main: de, AuxK: de_hat, multi-TopK: dm) and a = its selected activation:
//     da        = bf16( g[m] . W_dec[j] )                (one rounding, like the dense GEMM output it replaces)
//     dW_dec[j] += a  * g[m]           dW_enc[j] += da * sae_in[m]          d b_enc[j] += da        (ReLU gate: a > 0)
// i.e. per LATENT a weighted sum of gathered rows.  The pairs are first sorted by latent (a CSC view of the selection)
// and one wave then owns a latent (or a chunk of a long list): W_dec[j] and both accumulator rows stay in registers,
// each entry costs two gathered rows (g[m], sae_in[m]; 3 KB at d = 768, served by L2 / Infinity Cache: both arrays
// are 100 MB) instead of three dense [M x n] GEMM passes + a dense [M x n] dpre.
//
// Determinism: the CSC order inside a latent's list is fixed by construction (row block, then row, then pass -- never by
// which thread got there first), every work item writes its partial sums to its own slot, and a last kernel adds the
// slots of a latent in order: run-to-run bitwise identical, no float atomics.
//
// CSC build (stable counting sort by latent):
//   1. csc_count:  one WAVE per block of CSC_ROWS rows counts its entries per latent in LDS (u16 counters; the dictionary in
//                  segments of 32 768 latents, one wave per (row block, segment): any n_p);
//   2. csc_scan_blocks: per latent, exclusive prefix over the row blocks (+ the latent's total);
//   3. csc_scan_latents: exclusive prefix over the latents (list starts) and over the work-item counts;
//   4. csc_fill:   a wave per row block (and per segment of 16 384 latents) walks its rows IN ORDER; its LDS counters start at the
//                  (block, latent) runs' first slots and an LDS add-with-return hands every entry its slot -- LDS operations of
//                  one wave execute in program order, and the indices of one (row, pass) are distinct, so the order does not
//                  depend on timing.
#pragma once
#include "topk_kernels.h"

constexpr int CSC_SUB = 64;           // rows a wave holds in registers at a time
#ifndef CSC_ROWS_N
#define CSC_ROWS_N 128
#endif
constexpr int CSC_ROWS = CSC_ROWS_N;  // rows per counting block (one wave, CSC_ROWS / CSC_SUB rounds; round 3: 64 -- twice the count / offset
                                      // tables, 150 MB of them at C3)
static_assert(CSC_ROWS % CSC_SUB == 0 && CSC_ROWS * 3 < 65536, "u16 counters per (block, latent)");
constexpr int CSC_MAX_NP = 32768;     // latents per counting segment (u16 LDS counters: 64 KiB per wave)
constexpr int CSC_CHUNK = 256;        // entries per work item of the gradient kernel

struct SparsePasses {                 // autograd's execution order: multi-TopK, AuxK, main
  const int* idx[3];                  // [M_p][kcap] (null = pass absent)
  const bf16_t* vals[3];              // selected activations of the pass, compact [M_p][kcap]
  const bf16_t* g[3];                 // d (decoder output) [M_p][d_p]
  int kcap[3];
  int gated[3];                       // 1: only while tk[0] > 0 (AuxK)
};

struct CscEntry {                     // 8 bytes
  unsigned int row_pass;              // row | pass << 30
  float act;
};

// ---- 1. counts[b][j] (u16) --------------------------------------------------------------------------------------------
// (grid.y = dictionary segments of CSC_MAX_NP latents: a wave counts only the entries of its segment, so any n_p fits the
// 64 KiB of u16 counters; the rows' index lists are read once per segment -- 16 MB per pass against the gathers' 13 GB)
__global__ __launch_bounds__(64) void csc_count_kernel(SparsePasses ps, const int* __restrict__ tk, int64_t M, int n_p,
                                                       unsigned short* __restrict__ counts) {
  extern __shared__ unsigned int ctr32[];                     // segment / 2 words = one u16 counter per latent of the segment
  const int lane = threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * CSC_ROWS;
  const int seg0 = blockIdx.y * CSC_MAX_NP, seg1 = min(seg0 + CSC_MAX_NP, n_p), segn = seg1 - seg0;
  for (int i = lane; i < segn / 8; i += 64) reinterpret_cast<u32x4*>(ctr32)[i] = u32x4{0u, 0u, 0u, 0u};     // (segn: a multiple of 128)
  __syncthreads();
  for (int pass = 0; pass < 3; ++pass) {
    if (!ps.idx[pass] || (ps.gated[pass] && tk[0] <= 0)) continue;
    const int kcap = ps.kcap[pass];
    for (int q0 = 0; q0 < kcap; q0 += 64)
    for (int64_t rs = r0; rs < r0 + CSC_ROWS; rs += CSC_SUB) {
      // CSC_SUB rows of this 64-wide index chunk: every load is issued before the first LDS add (one memory
      // latency per chunk instead of one per row) -- from CLAMPED addresses, the lanes / rows without an entry marked by an OR into
      // the index: with `ok ? load : -1` hipcc predicates every load and waits for it inside its block (csc_fill_kernel)
      int v[CSC_SUB];
      const int q = q0 + lane;
      const int qc = q < kcap ? q : kcap - 1;
      const int qbad = q < kcap ? 0 : (int)0x80000000;
#pragma unroll
      for (int r = 0; r < CSC_SUB; ++r) {
        const int64_t rr = rs + r < M ? rs + r : M - 1;
        v[r] = ps.idx[pass][rr * kcap + qc] | qbad | (rs + r < M ? 0 : (int)0x80000000);
      }
#pragma unroll
      for (int r = 0; r < CSC_SUB; ++r)
        if (v[r] >= seg0 && v[r] < seg1)
          atomicAdd(&ctr32[(v[r] - seg0) >> 1], (v[r] & 1) ? 0x10000u : 1u);   // (a block holds < 65536 entries of a latent)
    }
  }
  __syncthreads();
  u32x4* out = reinterpret_cast<u32x4*>(counts + (int64_t)blockIdx.x * n_p + seg0);
  for (int i = lane; i < segn / 8; i += 64) out[i] = reinterpret_cast<const u32x4*>(ctr32)[i];
}

// ---- 2. per latent: exclusive prefix over the row blocks, total -----------------------------------------------------------
// One workgroup = 64 latents (lane) x 16 waves; wave w owns the row blocks [w nb/16, (w+1) nb/16): partial sums, an exclusive
// prefix over the 16 waves through LDS, then the same walk again writing the offsets (a single thread per latent walking
// all 1024 blocks was a 236 us latency chain).
__global__ __launch_bounds__(1024) void csc_scan_blocks_kernel(const unsigned short* __restrict__ counts, int nblocks, int n_p,
                                                               unsigned int* __restrict__ block_off, unsigned int* __restrict__ total) {
  __shared__ unsigned int part[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  const int b0 = (int)((int64_t)nblocks * w / 16), b1 = (int)((int64_t)nblocks * (w + 1) / 16);
  unsigned int run = 0;
  if (j < n_p)
    for (int b = b0; b < b1; ++b) run += counts[(int64_t)b * n_p + j];
  part[w][lane] = run;
  __syncthreads();
  unsigned int base = 0, tot = 0;
  for (int ww = 0; ww < 16; ++ww) {
    const unsigned int v = part[ww][lane];
    base += ww < w ? v : 0u;
    tot += v;
  }
  if (j >= n_p) return;
  run = base;
  for (int b = b0; b < b1; ++b) {
    block_off[(int64_t)b * n_p + j] = run;
    run += counts[(int64_t)b * n_p + j];
  }
  if (w == 0) total[j] = tot;
}

// ---- 3. one block: list starts and work-item starts (exclusive prefixes over the latents) ---------------------------------
//   start[j] (j <= n_p), item_start[j] (j <= n_p) with ceil(total / CSC_CHUNK) items per latent (none for an empty list)
__global__ __launch_bounds__(1024) void csc_scan_latents_kernel(const unsigned int* __restrict__ total, int n_p,
                                                                unsigned int* __restrict__ start, unsigned int* __restrict__ item_start) {
  __shared__ unsigned int wsum[2][16];
  __shared__ unsigned int carry[2];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t == 0) carry[0] = carry[1] = 0;
  __syncthreads();
  for (int base = 0; base < n_p; base += 1024) {
    const int j = base + t;
    const unsigned int c = j < n_p ? total[j] : 0u;
    unsigned int v0 = c, v1 = (c + CSC_CHUNK - 1) / CSC_CHUNK;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned int a = __shfl_up(v0, o, 64), b = __shfl_up(v1, o, 64);
      if (lane >= o) { v0 += a; v1 += b; }
    }
    if (lane == 63) { wsum[0][w] = v0; wsum[1][w] = v1; }
    __syncthreads();
    unsigned int p0 = carry[0], p1 = carry[1];
    for (int ww = 0; ww < w; ++ww) { p0 += wsum[0][ww]; p1 += wsum[1][ww]; }
    if (j < n_p) {
      start[j] = p0 + v0 - c;
      item_start[j] = p1 + v1 - (c + CSC_CHUNK - 1) / CSC_CHUNK;
    }
    __syncthreads();
    if (t == 1023) { carry[0] = p0 + v0; carry[1] = p1 + v1; }
    __syncthreads();
  }
  if (t == 0) { start[n_p] = carry[0]; item_start[n_p] = carry[1]; }
}

// ---- 3b. item -> latent table (one thread per latent writes its chunks' slots) --------------------------------------------------
// ... and the list of the latents with no or several work items (multi[0] = their number, multi[4 ...] = the latents, in any order:
// sparse_combine_kernel treats them independently), zeroed by the host before this kernel.
__global__ __launch_bounds__(256) void csc_items_kernel(const unsigned int* __restrict__ item_start, int n_p,
                                                        unsigned int* __restrict__ item_latent, unsigned int* __restrict__ multi) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n_p) return;
  const unsigned int i0 = item_start[j], i1 = item_start[j + 1];
  for (unsigned int i = i0; i < i1; ++i) item_latent[i] = (unsigned int)j;
  if (i1 - i0 != 1) multi[4 + atomicAdd(&multi[0], 1u)] = (unsigned int)j;
}

#ifdef CSCF_STAMP
// diagnostic build (tools/build_variant.sh cscfstamp -DCSCF_STAMP; bench.py --dbg 69): s_memtime at the phase boundaries of csc_fill,
// one row of 8 per workgroup: cycles since the workgroup's start
__device__ unsigned long long cscf_stamp_buf[4096 * 8];
#define CSCF_MARK() do { if (lane == 0 && nst_ < 8) stamp_[nst_++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CSCF_MARK() do {} while (0)
#endif
// ---- 4. fill: entries[start[j] + block_off[b][j] + rank] ----------------------------------------------------------------------
// Round 4: the LDS counters of a (row block, dictionary segment) are 32-bit POSITIONS, initialised to start[j] + block_off[b][j] by a
// coalesced stream of the two tables (16-byte loads) -- the LDS add-with-return then IS the entry's slot.  Before, u16 rank counters
// started at zero and every entry gathered start[j] and block_off[b][j] (two scattered 4-byte loads per entry: a third of the
// kernel by its stamps).  Segments of CSC_FILL_NP latents: 64 KiB of counters per wave.
constexpr int CSC_FILL_NP = 16384;
__global__ __launch_bounds__(64) void csc_fill_kernel(SparsePasses ps, const int* __restrict__ tk, int64_t M, int n_p,
                                                      const unsigned int* __restrict__ block_off, const unsigned int* __restrict__ start,
                                                      CscEntry* __restrict__ entries) {
  extern __shared__ unsigned int ctr32[];
  const int lane = threadIdx.x;
  // Row block of this workgroup.  A latent's entries of CONSECUTIVE row blocks are neighbours in the list (8 bytes each: sixteen
  // blocks share a 128-byte line), and workgroups go to the XCDs round robin: with block = blockIdx.x the neighbours were written
  // through eight different L2s, every store a partial line on its way to HBM.  XCD x now takes the blocks [x nb/8, (x+1) nb/8):
  // neighbours meet in ONE L2 within microseconds and leave as whole lines.
#ifdef CSCF_NO_XCD_REMAP
  const int blk = blockIdx.x;
#else
  const int nb8 = gridDim.x >> 3;
  const int blk = (int)blockIdx.x < 8 * nb8 ? (int)(blockIdx.x & 7) * nb8 + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
#endif
  const int64_t r0 = (int64_t)blk * CSC_ROWS;
  const int seg0 = blockIdx.y * CSC_FILL_NP, seg1 = min(seg0 + CSC_FILL_NP, n_p), segn = seg1 - seg0;    // (n_p: a multiple of 128)
#ifdef CSCF_STAMP
  unsigned long long stamp_[8];
  int nst_ = 0;
#endif
  CSCF_MARK();                                    // [0]
  {
    const u32x4* s4 = reinterpret_cast<const u32x4*>(start + seg0);
    const u32x4* b4 = reinterpret_cast<const u32x4*>(block_off + (int64_t)blk * n_p + seg0);
    u32x4* c4 = reinterpret_cast<u32x4*>(ctr32);
#pragma unroll 8
    for (int i = lane; i < segn / 4; i += 64) c4[i] = s4[i] + b4[i];
  }
  __syncthreads();
  CSCF_MARK();                                    // [1] positions initialised
  for (int pass = 0; pass < 3; ++pass) {
    if (!ps.idx[pass] || (ps.gated[pass] && tk[0] <= 0)) continue;
    const int kcap = ps.kcap[pass];
    for (int q0 = 0; q0 < kcap; q0 += 64)       // order inside (block, latent): pass, index chunk, row -- fixed, never timing
    for (int64_t rs = r0; rs < r0 + CSC_ROWS; rs += CSC_SUB) {
      // Phases with all of their memory operations in flight together (with the load, the LDS add and the store of a row in one loop
      // body the wave made 64 dependent round trips per chunk), and loads from CLAMPED addresses + a select instead of predicated
      // loads (a load in its own exec-masked block is followed by its use inside that block, i.e. by a wait).  The activations are
      // loaded after the LDS phase, one register each: next to the indices they made a fourth 64-entry array, which went to AGPRs
      // through a copy that WAITED for every 2-byte load in turn (84 k of the workgroup's 211 k cycles, -DCSCF_STAMP).
      int v[CSC_SUB];
      const int q = q0 + lane;
      const int qc = q < kcap ? q : kcap - 1;
      const int qbad = q < kcap ? 0 : (int)0x80000000;          // (ORed into the index: negative = no entry; a select on the loaded
#pragma unroll                                                  //  value let hipcc turn the load back into a predicated one)
      for (int r = 0; r < CSC_SUB; ++r) {        // (1) the chunk's indices
        const int64_t rr = rs + r < M ? rs + r : M - 1;
        v[r] = ps.idx[pass][rr * kcap + qc] | qbad | (rs + r < M ? 0 : (int)0x80000000);
      }
#ifdef CSCF_STAMP
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      CSCF_MARK();                                // [2] indices loaded
      unsigned int pos[CSC_SUB];
#pragma unroll
      for (int r = 0; r < CSC_SUB; ++r) {        // (2) the slot: LDS add with return; one wave, program order, distinct j within a
        const int j = v[r];                       //     (row, pass) -- the order inside (block, latent) never depends on timing
        pos[r] = 0xFFFFFFFFu;
        if (j >= seg0 && j < seg1) pos[r] = atomicAdd(&ctr32[j - seg0], 1u);
      }
#ifdef CSCF_STAMP
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
      CSCF_MARK();                                // [3] slots
      __builtin_amdgcn_sched_barrier(0);          // (the activation loads below must not move up into the phases above)
      unsigned int a[CSC_SUB];
#pragma unroll
      for (int r = 0; r < CSC_SUB; ++r) {        // (3) the activations, one register each, all in flight
        const int64_t rr = rs + r < M ? rs + r : M - 1;
        a[r] = (unsigned int)reinterpret_cast<const unsigned short*>(ps.vals[pass])[rr * kcap + qc];
      }
      __builtin_amdgcn_sched_barrier(0);
#ifdef CSCF_STAMP
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      CSCF_MARK();                                // [4] activations loaded
#pragma unroll
      for (int r = 0; r < CSC_SUB; ++r) {        // (4) the entries
        if (pos[r] == 0xFFFFFFFFu) continue;
        CscEntry e;
        e.row_pass = (unsigned int)(rs + r) | ((unsigned int)pass << 30);
        e.act = __uint_as_float(a[r] << 16);
        entries[pos[r]] = e;
      }
      CSCF_MARK();                                // [5] stores issued
#ifdef CSCF_STAMP
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      CSCF_MARK();                                // [6] stores acknowledged
    }
  }
#ifdef CSCF_STAMP
  if (lane == 0 && blockIdx.x < 4096 && blockIdx.y == 0)
    for (int i = 0; i < 8; ++i) cscf_stamp_buf[blockIdx.x * 8 + i] = i < nst_ ? stamp_[i] - stamp_[0] : 0ull;
#endif
}

#ifndef SB_E
#define SB_E 4          // entries in flight per wave and trip of sparse_bwd_kernel (even)
#endif
// ---- 5. gradient work items -----------------------------------------------------------------------------------------------
// item -> (latent j = item_latent[item], chunk item - item_start[j]); partial sums to part[item][2][d_p] and pbe[item].
#ifndef SB_WAVES
#define SB_WAVES 4      // waves (work items) per workgroup of sparse_bwd_kernel
#endif
template <int NPAIR>
__global__ __launch_bounds__(64 * SB_WAVES) void sparse_bwd_kernel(SparsePasses ps, const bf16_t* __restrict__ xs, const bf16_t* __restrict__ Wd,
                                                         const CscEntry* __restrict__ entries, const unsigned int* __restrict__ start,
                                                         const unsigned int* __restrict__ item_start,
                                                         const unsigned int* __restrict__ item_latent, int n_p,
                                                         float* __restrict__ part, float* __restrict__ pbe,
                                                         float* __restrict__ gWd, float* __restrict__ gWe, float* __restrict__ gbe,
                                                         float* __restrict__ dbe_exact) {
  constexpr int d_p = 128 * NPAIR, CPL = 2 * NPAIR;           // columns per lane
  const int lane = threadIdx.x & 63;
  const unsigned int nitems = item_start[n_p];
  const unsigned int item = blockIdx.x * SB_WAVES + (threadIdx.x >> 6);
  if (item >= nitems) return;                                 // wave-uniform
  const int j = (int)item_latent[item];
  const unsigned int e0 = start[j] + (item - item_start[j]) * CSC_CHUNK;
  const unsigned int e1 = min(e0 + CSC_CHUNK, start[j + 1]);
  const int c0 = lane * CPL;
  float wd[CPL], accd[CPL], acce[CPL];
  {
    const unsigned* wp = reinterpret_cast<const unsigned*>(Wd + (int64_t)j * d_p + c0);
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) {
      const unsigned u = wp[p];
      wd[2 * p] = __uint_as_float(u << 16);
      wd[2 * p + 1] = __uint_as_float(u & 0xFFFF0000u);
      accd[2 * p] = accd[2 * p + 1] = acce[2 * p] = acce[2 * p + 1] = 0.f;
    }
  }
  float dbe = 0.f;
  // SB_E entries in flight per trip; the NEXT trip's (row, activation) words are requested one trip ahead, so that a trip's
  // gathers do not start behind another dependent load (the kernel waits on memory 83 % of its wave time)
  constexpr int E = SB_E;
  CscEntry nx[E];
#pragma unroll
  for (int u = 0; u < E; ++u) nx[u] = entries[e0 + u < e1 ? e0 + u : (e0 < e1 ? e1 - 1 : 0)];
  for (unsigned int e = e0; e < e1; e += E) {
    CscEntry en[E];
    float act[E];
    unsigned ug[E][NPAIR], ux[E][NPAIR];
#pragma unroll
    for (int u = 0; u < E; ++u) {
      en[u] = nx[u];
      act[u] = e + u < e1 ? en[u].act : 0.f;            // entries past the end repeat the last one with weight 0
    }
    if (e + E < e1) {
#pragma unroll
      for (int u = 0; u < E; ++u) nx[u] = entries[e + E + u < e1 ? e + E + u : e1 - 1];
    }
#pragma unroll
    for (int u = 0; u < E; ++u) {
      const int64_t m = en[u].row_pass & 0x3FFFFFFFu;
      const unsigned* gp = reinterpret_cast<const unsigned*>(ps.g[en[u].row_pass >> 30] + m * d_p + c0);
#pragma unroll
      for (int p = 0; p < NPAIR; ++p) ug[u][p] = gp[p];
    }
#pragma unroll
    for (int u = 0; u < E; ++u) {
      const int64_t m = en[u].row_pass & 0x3FFFFFFFu;
      const unsigned* xp = reinterpret_cast<const unsigned*>(xs + m * d_p + c0);
#pragma unroll
      for (int p = 0; p < NPAIR; ++p) ux[u][p] = xp[p];
    }
    float da[E];
#pragma unroll
    for (int u = 0; u < E; ++u) {
      float sdot = 0.f;
#pragma unroll
      for (int p = 0; p < NPAIR; ++p)
        sdot += __uint_as_float(ug[u][p] << 16) * wd[2 * p] + __uint_as_float(ug[u][p] & 0xFFFF0000u) * wd[2 * p + 1];
      sdot = wave_sum(sdot);
      da[u] = act[u] > 0.f ? bf16_round(sdot) : 0.f;      // ReLU gate
    }
    // (the sums below keep the two-entries-per-trip association of the first version of this kernel: (a + b), then (c + d))
#pragma unroll
    for (int u = 0; u < E; u += 2) {
      dbe += da[u] + da[u + 1];
#pragma unroll
      for (int p = 0; p < NPAIR; ++p) {
        accd[2 * p] += act[u] * __uint_as_float(ug[u][p] << 16) + act[u + 1] * __uint_as_float(ug[u + 1][p] << 16);
        accd[2 * p + 1] += act[u] * __uint_as_float(ug[u][p] & 0xFFFF0000u) + act[u + 1] * __uint_as_float(ug[u + 1][p] & 0xFFFF0000u);
        acce[2 * p] += da[u] * __uint_as_float(ux[u][p] << 16) + da[u + 1] * __uint_as_float(ux[u + 1][p] << 16);
        acce[2 * p + 1] += da[u] * __uint_as_float(ux[u][p] & 0xFFFF0000u) + da[u + 1] * __uint_as_float(ux[u + 1][p] & 0xFFFF0000u);
      }
    }
  }
  if (item_start[j + 1] - item_start[j] == 1) {     // the latent's only work item: these ARE its gradient rows
    float* od = gWd + (int64_t)j * d_p + c0;
    float* oe = gWe + (int64_t)j * d_p + c0;
#pragma unroll
    for (int p = 0; p < CPL; ++p) {
      od[p] = accd[p];
      oe[p] = acce[p];
    }
    if (lane == 0) {
      dbe_exact[j] = dbe;
      gbe[j] = bf16_round(dbe);
    }
    return;
  }
  float* po = part + (int64_t)item * 2 * d_p + c0;
#pragma unroll
  for (int p = 0; p < CPL; ++p) {
    po[p] = accd[p];
    po[d_p + p] = acce[p];
  }
  if (lane == 0) pbe[item] = dbe;
}

// ---- 6. per latent with no or several work items: zero rows / its items' partial sums in order -> dW_dec[j], dW_enc[j],
//         d b_enc[j] (bf16-rounded) + exact copy
__global__ __launch_bounds__(256) void sparse_combine_kernel(const float* __restrict__ part, const float* __restrict__ pbe,
                                                             const unsigned int* __restrict__ item_start, int n_p, int d_p,
                                                             float* __restrict__ gWd, float* __restrict__ gWe, float* __restrict__ gbe,
                                                             float* __restrict__ dbe_exact, const unsigned int* __restrict__ multi) {
  // A fixed grid of workgroups walks the units (listed latent, 64 columns); the 4 waves take the items i0 + w, i0 + w + 4, ... and
  // their sums meet in LDS in wave order (fixed order -> deterministic); a latent that fires on most rows has hundreds of items.
  // (Round 3 launched one workgroup per (latent, 64 columns) of the WHOLE dictionary -- 295 k at C3, nearly all of which read two
  // words and left: the kernel's 98 us were mostly their dispatch.)
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int chunks = d_p >> 6;
  const unsigned int units = multi[0] * (unsigned int)chunks;
  for (unsigned int u = blockIdx.x; u < units; u += gridDim.x) {
  const int j = (int)multi[4 + u / chunks], cy = (int)(u % chunks);
  const unsigned int i0 = item_start[j], i1 = item_start[j + 1];
  const int c = cy * 64 + lane;
  float a = 0.f, b = 0.f;
#pragma unroll 8      // (eight trips' loads in flight, same order: a latent that fires on most rows has 64 trips per wave here)
  for (unsigned int i = i0 + w; i < i1; i += 4) {
    a += part[(int64_t)i * 2 * d_p + c];
    b += part[(int64_t)i * 2 * d_p + d_p + c];
  }
  red[0][w][lane] = a;
  red[1][w][lane] = b;
  __syncthreads();
  if (w == 0) {
    gWd[(int64_t)j * d_p + c] = (red[0][0][lane] + red[0][1][lane]) + (red[0][2][lane] + red[0][3][lane]);
    gWe[(int64_t)j * d_p + c] = (red[1][0][lane] + red[1][1][lane]) + (red[1][2][lane] + red[1][3][lane]);
  }
  if (cy == 0 && threadIdx.x == 0) {
    float s = 0.f;
    for (unsigned int i = i0; i < i1; ++i) s += pbe[i];
    dbe_exact[j] = s;
    gbe[j] = bf16_round(s);
  }
  __syncthreads();                                    // (red is reused by the next unit)
  }
}
