// Fused backward of the tied-weight L1 SAE for d_model (padded) == 384 on gfx950.
//
// Per (row r, dictionary column j):   dc = dx_hat[r,:] . W[:,j]
//                                     dpre = bf16(dc + 1/M) * [c > 0]      (the 1/M term rides in the fp32 accumulator: one
//                                                                           rounding, where CPU autocast rounds dc first)
// and the single tied weight gradient dW[:, j] += dx_hat[r,:]^T c[r,j] + x[r,:]^T dpre[r,j],  db[j] += dpre[r,j]
// (reference: autograd of src/models/l1autoencoder.py:69-95).
//
// One workgroup (4 waves, one per SIMD, whole 512-register file per wave) owns 128 dictionary columns
// and a contiguous range of activation rows; wave w owns columns 32w..32w+31:
//   - its slice of W^T (32 columns x K=384, 24 MFMA B-fragments) stays in registers for the whole kernel;
//   - its dW slab [384 x 32] fp32 stays in 192 accumulator registers (12 MFMA tiles) across all rows;
//   - per 32-row step the dx_hat / x / c tiles are staged by LDS-DMA (global_load_lds, double buffered; the
//     latent tile with the non-temporal policy: it is read exactly once);
//     dc is one 32x32 MFMA tile (24 MFMAs, A = dx_hat rows by ds_read_b128); its accumulator registers are
//     gated by c and turned into the bf16 B-operand of the x^T dpre product without leaving the register
//     file (accumulator-as-operand: rows of the 32x32 tile are the next product's K index);
//     the A-operands dx_hat^T and x^T are read from the SAME row-major LDS images with the hardware
//     transposing read ds_read_b64_tr_b16, in the permuted k order the accumulator layout dictates.
// dpre never exists in HBM; c is read exactly once.  Partial dW slabs / db vectors (one per row range)
// are summed by reduce_grads_kernel in a fixed order (deterministic).
#pragma once
#include "common.h"
#include "gemm.h"

constexpr int BF_D = 384;                 // padded d_model this kernel is specialised for
constexpr int BF_BM = 32;                 // rows per step
constexpr int BF_BN = 128;                // dictionary columns per workgroup
constexpr int BF_DXH_BYTES = BF_BM * BF_D * 2;   // 24576
constexpr int BF_C_BYTES = BF_BM * BF_BN * 2;    // 8192
constexpr int BF_STAGE_BYTES = 2 * BF_DXH_BYTES + BF_C_BYTES;   // 57344
constexpr int BF_LDS_BYTES = 2 * BF_STAGE_BYTES;                // 114688

// byte offset of 16-B chunk `ch` (0..15) of row `row` in a [rows][128 x bf16] image (256-B rows) that serves
// both ds_read_b128 row reads and ds_read_b64_tr_b16 transposed reads without bank conflicts
__device__ __forceinline__ int dual_off(int row, int ch) {
  return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
}

struct BwdFusedArgs {
  const bf16_t* dxh;   // [M_p][384]
  const bf16_t* xb;    // [M_p][384]
  const bf16_t* c;     // [M_p][n_p]
  const bf16_t* Wt;    // [n_p][384]
  const float* scal;   // scal[1] = alpha/count, scal[2] = 1/M
  int unscaled;        // 1: dxh holds 2 (x_hat - x)[keep] without the alpha/count factor (fused forward)
  // cnt_part != null (single GPU, fused forward): alpha/count and 1/M are taken HERE from the forward's per-workgroup
  // masked-entry counts (exact integers: any summation order gives the same value in every workgroup), so that the one-block
  // loss finalisation no longer sits between the forward and the backward (it rides in reduce_grads_kernel's last block)
  const float* cnt_part;
  int n_cnt;
  float alpha;
  int64_t M;
  int d;
  float* slab;         // [splits][384][n_p]
  float* db_part;      // [splits][n_p]
  int n_p;
  int ntiles;          // column tiles of this launch (n_p / 128, or one column range of them)
  int tile0;           // first column tile of the launch (column-range launches; 0 otherwise)
  int splits;
  int steps_total;     // M_p / 32
  // Balanced form (round 4; bal_m > 0): the work is cut into QUANTA -- (column tile, 1/32 of the rows) -- linearised tile-major,
  // and workgroup k takes quanta [k bal_m, (k + 1) bal_m): every CU gets the same number of rows whatever ntiles is (24 tiles
  // x 10 row ranges left 16 of 256 CUs idle at C2), at the price of workgroups that finish one tile's rows and go on with the
  // next tile's (a "segment" each: own W^T fragments, own slab piece).  A quantum is bal_q steps (ceil(steps_total / 32)), so
  // workgroups start at only 32 / gcd(bal_m, 32) distinct row phases and those of equal phase walk the same dx_hat / x rows in
  // lockstep; wg_map places them on one XCD.  Tile j's rows arrive as pieces k - (32 j) / bal_m of the workgroups k that touch
  // it: slab / db_part are indexed [piece][...] and reduce_grads_kernel adds bal_pieces(j, bal_m) of them.
  int bal_m, bal_q;
  const short* wg_map; // blockIdx.x -> k (null: identity)
  unsigned long long* clk;   // diagnostic (bench.py --dbg 66), normally null: [wg][4] = s_memtime / s_memrealtime around the loop
};

// transposed-read fragment (A or B operand of v_mfma_f32_32x32x16_bf16) from a dual-use image, 32 columns
// starting at col0 (multiple of 32) of a 128-column sub-tile, for k-step s (16 rows), in the PERMUTED k order
// that matches an accumulator tile used as the other operand: element j of lane half h <-> row
// 16 s + 8 (j >> 2) + 4 h + (j & 3).
__device__ __forceinline__ bf16x8 tr_frag_perm(const char* img, int col0, int s, int lane) {
  const int h = lane >> 5, g = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
  const int col = col0 + 16 * g + 4 * p;
  const int r0 = 16 * s + 4 * h + q, r1 = r0 + 8;
  const int o0 = dual_off(r0, col >> 3) + (col & 7) * 2;
  const int o1 = dual_off(r1, col >> 3) + (col & 7) * 2;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, img + o0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, img + o1));
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// Pair p of the next step's 7 DMA pairs is issued in gap BF_DMA_A * p + BF_DMA_B (macros so that tools/build_variant.sh
// can sweep them).  Same-box sweeps of the backward at C2: with the default cache policy on the latent (cycles per step)
// 3p+1 3360, 5p+1 3325, 2p+1 3565, p+1 3814 (back-to-back pieces cost more each), 8p+1 3860 and 3p+24 4044 (the last
// pieces land after the hand-over barrier), one wave per gap (staggered) 3710; with the non-temporal latent reads (ms)
// 4p+1 0.2871 (shipped), 4p+2 0.2874, 3p+2 0.2886, 3p+1 0.2895, 5p+1 0.2925, 2p+1 0.299, 6p+1 0.314, p+1 0.312, 8p+2 0.355.
#ifndef BF_DMA_A_
#define BF_DMA_A_ 4
#define BF_DMA_B_ 1
#endif
constexpr int BF_DMA_A = BF_DMA_A_, BF_DMA_B = BF_DMA_B_;
// Round 6: the step loop unrolled over the two LDS stages (the stage a step reads is a compile-time constant).  With `cur` a run-time
// variable every step computed its stage bases again -- by the disassembly 39 scalar instructions in ONE block at the loop head (two
// stage offsets by multiplication, three 64-bit row pointers from the step number, fourteen LDS destinations of the DMA pieces) and 26
// v_add_u32 (stage base + lane offset in front of the LDS reads whose immediate field cannot hold an offset into the second stage) --
// ~65 of the ~300 instructions a wave issues per step, on the ONE issue port a single wave per SIMD has, most of them with the matrix
// pipe idle.  Now: a second set of lane offsets for the second stage (18 registers), the row pointers advanced by their stride, the
// LDS destination formed by the `s_add_u32 m0, ...` that writes M0 anyway.  -DBF_STATIC_STAGES=0 = round 5's loop.
#ifndef BF_STATIC_STAGES
#define BF_STATIC_STAGES 1
#endif
#ifndef BF_GATE_AHEAD
#define BF_GATE_AHEAD 1      // the ReLU gate's compare one MFMA gap ahead of its select (see the gate below); 0 = both in one gap
#endif
#ifndef BF_DMA_SINGLE
#define BF_DMA_SINGLE 1      // round 6: the 14 LDS-DMA pieces of a step one per second gap (1, 3, ..., 27) instead of 7 pairs in gaps 1, 5, ..., 25: 2620 -> 2581 cycles per step
#endif
#ifndef BF_RING
#define BF_RING 12           // fragment ring (divides 72); the prefetch distance is BF_RING - 1 MFMAs
#endif

// two LDS-DMA pieces whose LDS destinations are base + l0 / base + l1: the addition IS the write of M0
__device__ __forceinline__ void glds16_x2_add(const void* sbase0, const void* sbase1, unsigned voff0, unsigned voff1, unsigned base,
                                              unsigned l0, unsigned l1) {
  asm volatile(
      "s_add_u32 m0, %4, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0\n\t"
      "s_add_u32 m0, %4, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1"
      :
      : "s"(sbase0), "s"(sbase1), "v"(voff0), "v"(voff1), "s"(base), "s"(l0), "s"(l1)
      : "memory", "m0", "scc");
}
// single pieces of the same (BF_DMA_SINGLE)
__device__ __forceinline__ void glds16_add(const void* sbase, unsigned voff, unsigned base, unsigned l0) {
  asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" : : "s"(sbase), "v"(voff), "s"(base), "s"(l0) : "memory", "m0", "scc");
}
__device__ __forceinline__ void glds16_add_nt(const void* sbase, unsigned voff, unsigned base, unsigned l0) {
  asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0 nt" : : "s"(sbase), "v"(voff), "s"(base), "s"(l0) : "memory", "m0", "scc");
}
__device__ __forceinline__ void glds16_x2_add_nt(const void* sbase0, const void* sbase1, unsigned voff0, unsigned voff1, unsigned base,
                                                 unsigned l0, unsigned l1) {
  asm volatile(
      "s_add_u32 m0, %4, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0 nt\n\t"
      "s_add_u32 m0, %4, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1 nt"
      :
      : "s"(sbase0), "s"(sbase1), "v"(voff0), "v"(voff1), "s"(base), "s"(l0), "s"(l1)
      : "memory", "m0", "scc");
}

__global__ __launch_bounds__(256, 1) void bwd_fused_d384_kernel(BwdFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  // the workgroup's segments: one (column tile, step range, slab piece) in the uniform form; in the balanced form one per column
  // tile its quanta touch (usually one, two for the workgroups that straddle a tile boundary)
  // (Round 5: the prologue's loads are issued TOGETHER -- the workgroup-number table entry and this thread's share of the forward's
  // count partials -- and waited for once.  As written before, the table entry, then each count partial in turn, were a load and an
  // `s_waitcnt vmcnt(0)` each: three memory latencies in a row before the first W^T fragment was requested.)
  int qa = 0, qe = 1, wgk = 0;
  unsigned long long clk_k0 = 0;
  if (a.clk) clk_k0 = __builtin_amdgcn_s_memtime();
  int wgk_raw = (int)blockIdx.x;
  if (a.bal_m > 0 && a.wg_map) wgk_raw = (int)a.wg_map[blockIdx.x];
  double cnt_m = 0;
  if (a.cnt_part) {
    for (int i0 = t; i0 < a.n_cnt; i0 += 1024) {           // four independent loads per trip (512 partials at C2: one trip, two live)
      float cv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) cv[u] = a.cnt_part[min(i0 + 256 * u, a.n_cnt - 1)];     // (clamped, not predicated: no branch, no wait between them)
#pragma unroll
      for (int u = 0; u < 4; ++u) cnt_m += i0 + 256 * u < a.n_cnt ? (double)cv[u] : 0.0;  // the same additions in the same order as one partial per trip
    }
  }
  if (a.bal_m > 0) {
    wgk = __builtin_amdgcn_readfirstlane(wgk_raw);
    qa = wgk * a.bal_m;
    qe = qa + a.bal_m < 32 * a.ntiles ? qa + a.bal_m : 32 * a.ntiles;
  }
  int n0 = 0, nw = 0;                     // first dictionary column of the workgroup / of this wave (per segment)
  // With an unscaled dx_hat everything is computed in units of 1/scale: dpre' = dc' + (1/M)/scale, and the
  // epilogue multiplies the dW slab and db by scale = alpha/count.
  float scal1, scal2;
  if (a.cnt_part) {
    double* redc = reinterpret_cast<double*>(smem + BF_DXH_BYTES);        // the second stage's dx_hat image (either layout): idle until the loop's first hand-over
    double m = wave_sum_d(cnt_m);
    if (lane == 0) redc[w] = m;
    __syncthreads();
    const double count = (double)a.M * a.d - ((redc[0] + redc[1]) + (redc[2] + redc[3]));
    scal1 = a.alpha / (float)count;          // the same float operations as finalize_losses_kernel
    scal2 = 1.0f / (float)a.M;
    __syncthreads();
  } else {
    scal1 = a.scal[1];
    scal2 = a.scal[2];
  }
  const float scale = a.unscaled ? scal1 : 1.0f;
  const float inv_m = a.unscaled ? scal2 / scal1 : scal2;

  // ---- staging by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B land contiguously at a wave-uniform LDS
  // address).  One instruction fills 4 rows x 256 B of a [32][256 B] sub-tile image; the dual-use swizzle is
  // applied to the per-lane SOURCE chunk (linear destination + swizzled source + swizzled reads).
  // Per step: dx_hat 24 + x 24 + c 8 instructions, 6 + 6 + 2 per wave, issued as 7 two-piece statements.
  const int srow = lane >> 4;                         // row within the 4-row group
  const int pc = lane & 15;                           // physical 16-B chunk the lane's data lands in
  unsigned voff_d[6], voff_c[2];                      // per-lane source byte offsets relative to the step's first row
  unsigned loff_d[6], loff_c[2];                      // wave-uniform LDS byte offsets inside a stage
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int inst = w + 4 * i, sub = inst >> 3, row = 4 * (inst & 7) + srow;
    const int ch = pc ^ (((row & 3) << 2) | ((row >> 2) & 3));
    voff_d[i] = (unsigned)(row * (BF_D * 2) + (sub * 16 + ch) * 16);
    loff_d[i] = (unsigned)__builtin_amdgcn_readfirstlane(sub * 8192 + (inst & 7) * 1024);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int inst = w + 4 * i, row = 4 * inst + srow;
    const int ch = pc ^ (((row & 3) << 2) | ((row >> 2) & 3));
    voff_c[i] = (unsigned)(row * (a.n_p * 2) + ch * 16);
    loff_c[i] = (unsigned)__builtin_amdgcn_readfirstlane(2 * BF_DXH_BYTES + inst * 1024);
  }
  typedef __attribute__((address_space(3))) char* lptr_t;
  const unsigned smem_base = (unsigned)(uintptr_t)(lptr_t)smem;
#if BF_STATIC_STAGES
  // LDS layout of this form: [dx_hat stage 0 | dx_hat stage 1 | x stage 0 | x stage 1 | c stage 0 | c stage 1] (24 + 24 + 24 + 24 + 8 + 8
  // KiB) instead of two contiguous stages: the 16-bit immediate offset of an LDS read then reaches BOTH stages of an image family from
  // one set of lane offsets (row reads and transposed reads of dx_hat: 0 ... 48 KiB; x: 48 ... 96 KiB from a second set of eight; c: from
  // its own two) -- two whole sets for a contiguous second stage did not fit the 256 vector registers (596 bytes of scratch).
  // The same pair from RUNNING row pointers (pd / px: dx_hat and x rows of the step being fetched, pcl: its latent rows at this
  // workgroup's columns) into stage `stage`: the LDS destination is formed by the s_add that writes M0.
  unsigned loff_x[6], loff_cs[2];
#pragma unroll
  for (int i = 0; i < 6; ++i) loff_x[i] = (unsigned)__builtin_amdgcn_readfirstlane((int)(loff_d[i] + 2 * BF_DXH_BYTES));
#pragma unroll
  for (int i = 0; i < 2; ++i) loff_cs[i] = (unsigned)__builtin_amdgcn_readfirstlane((w + 4 * i) * 1024);
  const unsigned stage_base_d[2] = {smem_base, smem_base + BF_DXH_BYTES};
  const unsigned stage_base_c[2] = {smem_base + 4 * BF_DXH_BYTES, smem_base + 4 * BF_DXH_BYTES + BF_C_BYTES};
  auto dma_one_at = [&](int k, const bf16_t* pd, const bf16_t* px, const bf16_t* pcl, int stage) {      // piece k of 14: dx_hat / x alternating, then c
    if (k < 12) {
      if ((k & 1) == 0) glds16_add(pd, voff_d[k >> 1], stage_base_d[stage], loff_d[k >> 1]);
      else glds16_add(px, voff_d[k >> 1], stage_base_d[stage], loff_x[k >> 1]);
    } else {
      glds16_add_nt(pcl, voff_c[k - 12], stage_base_c[stage], loff_cs[k - 12]);
    }
  };
  auto dma_pair_at = [&](int p, const bf16_t* pd, const bf16_t* px, const bf16_t* pcl, int stage) {
    if (p < 6) glds16_x2_add(pd, px, voff_d[p], voff_d[p], stage_base_d[stage], loff_d[p], loff_x[p]);
    else glds16_x2_add_nt(pcl, pcl, voff_c[0], voff_c[1], stage_base_c[stage], loff_cs[0], loff_cs[1]);
  };
#endif
  // piece pair p (0..6) of the step whose first row is row0, into stage `stage`
  auto dma_pair = [&](int p, int64_t row0, int stage) {
    const unsigned buf = smem_base + stage * BF_STAGE_BYTES;
    if (p < 6) {
      glds16_x2(a.dxh + row0 * BF_D, a.xb + row0 * BF_D, voff_d[p], voff_d[p], buf + loff_d[p],
                buf + BF_DXH_BYTES + loff_d[p]);
    } else {
      const bf16_t* gc = a.c + row0 * a.n_p + n0;
      // the latent is read exactly once: non-temporal policy, so that it does not displace the dx_hat / x tiles the
      // 24 column-tile workgroups of a row range share in L2 (same-box A/B: backward -4 %, -9...15 % together with the
      // forward's non-temporal latent stores)
      glds16_x2_nt(gc, gc, voff_c[0], voff_c[1], buf + loff_c[0], buf + loff_c[1]);
    }
  };

  // ---- loop-invariant per-lane LDS read offsets (everything else folds into instruction immediates)
  //   row read of chunk 2*(kk&7)+h of row (lane&31):         roff[kk & 7]
  //   transposed read, column group dt&3, first/second half: toff[2 * (dt & 3) + second]   (k-step adds 4096)
  const int arow = lane & 31, ah = lane >> 5;
  int roff[8], toff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) roff[i] = dual_off(arow, 2 * i + ah);
  {
    const int g = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int col = 32 * (i >> 1) + 16 * g + 4 * p;
      const int r = 4 * ah + q + 8 * (i & 1);
      toff[i] = dual_off(r, col >> 3) + (col & 7) * 2;
    }
  }
  const int coff0 = [&] {   // c fragments: this wave's 32 columns of the [32][128] c image
    const int g = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
    const int col = 32 * w + 16 * g + 4 * p;
    return dual_off(4 * ah + q, col >> 3) + (col & 7) * 2;
  }();
  const int coff1 = [&] {
    const int g = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
    const int col = 32 * w + 16 * g + 4 * p;
    return dual_off(4 * ah + q + 8, col >> 3) + (col & 7) * 2;
  }();
#if BF_STATIC_STAGES
  // lane offsets of the x images (transposed reads) and of the c images in the layout above; opaque, or they are folded back into
  // `toff + constant` and rebuilt by a v_add_u32 in front of every read
  int toffX[8], coffC0 = coff0 + 4 * BF_DXH_BYTES, coffC1 = coff1 + 4 * BF_DXH_BYTES;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    toffX[i] = toff[i] + 2 * BF_DXH_BYTES;
    asm volatile("" : "+v"(toffX[i]));
  }
  asm volatile("" : "+v"(coffC0), "+v"(coffC1));
#endif
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  auto tr_pair = [&](const char* p0, const char* p1) -> bf16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, p1));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  };

  // The step loop is rotated: the barrier that hands over the next stage sits DIST MFMAs before the end of a step,
  // and the gaps after it already request the first fragments (and the c fragments) of the NEXT step from the other
  // stage, so no step starts with an exposed LDS round trip.  RING divides 72, which keeps every ring slot static.
  constexpr int RING = BF_RING, DIST = RING - 1;
  static_assert(72 % RING == 0, "ring slots must be static across steps");
  auto load_frag = [&](const char* img_d, int i) -> bf16x8 {
    // A-operand fragment of MFMA i (0..71) of the step whose stage image starts at img_d:
    //   i in [ 0,24): dc += dx_hat(rows; ds_read_b128)      . W^T fragment i (registers)
    //   i in [24,48): dW[dt] += dx_hat^T (transposed reads) . c fragment s      (dt = (i-24)/2, s = i&1)
    //   i in [48,72): dW[dt] += x^T (transposed reads)      . dpre fragment s   (dpre made from dc in gaps 30..45)
    if (i < 24) {
      return *reinterpret_cast<const bf16x8*>(img_d + (i >> 3) * 8192 + roff[i & 7]);
    } else {
      const char* img = i < 48 ? img_d : img_d + BF_DXH_BYTES;
      const int tt = i < 48 ? i - 24 : i - 48, dt = tt >> 1, sk = tt & 1;
      const char* b = img + (dt >> 2) * 8192 + sk * 4096;
      return tr_pair(b + toff[2 * (dt & 3)], b + toff[2 * (dt & 3) + 1]);
    }
  };
#if BF_STATIC_STAGES
  auto load_frag_s = [&](auto stage_tag, int i) -> bf16x8 {      // load_frag of stage ST (a constant) in the interleaved layout
    constexpr int ST = decltype(stage_tag)::value;
    if (i < 24) {
      return *reinterpret_cast<const bf16x8*>(smem + ST * BF_DXH_BYTES + (i >> 3) * 8192 + roff[i & 7]);
    } else {
      const int tt = i < 48 ? i - 24 : i - 48, dt = tt >> 1, sk = tt & 1;
      const char* b = smem + ST * BF_DXH_BYTES + (dt >> 2) * 8192 + sk * 4096;
      return i < 48 ? tr_pair(b + toff[2 * (dt & 3)], b + toff[2 * (dt & 3) + 1]) : tr_pair(b + toffX[2 * (dt & 3)], b + toffX[2 * (dt & 3) + 1]);
    }
  };
  auto load_c_s = [&](auto stage_tag, int half) -> bf16x8 {
    constexpr int ST = decltype(stage_tag)::value;
    const char* b = smem + ST * BF_C_BYTES + half * 4096;
    return tr_pair(b + coffC0, b + coffC1);
  };
#endif
  // dc starts at (1/M)/scale: the L1 term sign(c)/M of d loss / d c, so the gate needs no add.  The dc chain is issued
  // as VGPR-form MFMAs (inline asm): the gate reads dc with plain VALU, and hipcc no longer parks a dW accumulator in
  // VGPRs to make room for it (that cost 48 v_accvgpr moves per step).
  f32x16 cinit;
#pragma unroll
  for (int r = 0; r < 16; ++r) cinit[r] = inv_m;

  unsigned long long clk_t0 = 0, clk_r0 = 0, clk_loop = 0, clk_real = 0, clk_steps = 0;
#pragma unroll 1
  for (bool first_seg = true; qa < qe; first_seg = false) {
  // ---- this segment: column tile, step range, slab piece
  bf16x8 wfrag[24];
  f32x16 acc[12];
  float db_acc = 0.f;
  bf16x8 ring[RING], cf[2], pf[2];
  int ntile, step_begin, step_end, split;
  if (a.bal_m > 0) {
    const int j = qa >> 5, qb = qe < 32 * (j + 1) ? qe : 32 * (j + 1);
    ntile = a.tile0 + j;
    step_begin = (qa - 32 * j) * a.bal_q;
    step_end = (qb - 32 * j) * a.bal_q;
    if (step_begin > a.steps_total) step_begin = a.steps_total;
    if (step_end > a.steps_total) step_end = a.steps_total;
    split = wgk - bal_first_wg(j, a.bal_m);
    qa = qb;
  } else {
    const int nblk = a.ntiles * a.splits;
    const int id = xcd_remap(blockIdx.x, nblk);
    split = id / a.ntiles;
    ntile = a.tile0 + id - split * a.ntiles;
    step_begin = (int)((int64_t)a.steps_total * split / a.splits);
    step_end = (int)((int64_t)a.steps_total * (split + 1) / a.splits);
    qa = qe;
  }
  ntile = __builtin_amdgcn_readfirstlane(ntile); step_begin = __builtin_amdgcn_readfirstlane(step_begin);
  step_end = __builtin_amdgcn_readfirstlane(step_end); split = __builtin_amdgcn_readfirstlane(split);
  n0 = ntile * BF_BN;
  nw = n0 + 32 * w;
  if (!first_seg) {      // the previous segment's last (redundant) DMA pieces have landed and every wave has left its loop
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
  }
  // W^T fragments of this wave: B[k = d][col = n] -> lane (n = lane & 31, h) holds Wt[nw + n][16 kk + 8 h ..+8]
  {
    const bf16_t* wp = a.Wt + (int64_t)(nw + (lane & 31)) * BF_D + 8 * (lane >> 5);
#pragma unroll
    for (int kk = 0; kk < 24; ++kk) wfrag[kk] = *reinterpret_cast<const bf16x8*>(wp + 16 * kk);
  }
#pragma unroll
  for (int i = 0; i < 12; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  db_acc = 0.f;
  if (step_begin < step_end) {
#if BF_STATIC_STAGES
    const int64_t r0 = (int64_t)step_begin * BF_BM;
#pragma unroll
    for (int p = 0; p < 7; ++p) dma_pair_at(p, a.dxh + r0 * BF_D, a.xb + r0 * BF_D, a.c + r0 * a.n_p + n0, 0);
#else
#pragma unroll
    for (int p = 0; p < 7; ++p) dma_pair(p, (int64_t)step_begin * BF_BM, 0);
#endif
  }
  // make hipcc retire the W^T fragment loads HERE: otherwise it places its vmcnt waits for them inside the loop,
  // where they would also wait for the (untracked) LDS-DMA of the next step and serialise copy and compute
#pragma unroll
  for (int kk = 0; kk < 24; ++kk) asm volatile("" : "+v"(wfrag[kk]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#if BF_STATIC_STAGES
#pragma unroll
  for (int i = 0; i <= DIST; ++i) ring[i] = load_frag_s(std::integral_constant<int, 0>{}, i);     // includes gap 0 of the first step
  cf[0] = load_c_s(std::integral_constant<int, 0>{}, 0);
  cf[1] = load_c_s(std::integral_constant<int, 0>{}, 1);
#else
#pragma unroll
  for (int i = 0; i <= DIST; ++i) ring[i] = load_frag(smem, i);     // includes gap 0 of the first step
  cf[0] = tr_pair(smem + 2 * BF_DXH_BYTES + coff0, smem + 2 * BF_DXH_BYTES + coff1);
  cf[1] = tr_pair(smem + 2 * BF_DXH_BYTES + 4096 + coff0, smem + 2 * BF_DXH_BYTES + 4096 + coff1);
#endif
  if (a.clk) { clk_t0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
#if BF_STATIC_STAGES
  // rows fetched during a step: those of the step after it (the last step re-copies its own rows into the idle stage instead of
  // branching around the DMA) -- as running pointers, advanced by one step's stride while a later step exists
  const int64_t nrow_first = (int64_t)(step_begin + 1 < step_end ? step_begin + 1 : step_begin) * BF_BM;
  const bf16_t* nd = a.dxh + nrow_first * BF_D;
  const bf16_t* nx = a.xb + nrow_first * BF_D;
  const bf16_t* nc = a.c + nrow_first * a.n_p + n0;
  const int64_t c_stride = (int64_t)BF_BM * a.n_p;
  auto step_body = [&](auto cur_tag, int step) {
    constexpr int CUR = decltype(cur_tag)::value;
    using CurT = std::integral_constant<int, CUR>;
    using NxtT = std::integral_constant<int, CUR ^ 1>;
    f32x16 dc;
    bool gate_cur = false, gate_next = false;
#else
  int cur = 0;
  for (int step = step_begin; step < step_end; ++step) {
    // the last step re-copies its own rows into the idle stage instead of branching around the DMA
    const int64_t next_row0 = (int64_t)(step + 1 < step_end ? step + 1 : step) * BF_BM;
    const char* img_d = smem + cur * BF_STAGE_BYTES;
    const char* nxt_d = smem + (cur ^ 1) * BF_STAGE_BYTES;
    f32x16 dc;
    bool gate_cur = false, gate_next = false;
#endif

#pragma unroll
    for (int i = 0; i < 72; ++i) {
      const bf16x8 fa = ring[i % RING];
#ifdef BF_PROXY16
      // TIMING PROXY ONLY (tools/build_variant.sh proxy16 -DBF_PROXY16; results are WRONG): every 32x32x16 MFMA replaced by two
      // 16x16x32 MFMAs on the same operand registers and a quarter each of the same accumulator -- the same FLOPs, the same LDS
      // and register traffic, the other MFMA shape: what clock / time would a 16x16x32 backward get under the chip's power limit?
      typedef __attribute__((ext_vector_type(4))) float f32x4_;
#define BF_TWO16(C, A, Bv, ODD)                                                                                   \
  do {                                                                                                             \
    f32x4_ q0_ = (ODD) ? __builtin_shufflevector(C, C, 4, 5, 6, 7) : __builtin_shufflevector(C, C, 0, 1, 2, 3);      \
    f32x4_ q1_ = (ODD) ? __builtin_shufflevector(C, C, 12, 13, 14, 15) : __builtin_shufflevector(C, C, 8, 9, 10, 11); \
    q0_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, Bv, q0_, 0, 0, 0);                                            \
    q1_ = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, Bv, q1_, 0, 0, 0);                                            \
    if (ODD) { C[4] = q0_[0]; C[5] = q0_[1]; C[6] = q0_[2]; C[7] = q0_[3]; C[12] = q1_[0]; C[13] = q1_[1]; C[14] = q1_[2]; C[15] = q1_[3]; } \
    else { C[0] = q0_[0]; C[1] = q0_[1]; C[2] = q0_[2]; C[3] = q0_[3]; C[8] = q1_[0]; C[9] = q1_[1]; C[10] = q1_[2]; C[11] = q1_[3]; }       \
  } while (0)
      if (i == 0) {
        dc = cinit;
#pragma unroll
        for (int z = 8; z < 16; ++z) dc[z] = cinit[z] * (1.0f + 1e-6f * (float)z);      // (keeps the two half-chains distinct: no CSE)
        BF_TWO16(dc, fa, wfrag[0], false);
      } else if (i < 24) {
        BF_TWO16(dc, fa, wfrag[i], (i & 1) != 0);
      } else if (i < 48) {
        BF_TWO16(acc[(i - 24) >> 1], fa, cf[i & 1], (i & 1) != 0);
      } else {
        BF_TWO16(acc[(i - 48) >> 1], fa, pf[i & 1], (i & 1) != 0);
      }
#else
      if (i == 0) {
        asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(dc) : "v"(fa), "v"(wfrag[0]), "v"(cinit));
      } else if (i < 24) {
        asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(dc) : "v"(fa), "v"(wfrag[i]));
      } else if (i < 48) {
        acc[(i - 24) >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, cf[i & 1], acc[(i - 24) >> 1], 0, 0, 0);
      } else {
        acc[(i - 48) >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, pf[i & 1], acc[(i - 48) >> 1], 0, 0, 0);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
      // ---- gap g = i + 1 (issues while MFMA i occupies the matrix pipe); gap 72 is gap 0 of the next step
      const int g = i + 1;
#if BF_STATIC_STAGES
      if (g == 72 - DIST) {
        // every read of this stage has been issued and returned, this wave's DMA pieces of the next step have landed
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // ... and everybody's
        cf[0] = load_c_s(NxtT{}, 0);
      }
      if (g == 72 - DIST + 1) cf[1] = load_c_s(NxtT{}, 1);
      if (g + DIST < 72) ring[(g + DIST) % RING] = load_frag_s(CurT{}, g + DIST);
      else ring[(g + DIST) % RING] = load_frag_s(NxtT{}, g + DIST - 72);
#if BF_DMA_SINGLE
      if (g >= 1 && (g - 1) % 2 == 0 && (g - 1) / 2 < 14) dma_one_at((g - 1) / 2, nd, nx, nc, CUR ^ 1);      // experiment: one piece per second gap
#else
      if (g >= BF_DMA_B && (g - BF_DMA_B) % BF_DMA_A == 0 && (g - BF_DMA_B) / BF_DMA_A < 7)
        dma_pair_at((g - BF_DMA_B) / BF_DMA_A, nd, nx, nc, CUR ^ 1);
#endif
#else
      if (g == 72 - DIST) {
        // every read of this stage has been issued and returned, this wave's DMA pieces of the next step have landed
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // ... and everybody's
        cf[0] = tr_pair(nxt_d + 2 * BF_DXH_BYTES + coff0, nxt_d + 2 * BF_DXH_BYTES + coff1);
      }
      if (g == 72 - DIST + 1) cf[1] = tr_pair(nxt_d + 2 * BF_DXH_BYTES + 4096 + coff0, nxt_d + 2 * BF_DXH_BYTES + 4096 + coff1);
      if (g + DIST < 72) ring[(g + DIST) % RING] = load_frag(img_d, g + DIST);
      else ring[(g + DIST) % RING] = load_frag(nxt_d, g + DIST - 72);
      if (g >= BF_DMA_B && (g - BF_DMA_B) % BF_DMA_A == 0 && (g - BF_DMA_B) / BF_DMA_A < 7)
        dma_pair((g - BF_DMA_B) / BF_DMA_A, next_row0, cur ^ 1);
#endif
#if BF_GATE_AHEAD
      // Round 6: the gate's COMPARE one gap ahead of its select.  `v_cmp` writes a scalar register and gfx950 wants two wait states before
      // a vector instruction reads it; with compare and select in the same gap (nothing else may move in between: the gaps are fenced)
      // hipcc filled them with `s_nop 1` -- sixteen per step on the wave's one issue port.  One gap apart the distance is there for free.
      if (g >= 29 && g < 45) {
        const int e1 = g - 29;
        gate_next = (float)cf[e1 >> 3][e1 & 7] > 0.f;
      }
      if (g >= 30 && g < 46) {      // one dpre element per gap: dc register e, gated by c at the same (row, col)
        const int e = g - 30, s2 = e >> 3, j = e & 7;
        const float gv = gate_cur ? dc[e] : 0.f;                    // (1/M term rides in the accumulator) bf16 once, when packed
        db_acc += gv;
        pf[s2][j] = (bf16_t)gv;
      }
      if (g >= 29 && g < 45) gate_cur = gate_next;
#else
      if (g >= 30 && g < 46) {      // one dpre element per gap: dc register e, gated by c at the same (row, col)
        const int e = g - 30, s2 = e >> 3, j = e & 7;
        const float gv = ((float)cf[s2][j] > 0.f) ? dc[e] : 0.f;    // (1/M term rides in the accumulator) bf16 once, when packed
        db_acc += gv;
        pf[s2][j] = (bf16_t)gv;
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
#if BF_STATIC_STAGES
    if (step + 2 < step_end) {      // (wave-uniform: scalar pointer arithmetic)
      nd += BF_BM * BF_D;
      nx += BF_BM * BF_D;
      nc += c_stride;
    }
  };
  // pairs of steps (stage 0, stage 1) in a branch-free loop body, then the odd step if there is one (a conditional second half inside
  // the loop merges 300 live registers at the back edge: the allocator answered with 680 bytes of scratch)
  {
    int step = step_begin;
    for (; step + 1 < step_end; step += 2) {
      step_body(std::integral_constant<int, 0>{}, step);
      step_body(std::integral_constant<int, 1>{}, step + 1);
    }
    if (step < step_end) step_body(std::integral_constant<int, 0>{}, step);
  }
#else
    cur ^= 1;
  }
#endif
  if (a.clk) {
    clk_loop += __builtin_amdgcn_s_memtime() - clk_t0;
    clk_real += __builtin_amdgcn_s_memrealtime() - clk_r0;
    clk_steps += (unsigned long long)(step_end - step_begin);
  }

  // ---- epilogue: dW slab of this row range (rows d = 32 dt + (r&3) + 8 (r>>2) + 4 h, column nw + lane&31)
  {
    float* out = a.slab + (int64_t)split * BF_D * a.n_p;
    const int h = lane >> 5;
    const unsigned off0 = (unsigned)(nw + (lane & 31)) + (unsigned)(4 * h) * (unsigned)a.n_p;
#pragma unroll
    for (int dt = 0; dt < 12; ++dt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned drow = 32 * dt + (r & 3) + 8 * (r >> 2);
        out[off0 + drow * (unsigned)a.n_p] = acc[dt][r] * scale;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    db_acc += __shfl_xor(db_acc, 32, 64);
    if (lane < 32) a.db_part[(int64_t)split * a.n_p + nw + lane] = db_acc * scale;
  }
  }   // segments
  if (a.clk && threadIdx.x == 0) {
    unsigned long long* o = a.clk + (int64_t)blockIdx.x * 4;
    o[0] = clk_loop; o[1] = clk_real; o[2] = clk_steps; o[3] = 0;
  }
  if (a.clk && threadIdx.x == 0) {   // whole-kernel cycles of this workgroup in the upper half of the stamp buffer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    a.clk[(int64_t)(gridDim.x + blockIdx.x) * 4] = __builtin_amdgcn_s_memtime() - clk_k0;
  }
}
