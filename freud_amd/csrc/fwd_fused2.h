// Fused forward of the tied-weight L1 SAE for d_model (padded) == 384 on gfx950, second decomposition.
// Same arithmetic, same outputs and the same arguments as fwd_fused_d384_kernel (fwd_fused.h; reference
// src/models/l1autoencoder.py:69-95, mse_loss :29-36); what changes is who multiplies what in the DECODER product.
//
// Why: in fwd_fused.h every MFMA takes its A operand from LDS (W^T rows for the encoder, transposed reads of the same image for
// the decoder) -- 1 KiB per MFMA and wave, 4 waves, 32 cycles per MFMA = 128 B/clk, which IS the LDS bandwidth of a CU.  With
// the DMA writes, bias and staging traffic on top the kernel asks for ~165 B/clk at full matrix rate and runs at ~82 % of the
// LDS peak: 60-63 % MFMA busy, and knocking out the decoder's fragment reads alone returns 16 % (profiles/r03_fwd_knockouts.txt).
// Register blocking is the only way to feed more than one MFMA per fragment, and the [384 x 32] fp32 x_hat^T tile per wave
// (192 accumulator registers) leaves no room for a second row block.  So the decoder is split along d instead of along rows:
//
//   encoder (unchanged): wave w computes S^T[32 n x 32 rows of ITS row block] = W^T tile . x^T   (x fragments in registers)
//   bias / ReLU / L1 (unchanged) -> c^T of the row block, written to the latent staging image in LDS -- which already exists
//           (it feeds the full-line stores of the latent) and now doubles as the exchange buffer between the waves;
//   decoder: wave w owns the d-SLICE [96 w, 96 w + 96) of x_hat^T for ALL 128 rows of the workgroup:
//           x_hat^T[96 x 128] += W[96 x 32 n] . c^T[32 n x 128]  =  6 W fragments (transposed reads) x 8 c^T fragments
//           (ds_read_b128 from the staging image) for the same 24 MFMAs: every W fragment feeds 4 MFMAs, every c^T fragment 3.
//
// LDS reads per iteration and wave: 24 + 12 + 8 instead of 24 + 48; bytes 38 KiB instead of 48 KiB + ...; the accumulator
// budget is the same 192 registers ([96 x 128] = 12 tiles).  The price is one dependency across waves per tile: c^T of tile j
// is written during iteration j, published by the iteration's single barrier (at MFMA 40, which also hands over the DMA'd
// W^T tile as before) and multiplied in iteration j + 1.
//
// Staging image: two pair buffers [128 rows][128 B]; a row holds tiles 2t and 2t + 1 (64 B each) as four 16-byte chunks per
// tile, chunk (s, h) = the eight latents n = 16 s + 8 (q >> 2) + 4 h + (q & 3), q = 0..7 -- exactly the eight values lane
// (row, h) holds after the encoder for k-step s, and exactly the k order the transposed W reads deliver (tr_frag_perm), so
// writer and reader move whole 16-byte chunks; the chunk index is XORed with (row & 7) (conflict-free for both).  The drain to
// HBM reassembles natural column order from two 8-byte halves.
#pragma once
#include "fwd_fused.h"

// STAMP: diagnostic build (bench.py --dbg 65): s_memtime / s_memrealtime around the tile loop per wave into a.stamps[wg][wave][8]
// ([3] iterations, [4] loop cycles, [5] loop time in 10 ns ticks, [6] prologue cycles, [7] whole-kernel cycles).
#ifndef FF2_KO
#define FF2_KO 0
#endif
#ifndef FF2_EPI_PACKED
#define FF2_EPI_PACKED 1     // 1 = the packed residual arithmetic of the no-mask path (v_pk_add / v_pk_fma on pairs)
#endif
#ifndef FF2_L1_SUNK
#define FF2_L1_SUNK 0        // 1 = round 5's plain `l1_it += cv` (A/B: tools/build_variant.sh l1sunk -DFF2_L1_SUNK=1)
#endif
#ifndef FF2_EEDD
#define FF2_EEDD 0
#endif
#ifndef FF2_SPLIT_ELEM
#define FF2_SPLIT_ELEM 1     // round 6: a pair's latent arithmetic over four gaps, three vector instructions each (see the slot loop): -4.6 % loop cycles
#endif
#ifndef FF2_DMA_SPREAD
#define FF2_DMA_SPREAD 1     // round 6: the six LDS-DMA pieces one per gap, four gaps apart (see the slot loop): loop 1969 -> 1892 cycles
#endif
#ifndef FF2_PAIR_ROUND
#define FF2_PAIR_ROUND 1     // round 6: the rounding of S to bf16 (CPU autocast's GEMM output) for TWO elements by one v_cvt_pk_bf16_f32, brought back
#endif                       // to fp32 by a shift / a mask: 3 vector instructions per pair instead of 4, the same bits (0 = one cvt + shift per element)
#ifndef FF2_PAIR_FRAGS
#define FF2_PAIR_FRAGS 1     // round 6: encoder fragments requested two at a time, every fourth slot: the second MFMA of a pair needs no s_waitcnt of
#endif                       // its own (LDS reads return in order) -- 12 waits per iteration less on the wave's one issue port (0 = one per even slot)
#ifndef FF2_PACKED
#define FF2_PACKED 0         // 1 = the packed form below: 91 instead of 111 vector instructions per iteration and NOT faster (profiles/r05_ab_fwd_packed_valu.txt)
#endif
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

template <typename T, bool PAD, bool STAMP = false>
__global__ __launch_bounds__(256, 1) void fwd_fused2_d384_kernel(FwdFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int arow = lane & 31, ah = lane >> 5;
  const int wg = blockIdx.x + a.block_offset;
  const int64_t m0 = (int64_t)wg * FF_BM + 32 * w;             // first row of this wave's row block
  const int64_t mrow = m0 + arow;
  const bool row_ok = mrow < a.M;
  char* cst = smem + FF_RING_BYTES;                             // two pair buffers [128 rows][128 B]
  float* bias_s = reinterpret_cast<float*>(smem + FF_FIXED_LDS);
  unsigned long long clk_k0 = 0, clk_t0 = 0, clk_r0 = 0, clk_t1 = 0;
  if (STAMP) clk_k0 = __builtin_amdgcn_s_memtime();

#ifndef FF2_PRESCAN
#define FF2_PRESCAN 1
#endif
#ifndef FF2_SCAN_EARLY
#define FF2_SCAN_EARLY 1     // the -1.0 scan of x used in place: 1 = here in the prologue, as the fragments arrive (under the memory latency);
#endif                       // 0 = while the staging image is filled (first form of round 5: +1.3 k cycles in the epilogue)
  typedef __attribute__((ext_vector_type(2))) unsigned short us2_t;
  constexpr unsigned MASK_BITS2 = std::is_same<T, bf16_t>::value ? 0xBF80BF80u : 0xBC00BC00u;     // -1.0 twice, bf16 / fp16
  constexpr bool SCAN_EARLY = FF2_PRESCAN && FF2_SCAN_EARLY && !PAD && std::is_same<T, bf16_t>::value;
  bool wany_early = false;     // (wave-uniform) some 16-bit word of this wave's 32 x 384 block of a.xb reads -1.0

  bf16x8 xfrag[24];
#ifdef FF2_BIAS_V1
  if (t < 2 * FF_BN) bias_s[t] = a.bias[t];
#endif
  // ---- LDS-DMA plan of a W^T tile (as in fwd_fused.h)
  unsigned voff_t[6], loff_t[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int inst = w + 4 * i, sub = inst >> 3, row = 4 * (inst & 7) + (lane >> 4), pc = lane & 15;
    const int ch = pc ^ (((row & 3) << 2) | ((row >> 2) & 3));
    voff_t[i] = (unsigned)(row * (FF_D * 2) + (sub * 16 + ch) * 16);
    loff_t[i] = (unsigned)__builtin_amdgcn_readfirstlane(sub * 8192 + (inst & 7) * 1024);
  }
  typedef __attribute__((address_space(3))) char* lptr_t;
  const unsigned smem_base = (unsigned)(uintptr_t)(lptr_t)smem;
  auto dma_pair = [&](int p, int jt, int st) {
    const bf16_t* src = a.Wt + (int64_t)jt * FF_BN * FF_D;
    const unsigned dst = smem_base + st * FF_WT_BYTES;
    glds16_x2(src, src, voff_t[2 * p], voff_t[2 * p + 1], (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + loff_t[2 * p])),
              (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + loff_t[2 * p + 1])));
  };
  auto dma_one = [&](int k, int jt, int st) {       // piece k (0..5) of this wave's share of W^T tile jt into ring slot st
    const bf16_t* src = a.Wt + (int64_t)jt * FF_BN * FF_D;
    glds16(src, voff_t[k], (unsigned)__builtin_amdgcn_readfirstlane((int)(smem_base + st * FF_WT_BYTES + loff_t[k])));
  };
  const int last = a.ntiles - 1;
#pragma unroll
  for (int q = 0; q < 3; ++q)                       // prologue: W^T tiles 0, 1, 2 into slots 0, 1, 2
#pragma unroll
    for (int p = 0; p < 3; ++p) dma_pair(p, q <= last ? q : last, q);
#ifndef FF2_BIAS_V1
  // The first two tiles' biases by LDS-DMA like every later pair's (round 5).  As `bias_s[t] = a.bias[t]` this was a load, an
  // `s_waitcnt vmcnt(0)` and an LDS write in wave 0 -- behind that wave's 24 x loads: it sat out a whole memory latency before it issued
  // its share of the W^T tiles, and the other three waves waited for it at the barrier (the disassembly showed it; -DFF2_BIAS_V1 = that form).
  static_assert(2 * FF_BN == 64, "one 4-byte piece per lane of wave 0");
  if (w == 0) glds4(a.bias, (unsigned)(lane * 4), (unsigned)__builtin_amdgcn_readfirstlane((int)(smem_base + FF_FIXED_LDS)));
#endif
  {
#ifdef FF2_KO_XSAME      // knock-out (results wrong): every workgroup reads block 0's x -- the prologue with x served by the L2 instead of HBM
    const bf16_t* xp = a.xb + (int64_t)(32 * w + arow) * FF_D + 8 * ah;
#else
    const bf16_t* xp = a.xb + mrow * FF_D + 8 * ah;
#endif
#ifdef FF2_KO_XLOAD      // knock-out (results wrong): the same 24 KB per wave as fully coalesced 16-byte loads -- does the access pattern cost?
    const bf16_t* xc = a.xb + m0 * FF_D + lane * 8;
#pragma unroll
    for (int kk = 0; kk < 24; ++kk) xfrag[kk] = *reinterpret_cast<const bf16x8*>(xc + 512 * kk);
#else
#pragma unroll
    for (int kk = 0; kk < 24; ++kk) xfrag[kk] = *reinterpret_cast<const bf16x8*>(xp + 16 * kk);
#endif
  }
  // ring slot 3 and the staging image are read (times zero / as zeros) by the first iteration's decoder phase
  for (int i = t; i < FF_WT_BYTES / 16; i += 256) reinterpret_cast<u32x4*>(smem + 3 * FF_WT_BYTES)[i] = u32x4{0u, 0u, 0u, 0u};
  for (int i = t; i < FF_CST_BYTES / 16; i += 256) reinterpret_cast<u32x4*>(cst)[i] = u32x4{0u, 0u, 0u, 0u};

  f32x16 acc[12];                       // acc[4 dtl + mb]: rows d = 96 w + 32 dtl + ..., columns = row block mb
#pragma unroll
  for (int i = 0; i < 12; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // x AFTER the LDS-DMA pieces (round 5): the compiler counts only the loads it sees, so with the x loads youngest its vmcnt waits for the
  // fragments are exact, and the -1.0 scan below runs on each fragment as it arrives, under the latency of the ones behind it
  // (the W^T pieces are older and mostly L2 hits: they have landed by then).  Used only if a.x == a.xb.
  if (SCAN_EARLY) {
    us2_t mz = {0xFFFFu, 0xFFFFu};
#pragma unroll
    for (int kk = 0; kk < 24; ++kk) {
      const u32x4 xw = __builtin_bit_cast(u32x4, xfrag[kk]);
#pragma unroll
      for (int c = 0; c < 4; ++c) mz = __builtin_elementwise_min(mz, __builtin_bit_cast(us2_t, xw[c] ^ MASK_BITS2));
    }
    wany_early = __builtin_amdgcn_ballot_w64((mz[0] == 0) | (mz[1] == 0)) != 0ull;
  }
#pragma unroll
  for (int kk = 0; kk < 24; ++kk) asm volatile("" : "+v"(xfrag[kk]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- loop-invariant per-lane LDS offsets
  int roff[8];                       // encoder: row reads of a W^T tile
#pragma unroll
  for (int i = 0; i < 8; ++i) roff[i] = dual_off(arow, 2 * i + ah);
  // decoder: the six W fragments (k-step ks, slice tile dtl) of this wave's d-slice: two transposed 8-byte reads each
  int aoff0[6], aoff1[6];
  {
    const int g = (lane >> 4) & 1, q = (lane & 15) >> 2, p = lane & 3;
#pragma unroll
    for (int f = 0; f < 6; ++f) {
      const int ks = f / 3, dt = 3 * w + f % 3;
      const int col = 32 * (dt & 3) + 16 * g + 4 * p;
      const int r0 = 4 * ah + q, r1 = r0 + 8;
      const int base = (dt >> 2) * 8192 + ks * 4096;
      aoff0[f] = base + dual_off(r0, col >> 3) + (col & 7) * 2;
      aoff1[f] = base + dual_off(r1, col >> 3) + (col & 7) * 2;
    }
  }
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  auto tr_pair = [&](const char* p0, const char* p1) -> bf16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, p1));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  };
  // staging image: byte offset of chunk (tile half hf, k-step s, this lane's h) in row `arow` of a 32-row block
  // Chunk swizzle f(row) = (row & 7) ^ ((row >> 1) & 1) ^ ((row >> 4) & 1).  Round 3 used (row & 7): fine for the 16-byte writes
  // (8 consecutive rows per LDS cycle), but a ds_read_b128 is served in groups of 16 lanes -- rows {0-3, 12-15, 20-27} and
  // {4-11, 16-19, 28-31} -- where (row & 7) takes every value twice on rows of equal parity (the row pitch is half the banks):
  // a 2-way conflict on every c^T fragment read, and the 8-byte drain reads (4 rows per 32 lanes) hit rows r and r + 2 on the
  // same chunks.  SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE read 0.18.  With f, the 16 rows of either group give 16
  // distinct (row parity, chunk) pairs, 8 consecutive rows still give 8 distinct chunks, and rows r, r + 2 of a drain piece read
  // chunks of different parity.  -DFF2_SWZ_V1 restores the old image for A/B runs.
#ifdef FF2_SWZ_V1
  auto swz = [](int row) { return row & 7; };
#else
  auto swz = [](int row) { return (row & 7) ^ ((row >> 1) & 1) ^ ((row >> 4) & 1); };
#endif
  int boff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) boff[i] = arow * 128 + (((4 * (i >> 1) + 2 * (i & 1) + ah) ^ swz(arow)) << 4);
  const int wrow = w * 4096;                                     // this wave's row block inside a pair buffer
  // drain: piece p = rows 8 p .. 8 p + 7 of the row block; lane -> row 8 p + lane / 8, 16 output bytes = columns 8 o .. 8 o + 7
  const int drow_l = lane >> 3, dch = lane & 7;
  int doff0, doff1;
  {
    const int tl = dch >> 2, s = (dch & 3) >> 1, u = dch & 1;
    doff0 = drow_l * 128 + (((4 * tl + 2 * s + 0) ^ swz(drow_l)) << 4) + 8 * u;
    doff1 = drow_l * 128 + (((4 * tl + 2 * s + 1) ^ swz(drow_l)) << 4) + 8 * u;
  }
  // pieces 2 and 3 hold rows 16..31 of the row block: bit 4 of the row flips the chunk's low bit = bit 4 of the byte offset
#ifdef FF2_SWZ_V1
  auto dpiece = [](int off, int) { return off; };
#else
  auto dpiece = [](int off, int p) { return (p & 2) ? (off ^ 16) : off; };
#endif
  bf16_t* cdrain = a.c + (m0 + drow_l) * a.n_p + dch * 8;

  float l1_acc = 0.f;

  // ---- S(0): encoder product of tile 0 (outside the pipeline)
  f32x16 Sn;
#pragma unroll
  for (int r = 0; r < 16; ++r) Sn[r] = 0.f;
#pragma unroll
  for (int kk = 0; kk < 24; ++kk) {
    const bf16x8 fa = *reinterpret_cast<const bf16x8*>(smem + (kk >> 3) * 8192 + roff[kk & 7]);
    Sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, xfrag[kk], Sn, 0, 0, 0);
  }

  // ---- pipeline.  Iteration j issues 48 MFMAs, decoder and encoder INTERLEAVED slot by slot:
  //   even slots: decoder of tile j-1:  acc[4 dtl + mb] += W fragment (ks, dtl) . c^T fragment (mb, ks)      m = 12 ks + 3 mb + dtl
  //   odd slots:  encoder of tile j+1:  S(j+1) += W^T rows (slot (j+1) % 4) . x^T
  // with the bias / ReLU / L1 work that turns S(j) into c(j) in the even gaps 4..34 (two 16-byte staging writes, gaps 18 and
  // 34), ONE barrier at gap 40 -- by then every wave's c(j) is in the staging image and the W^T tile j+2 (DMA'd during the
  // previous iteration) has landed -- and after it the DMA of tile j+3 into the slot whose tile j-1 was last read at gap 17,
  // the first c^T / W fragments of the next decoder phase, the first encoder fragments of tile j+2, the drain of the previous
  // tile pair.
  constexpr int RING = 8;
  const char *rlo[8], *rhi[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    rlo[i] = smem + roff[i];
    rhi[i] = smem + 2 * FF_WT_BYTES + roff[i];
  }
  const char *a0lo[6], *a1lo[6], *a0hi[6], *a1hi[6];
#pragma unroll
  for (int f = 0; f < 6; ++f) {
    a0lo[f] = smem + aoff0[f];
    a1lo[f] = smem + aoff1[f];
    a0hi[f] = smem + 2 * FF_WT_BYTES + aoff0[f];
    a1hi[f] = smem + 2 * FF_WT_BYTES + aoff1[f];
  }
  const char* bptr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) bptr[i] = cst + boff[i];
  auto dec_a = [&](int slot, int f) -> bf16x8 {          // W fragment f = 3 ks + dtl of ring slot `slot`
    const int off = (slot & 1) * FF_WT_BYTES;
    return tr_pair((slot < 2 ? a0lo : a0hi)[f] + off, (slot < 2 ? a1lo : a1hi)[f] + off);
  };
  auto enc_frag = [&](int slot, int kk) -> bf16x8 {
    const int off = (slot & 1) * FF_WT_BYTES + (kk >> 3) * 8192;
    return *reinterpret_cast<const bf16x8*>((slot < 2 ? rlo : rhi)[kk & 7] + off);
  };
  auto dec_b = [&](int pb, int hf, int b) -> bf16x8 {    // c^T fragment b = 4 ks + mb of pair buffer pb, tile half hf
    const int ks = b >> 2, mb = b & 3;
    return *reinterpret_cast<const bf16x8*>(bptr[2 * hf + ks] + pb * 16384 + mb * 4096);
  };

  bf16x8 ring[RING], Afr[6], Bq[4];
  f32x16 SA = Sn, SB;
#pragma unroll
  for (int r = 0; r < 16; ++r) SB[r] = 0.f;
  // decoder of "tile -1": zeros from pair buffer 1, half 1 (cleared above) times the cleared ring slot 3; the first three
  // encoder fragments of tile 1
#pragma unroll
  for (int f = 0; f < 3; ++f) Afr[f] = dec_a(3, f);
  Bq[0] = dec_b(1, 1, 0);
  Bq[1] = dec_b(1, 1, 1);
#pragma unroll
  for (int k = 0; k < 3; ++k) ring[k] = enc_frag(1, k);

#ifdef FF2_WAITSTAMP
  unsigned long long ws_vm = 0, ws_bar = 0, ws_gap = 0, ws_prev = 0;
#endif
  u32x4 dr[2];
  bf16_t* const dummy_line = a.c + (int64_t)a.c_rows * a.n_p + lane * 8;
  auto body = [&](auto ph_tag, int j) {
    constexpr int PH = decltype(ph_tag)::value;          // == j % 4
    constexpr int SLOT_D = (PH + 3) & 3, SLOT_DN = PH, SLOT_E = (PH + 1) & 3, SLOT_EN = (PH + 2) & 3, SLOT_DMA = (PH + 3) & 3;
    constexpr int PB_W = (PH >> 1) & 1, HF_W = PH & 1;               // where c(j) goes
    constexpr int PB_D = (((PH + 3) & 3) >> 1) & 1, HF_D = (PH + 3) & 1;   // where c(j-1) is read (this iteration's decoder)
    f32x16& Scur = (PH & 1) ? SB : SA;
    f32x16& Snxt = (PH & 1) ? SA : SB;
    const int jt = j + 3 <= last ? j + 3 : last;                     // DMA source (clamped in the tail)
    const float* bj = bias_s + (j & (FF_BIAS_RING_TILES - 1)) * FF_BN;
    const char* cst_r = cst + (PB_W ^ 1) * 16384 + wrow;             // pair buffer drained (tiles 2t-2, 2t-1), this wave's rows
    bf16_t* dst_pair = j >= 2 ? cdrain + 64 * ((j >> 1) - 1) : dummy_line;
    const int64_t dst_rstride = j >= 2 ? (int64_t)a.n_p : 0;
    f32x4 bq[4];
    float l1_it = 0.f;
    f32x2_t l1_pk = {0.f, 0.f}, pr = {0.f, 0.f};
    unsigned se_u = 0u;
    float se_t0 = 0.f, se_t1 = 0.f;
    bf16x8 cw;
    const f32x16 S = Scur;

    // 48 MFMA slots: EVEN slot i = decoder MFMA i / 2 of tile j-1 (m = 12 ks + 3 mb + dtl: accumulators rotate), ODD slot i =
    // encoder MFMA i / 2 of tile j+1 (a chain through one accumulator: interleaved with the decoder its dependent issue is
    // never back to back).  Fragments are requested >= 6 slots ahead.
    // (FF2_PACKED: static_for -- the slot index as a template constant: `#pragma unroll` gives up silently above LLVM's size
    // threshold, and with the packed form's few added lines the loop stayed rolled and indexed every register array dynamically,
    // 1.8 KB of scratch per lane.  The shipped element-wise form keeps round 4's pragma loop: bit-identical code.)
#if FF2_PACKED || defined(FF2_WAITSTAMP) || !FF2_L1_SUNK
#define FF2_STATIC_LOOP 1
    static_for<0, 48>([&](auto slot_tag) {
      constexpr int i = decltype(slot_tag)::value;
#else
#define FF2_STATIC_LOOP 0
#pragma unroll
    for (int i = 0; i < 48; ++i) {
#endif
      // ---- fragment prefetch
#if FF2_PAIR_FRAGS
      // encoder fragments k = i/2 + 3 and i/2 + 4 together, every fourth slot (consumed at slots i + 7 and i + 9; the ring holds 8)
      // -- the LATER one first: the wait in front of the earlier MFMA (its fragment is the younger request) then covers both
      if ((i & 3) == 0 && i <= 40) {
        if (i / 2 + 4 <= 23) ring[(i / 2 + 4) % RING] = enc_frag(SLOT_E, i / 2 + 4);
        ring[(i / 2 + 3) % RING] = enc_frag(SLOT_E, i / 2 + 3);
      }
#else
      if ((i & 1) == 0 && i <= 40) ring[(i / 2 + 3) % RING] = enc_frag(SLOT_E, i / 2 + 3);         // encoder k = i/2 + 3 (slot i + 7)
#endif
      if (i == 42 || i == 44 || i == 46) ring[(i - 42) / 2] = enc_frag(SLOT_EN, (i - 42) / 2);      // k = 0..2 of the NEXT iteration
      if (i == 15 || i == 16 || i == 17) Afr[i - 12] = dec_a(SLOT_D, i - 12);                       // k-step 1 (decoder MFMAs 12..23)
      if (i == 45 || i == 46 || i == 47) Afr[i - 45] = dec_a(SLOT_DN, i - 45);                      // k-step 0 of the next phase
      // c^T fragment b feeds decoder MFMAs 3 b .. 3 b + 2 (slots 6 b .. 6 b + 4): requested at slot 6 b - 7
      if (i >= 5 && i <= 35 && (i - 5) % 6 == 0) Bq[((i + 7) / 6) & 3] = dec_b(PB_D, HF_D, (i + 7) / 6);
      if (i == 41) Bq[0] = dec_b(PB_W, HF_W, 0);                                                    // next phase, after the barrier
      if (i == 47) Bq[1] = dec_b(PB_W, HF_W, 1);
      if (i < 4) bq[i] = *reinterpret_cast<const f32x4*>(bj + 8 * i + 4 * ah);
      if (i == 40) {
        // <= 1 VMEM operation outstanding (this iteration's first latent store): the six DMA pieces of tile j+2 are done;
        // <= 2 LDS operations outstanding (four fragment reads follow the staging write of gap 34): that write is done
        __builtin_amdgcn_sched_barrier(0);
#ifdef FF2_WAITSTAMP
        // diagnostic build (bench.py --dbg 65 with FREUD_FF2_WAITSTAMP=1): what the iteration's single hand-over
        // costs THIS wave -- w0..w1: its own LDS-DMA pieces of tile j+2 and its staging write (the counted wait), w1..w2: the other
        // three waves (the barrier).  s_memtime is a scalar-memory read: each stamp drains lgkmcnt, so the build is ~10 % slower
        // and the split, not the total, is what it is for.
        const unsigned long long w0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
        const unsigned long long w1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const unsigned long long w2 = __builtin_amdgcn_s_memtime();
        ws_vm += w1 - w0; ws_bar += w2 - w1;
        if (ws_prev != 0) ws_gap += w0 - ws_prev;      // barrier exit of iteration j-1 -> arrival at this hand-over
        ws_prev = w2;
#else
        asm volatile("s_waitcnt vmcnt(1) lgkmcnt(2)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
#if FF2_DMA_SPREAD
      // Round 6: the six LDS-DMA pieces of tile j+3 ONE per gap and four gaps apart -- two right behind this iteration's hand-over, four
      // in the first gaps of the NEXT iteration (that body issues its predecessor's pieces 2..5: slot and source of phase PH - 1) --
      // instead of three pairs in gaps 41 / 43 / 45.  A piece issued right behind another waits for the address path (the backward's
      // placement sweep, bwd_fused.h: back-to-back pieces cost more each), and a gap with two of them is ~150 cycles of issue behind a
      // 32-cycle MFMA.  The slot is free for all of them (every wave has passed the hand-over behind which tile j-1 was last read);
      // they land a thousand cycles before the next hand-over's vmcnt wait, which still counts them (they are older than the latent
      // store it lets pass).  First iteration: the early pieces re-copy tile 2 into slot 2 (as the prologue did); after the last one
      // the pieces not issued are redundant copies of the last tile.
      if (i == 41 || i == 45) dma_one((i - 41) / 4, jt, SLOT_DMA);
      if (i == 1 || i == 5 || i == 9 || i == 13) dma_one(2 + (i - 1) / 4, j + 2 <= last ? j + 2 : last, (PH + 2) & 3);
#else
      if (i == 41 || i == 43 || i == 45) dma_pair((i - 41) / 2, jt, SLOT_DMA);
#endif
      if ((PH & 1) == 0 && i == 46 && w == 0) {
        const int jb = j + 2 <= a.ntiles - 2 ? j + 2 : a.ntiles - 2;
        glds4(a.bias + (int64_t)jb * FF_BN, (unsigned)(lane * 4),
              (unsigned)__builtin_amdgcn_readfirstlane((int)(smem_base + FF_FIXED_LDS + ((j + 2) & (FF_BIAS_RING_TILES - 1)) * FF_BN * 4)));
      }
#if FF2_PACKED
      // latent elements in PAIRS (round 5): pair p = elements 2 p, 2 p + 1 of S at the gaps 4 + 4 p (both rounded to bf16 by ONE
      // v_cvt_pk_bf16_f32, then back to fp32: two shifts) and 6 + 4 p (bias by one v_pk_add_f32, two max, the L1 sum by one
      // v_pk_add_f32, the pair's bf16 by one v_cvt_pk).  The loop is bound by the vector instructions it issues between its MFMAs
      // (113 per iteration by the disassembly: 32 adds, 24 conversions, 16 shifts, 16 max, 16 AGPR reads); this form issues 24 fewer.
      // Same values per element; the L1 partial sums associate differently (two chains instead of one).
      if (i >= 4 && i <= 32 && (i & 3) == 0) {
        const int e = (i - 4) >> 1;                      // S register e <-> column n = (e&3) + 8 (e>>2) + 4 h
        const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{S[e], S[e + 1]}, bf16x2));
        pr = f32x2_t{__uint_as_float(u << 16), __uint_as_float(u & 0xFFFF0000u)};
      }
      if (i >= 6 && i <= 34 && (i & 3) == 2) {
        const int e = (i - 6) >> 1;
        // rounded to bf16 BEFORE the fp32 bias add (CPU autocast).  (inline asm: hipcc scalarises this one packed add into two
        // v_add_f32 although both operands sit in aligned register pairs)
        f32x2_t sv;
        const f32x2_t bp = {bq[e >> 2][e & 3], bq[e >> 2][(e & 3) + 1]};
        asm("v_pk_add_f32 %0, %1, %2" : "=v"(sv) : "v"(pr), "v"(bp));
        // ReLU on the BITS (signed integer max with 0: negative floats, -0.0 included, are negative integers): fmaxf on the
        // output of the asm above costs a second, canonicalising v_max per element
        sv = f32x2_t{__int_as_float(max(__float_as_int(sv[0]), 0)), __int_as_float(max(__float_as_int(sv[1]), 0))};
        if (PAD) sv = row_ok ? sv : f32x2_t{0.f, 0.f};
        l1_pk += sv;
        cw[e & 7] = (bf16_t)sv[0];
        cw[(e & 7) + 1] = (bf16_t)sv[1];
        if ((e & 7) == 6)      // the eight values of k-step e >> 3: one 16-byte chunk of this lane's row
          *reinterpret_cast<bf16x8*>(cst + PB_W * 16384 + wrow + boff[2 * HF_W + (e >> 3)]) = cw;
      }
#elif FF2_SPLIT_ELEM
      // Round 6: the latent arithmetic of a PAIR of elements spread over FOUR consecutive gaps, three vector instructions each, instead of
      // six in each of two even gaps and none in the odd ones.  One wave per SIMD issues in order: a gap with six vector instructions,
      // a fragment read and a wait is ~44 cycles of issue behind a 32-cycle MFMA, and the light gap next to it cannot give the time back.
      // Same operations on the same values in the same order (the L1 additions stay e = 0, 1, 2, ...).
      if (i >= 4 && i <= 35) {
        const int p = (i - 4) >> 2, ph = (i - 4) & 3, e0 = 2 * p;       // pair p = elements e0, e0 + 1 (S register e <-> column n = (e&3) + 8 (e>>2) + 4 h)
        if (ph == 0) {
          se_u = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{S[e0], S[e0 + 1]}, bf16x2));      // rounded to bf16 BEFORE the fp32 bias add (CPU autocast)
        } else if (ph == 1) {
          se_t0 = __uint_as_float(se_u << 16) + bq[e0 >> 2][e0 & 3];
          se_t1 = __uint_as_float(se_u & 0xFFFF0000u);
        } else if (ph == 2) {
          se_t1 = se_t1 + bq[e0 >> 2][(e0 & 3) + 1];
          se_t0 = fmaxf(se_t0, 0.f);
          se_t1 = fmaxf(se_t1, 0.f);
          if (PAD) { se_t0 = row_ok ? se_t0 : 0.f; se_t1 = row_ok ? se_t1 : 0.f; }
        } else {
          // (one statement: between two asm statements hipcc puts an s_nop of its own)
          asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(l1_it) : "v"(se_t0), "v"(se_t1));
          cw[e0 & 7] = (bf16_t)se_t0;
          cw[(e0 & 7) + 1] = (bf16_t)se_t1;
          if ((e0 & 7) == 6)      // the eight values of k-step e0 >> 3: one 16-byte chunk of this lane's row
            *reinterpret_cast<bf16x8*>(cst + PB_W * 16384 + wrow + boff[2 * HF_W + (e0 >> 3)]) = cw;
        }
      }
#else
      // latent element e at gap 4 + 2 e
      if (i >= 4 && i <= 34 && (i & 1) == 0) {
        const int e = (i - 4) >> 1;                      // S register e <-> column n = (e&3) + 8 (e>>2) + 4 h
#if FF2_PAIR_ROUND
        if ((e & 1) == 0) {      // both elements of the pair rounded by ONE conversion (round to nearest even, like bf16_round)
          const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2_t{S[e], S[e + 1]}, bf16x2));
          pr = f32x2_t{__uint_as_float(u << 16), __uint_as_float(u & 0xFFFF0000u)};
        }
        float cv = fmaxf(pr[e & 1] + bq[e >> 2][e & 3], 0.f);           // rounded to bf16 BEFORE the fp32 bias add (CPU autocast)
#else
        float cv = fmaxf(bf16_round(S[e]) + bq[e >> 2][e & 3], 0.f);    // rounded to bf16 BEFORE the fp32 bias add (CPU autocast)
#endif
        if (PAD) cv = row_ok ? cv : 0.f;
#if FF2_L1_SUNK
        l1_it += cv;
#else
        // Round 6 (the disassembly): written as `l1_it += cv`, LLVM SANK the sixteen additions of three of the four unrolled bodies to
        // the loop latch -- the values stay live in sixteen registers each and 51 dependent v_add_f32 run back to back at the back edge
        // with no MFMA between them (~200 cycles per four iterations with the matrix pipe idle).  As an opaque instruction the addition
        // stays in its MFMA gap; same operands, same order, same sum.
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(l1_it) : "v"(cv));
#endif
        cw[e & 7] = (bf16_t)cv;
        if ((e & 7) == 7)      // the eight values of k-step e >> 3: one 16-byte chunk of this lane's row
          *reinterpret_cast<bf16x8*>(cst + PB_W * 16384 + wrow + boff[2 * HF_W + (e >> 3)]) = cw;
      }
#endif
      // two full-line pieces of the finished pair per iteration: LDS reads in one gap, global store 8 gaps later
      if (i == 29 || i == 33) {
        const int p = 2 * (PH & 1) + (i == 33);
        const uint2 lo = *reinterpret_cast<const uint2*>(cst_r + p * 1024 + dpiece(doff0, p));
        const uint2 hi = *reinterpret_cast<const uint2*>(cst_r + p * 1024 + dpiece(doff1, p));
        dr[i == 33] = u32x4{lo.x, lo.y, hi.x, hi.y};
      }
      if (i == 37 || i == 42) {
        const int p = 2 * (PH & 1) + (i == 42);
        __builtin_nontemporal_store(dr[i == 42], reinterpret_cast<u32x4*>(dst_pair + (int64_t)(8 * p) * dst_rstride));
      }
      __builtin_amdgcn_sched_barrier(0);
#if FF2_EEDD
      // (experiment, round 6: the two MFMAs of slots 4 q + 2 and 4 q + 3 exchanged -- D E E D D E E D ... -- so that every second encoder MFMA
      // follows its predecessor in the accumulation chain directly; same operands, same chain order, bit-identical)
      if ((i & 3) == 0 || (i & 3) == 3) {
#else
      if ((i & 1) == 0) {
#endif
        const int m = i / 2, ks = m / 12, mb = (m % 12) / 3, dtl = m % 3;
        acc[4 * dtl + mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Afr[3 * ks + dtl], Bq[(4 * ks + mb) & 3], acc[4 * dtl + mb], 0, 0, 0);
      } else if (i == 1) {
        f32x16 zero;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero[r] = 0.f;
        Snxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[0], xfrag[0], zero, 0, 0, 0);
      } else {
        Snxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[(i / 2) % RING], xfrag[i / 2], Snxt, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#if FF2_STATIC_LOOP
    });
#else
    }
#endif
    l1_acc += l1_it + (l1_pk[0] + l1_pk[1]);
  };
  if (STAMP) { clk_t0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
  for (int j4 = 0; j4 < a.ntiles; j4 += 4) {            // ntiles is a multiple of 4 (n_p is a multiple of 128)
    body(std::integral_constant<int, 0>{}, j4);
    body(std::integral_constant<int, 1>{}, j4 + 1);
    body(std::integral_constant<int, 2>{}, j4 + 2);
    body(std::integral_constant<int, 3>{}, j4 + 3);
  }
  if (STAMP) {
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    clk_t1 = t1;
    if (lane == 0) {
      unsigned long long* o = a.stamps + ((int64_t)wg * 4 + w) * 8;
      o[0] = o[1] = o[2] = 0; o[3] = (unsigned long long)a.ntiles; o[4] = t1 - clk_t0; o[5] = r1 - clk_r0; o[6] = clk_t0 - clk_k0;
    }
  }
  // ---- final half iteration: decoder of the last tile (slot 3, pair buffer 1, half 1); its k-step 0 W fragments and its
  // first two c^T fragments were requested at the end of the last iteration (after its barrier)
  Bq[2] = dec_b(1, 1, 2);
#pragma unroll
  for (int m = 0; m < 24; ++m) {
    if (m == 3 || m == 4 || m == 5) Afr[m] = dec_a(3, m);
    if (m % 3 == 0 && m / 3 + 3 <= 7) Bq[(m / 3 + 3) & 3] = dec_b(1, 1, m / 3 + 3);
    __builtin_amdgcn_sched_barrier(0);
    const int ks = m / 12, mb = (m % 12) / 3, dtl = m % 3;
    acc[4 * dtl + mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Afr[3 * ks + dtl], Bq[(4 * ks + mb) & 3], acc[4 * dtl + mb], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  // drain the last pair of latent tiles (this wave's rows; published by the last iteration's barrier)
  {
    const char* cst_r = cst + (((a.ntiles >> 1) - 1) & 1) * 16384 + wrow;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const uint2 lo = *reinterpret_cast<const uint2*>(cst_r + p * 1024 + dpiece(doff0, p));
      const uint2 hi = *reinterpret_cast<const uint2*>(cst_r + p * 1024 + dpiece(doff1, p));
      __builtin_nontemporal_store(u32x4{lo.x, lo.y, hi.x, hi.y},
                                  reinterpret_cast<u32x4*>(cdrain + (int64_t)(8 * p) * a.n_p + 64 * ((a.ntiles >> 1) - 1)));
    }
  }
  // Everything but this wave's four youngest vector-memory operations -- the latent stores just above -- must be done: the LDS-DMA
  // pieces of the loop's last iterations still write ring slots that the epilogue's staging image overlays.  (Round 4 waited for
  // vmcnt(0) here and again inside __syncthreads: a drain of those stores with the matrix pipe idle.)
#ifndef FF2_TAIL_V1
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
#else
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
#endif

  unsigned long long clk_e0 = 0, clk_e1 = 0, clk_e2 = 0;
  if (STAMP) clk_e0 = __builtin_amdgcn_s_memtime();      // final half iteration + drain of the last latent pair done
  // ---- epilogue: x_hat^T accumulators -> residual / dx_hat / squared-error sums.
  // acc[4 dtl + mb][r] <-> d = 96 w + 32 dtl + (r&3) + 8 (r>>2) + 4 h, row 128 wg + 32 mb + arow: every wave needs all 128 x
  // rows (for its d-slice), so the whole [128 x 384] x block is staged in the (now idle) rings, transformed in place into
  // dx_hat by the four waves, and leaves as coalesced 16-byte stores.
  float sq = 0.f, plain = 0.f, nmask = 0.f;
  float msq = 0.f;               // lean epilogue: squared error of the MASKED entries (masked MSE numerator = plain - msq)
  f32x2_t plain2 = {0.f, 0.f};   // packed form of the lean epilogue: the squared-error sum of the even / odd elements
  typedef __attribute__((ext_vector_type(4))) T Tx4;
  const bool vec_ok = (a.d == FF_D) && ((reinterpret_cast<uintptr_t>(a.x) & (sizeof(T) * 4 - 1)) == 0);
  constexpr int FF_DXH_PITCH = FF_D * 2 + 16;                    // 784 B
  static_assert(128 * FF_DXH_PITCH <= FF_FIXED_LDS, "the dx_hat staging must fit the idle LDS");
  char* stg = smem;                                              // [128 rows][784 B]
  constexpr bool X_VIA_LDS = !PAD && sizeof(T) == 2;
#ifndef FF2_X_FROM_FRAGS
#define FF2_X_FROM_FRAGS 1     // tools/build_variant.sh A/B switch: 0 = round 4's reload of x from global memory
#endif
  constexpr bool X_FROM_FRAGS = FF2_X_FROM_FRAGS && X_VIA_LDS && std::is_same<T, bf16_t>::value;
  // Round 5 (found by knock-outs, profiles/r05_fwd_residual_knockouts.txt): the residual arithmetic below took 15.8 k cycles per workgroup
  // with the -1.0 test in it and 8.4 k without -- not for the test's instructions (keeping them in the vector unit changed nothing) but for
  // the BRANCH behind it: `if (any lane met a -1.0)` per group of 16 elements ends a basic block, so each of the 12 groups waited for its own
  // LDS reads and its own dependent chain instead of overlapping with its neighbours (one wave per SIMD: nobody else fills the gaps).  So the
  // test moves to where x is staged -- every 16-bit x of the block passes through some lane's registers there exactly once: x ^ bits(-1.0) is
  // zero for a masked entry, a packed unsigned minimum carries it -- one flag per wave rides on the staging barrier, and the block takes
  // either the arithmetic with no test and no branch (one basic block), or, if any of its 128 x 384 entries is masked, the per-group form.
  // Same additions in the same order either way.  (-DFF2_PRESCAN=0 = the per-group test always.)
  unsigned* mflag = reinterpret_cast<unsigned*>(smem + FF_FIXED_LDS + 64);                         // bias ring: idle since the tile loop
  bool blk_masked = true;
  if (X_FROM_FRAGS && vec_ok && reinterpret_cast<const void*>(a.x) == reinterpret_cast<const void*>(a.xb)) {
    // Round 5: bf16 activations used in place (a.x == a.xb) are still in this wave's B-fragment registers from the prologue -- lane
    // (arow, ah) holds x[m0 + arow][16 kk + 8 ah .. + 7] in xfrag[kk] -- so the staging image is filled from REGISTERS: no second read of
    // the 96 KB of x per workgroup (long evicted from L2 by the latent stream) and no memory latency at the head of the epilogue.
    char* srow_w = stg + (32 * w + arow) * FF_DXH_PITCH + 16 * ah;
    us2_t mz = {0xFFFFu, 0xFFFFu};
#pragma unroll
    for (int kk = 0; kk < 24; ++kk) {
      *reinterpret_cast<bf16x8*>(srow_w + 32 * kk) = xfrag[kk];
      if (FF2_PRESCAN && !SCAN_EARLY) {
        const u32x4 xw = __builtin_bit_cast(u32x4, xfrag[kk]);
#pragma unroll
        for (int c = 0; c < 4; ++c) mz = __builtin_elementwise_min(mz, __builtin_bit_cast(us2_t, xw[c] ^ MASK_BITS2));
      }
    }
    if (FF2_PRESCAN) {
      const bool wany = SCAN_EARLY ? wany_early : __builtin_amdgcn_ballot_w64((mz[0] == 0) | (mz[1] == 0)) != 0ull;
      if (lane == 0) mflag[w] = wany ? 1u : 0u;
    }
    __syncthreads();
    if (FF2_PRESCAN) { const u32x4 f = *reinterpret_cast<const u32x4*>(mflag); blk_masked = (f[0] | f[1] | f[2] | f[3]) != 0u; }
  } else
  if (X_VIA_LDS && vec_ok) {     // the wave's 32 x rows are one contiguous 24 KiB block: 24 coalesced 16-byte loads per lane
    const char* xblk = reinterpret_cast<const char*>(reinterpret_cast<const T*>(a.x) + m0 * FF_D);
    us2_t mz = {0xFFFFu, 0xFFFFu};
#pragma unroll
    for (int p0 = 0; p0 < 24; p0 += 4) {      // (four in flight, not eight: the registers are the accumulators' at this point)
      u32x4 xr[4];
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) xr[pc] = *reinterpret_cast<const u32x4*>(xblk + (p0 + pc) * 1024 + lane * 16);
#pragma unroll
      for (int pc = 0; pc < 4; ++pc) {
        const int off = (p0 + pc) * 1024 + lane * 16, r = off / (FF_D * 2), cb = off - r * (FF_D * 2);
        *reinterpret_cast<u32x4*>(stg + (32 * w + r) * FF_DXH_PITCH + cb) = xr[pc];
        if (FF2_PRESCAN) {
#pragma unroll
          for (int c = 0; c < 4; ++c) mz = __builtin_elementwise_min(mz, __builtin_bit_cast(us2_t, xr[pc][c] ^ MASK_BITS2));
        }
      }
    }
    if (FF2_PRESCAN) {
      const bool wany = __builtin_amdgcn_ballot_w64((mz[0] == 0) | (mz[1] == 0)) != 0ull;
      if (lane == 0) mflag[w] = wany ? 1u : 0u;
    }
    __syncthreads();
    if (FF2_PRESCAN) { const u32x4 f = *reinterpret_cast<const u32x4*>(mflag); blk_masked = (f[0] | f[1] | f[2] | f[3]) != 0u; }
  }
  if (STAMP) clk_e1 = __builtin_amdgcn_s_memtime();      // x staged (and published)
  // (vec_ok is a run-time, block-uniform flag: tested inside the loops it ends a basic block per group of 16 elements -- every group waits
  // for its own LDS reads -- and has the scalar path's 16 column indices computed ahead of each branch.  The no-mask instantiation knows it
  // is true, so its 12 groups are ONE basic block; the other instantiation keeps the test (three instantiations -- vec_ok as a tag too --
  // cost 32 spilled accumulators at the head of the epilogue).)
  auto residual = [&](auto notest_tag) {
  constexpr bool NOTEST = decltype(notest_tag)::value;      // (implies vec_ok: the scan that clears a block runs only on the vector path)
  // x of group (mb, dtl) out of the staging image.  The groups are software-pipelined by hand: group g + 1's four 8-byte reads are issued
  // before group g's arithmetic and a scheduling barrier closes each group -- left to itself in one basic block hipcc hoists all 48 reads
  // and spills 55 registers; fenced per group without the look-ahead every group waits for its own reads.
  constexpr bool FENCE = NOTEST && X_VIA_LDS;                             // a scheduling barrier between the groups
  constexpr bool PIPE = FENCE && std::is_same<T, bf16_t>::value;          // ... and the look-ahead (fp16 activations: 3 spilled registers with it)
  Tx4 xq[2][4];
  auto xload = [&](int g, Tx4 (&dst)[4]) {
    const char* sr = stg + (32 * (g / 3) + arow) * FF_DXH_PITCH + (96 * w + 32 * (g % 3) + 4 * ah) * 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) dst[k] = *reinterpret_cast<const Tx4*>(sr + 16 * k);
  };
  if constexpr (PIPE) xload(0, xq[0]);
#pragma unroll
  for (int mb = 0; mb < 4; ++mb) {
    const int64_t grow = (int64_t)wg * FF_BM + 32 * mb + arow;   // global activation row of this lane in row block mb
    const bool ok = !PAD || grow < a.M;
    const float rmask = ok ? 1.f : 0.f;
    const T* xrow = reinterpret_cast<const T*>(a.x) + (ok ? grow : a.M - 1) * a.d;
    char* srow = stg + (32 * mb + arow) * FF_DXH_PITCH;
#pragma unroll
    for (int dtl = 0; dtl < 3; ++dtl) {
      const int dbase = 96 * w + 32 * dtl + 4 * ah;
      if (NOTEST || vec_ok) {
        const int g = 3 * mb + dtl;
        if constexpr (FENCE) {
          if (g > 0) __builtin_amdgcn_sched_barrier(0);
          if (PIPE && g + 1 < 12) xload(g + 1, xq[(g + 1) & 1]);
        }
        Tx4 xv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          xv[k] = PIPE ? xq[g & 1][k]
                       : (X_VIA_LDS ? *reinterpret_cast<const Tx4*>(srow + (dbase + 8 * k) * 2) : *reinterpret_cast<const Tx4*>(xrow + dbase + 8 * k));
#ifndef FF2_EPI_V1
        if (!PAD) {
          // The lean form (round 4; round 5: which of its two halves runs is decided per BLOCK while x is staged, see X_FROM_FRAGS above).
          // Without masked entries: e = bf16(x_hat) - x, plain += e e (one fma), dx_hat = bf16(2 e) -- 6 vector instructions per element
          // instead of round 3's 13 (-DFF2_EPI_V1).  With: the same plus four selects / adds per element.  Both keep plain and the masked
          // entries' squared error (msq) and leave the masked MSE numerator as plain - msq.
          // (FF2_KO: knock-outs of this phase for the stamp builds of round 5 -- results become wrong: 1 = no squared-error chain,
          //  2 = no stores of dx_hat into the staging image, 4 = no -1.0 test, 8 = x taken as zero instead of read from LDS)
          if constexpr (NOTEST && X_VIA_LDS) {      // the block holds no masked entry (scanned while x was staged): no test, no branch
#if FF2_EPI_PACKED
            if constexpr (std::is_same<T, bf16_t>::value) {
              // The same in PAIRS (switch, measured in round 5: profiles/r05_fwd_epilogue.txt): one v_cvt_pk rounds both accumulators, two
              // shifts each bring x_hat and x to fp32, e = x_hat - x, plain += e e and 2 e as ONE packed instruction each (inline asm:
              // hipcc scalarises them), one v_cvt_pk for the output.  The squared-error sum runs in two chains (even / odd elements):
              // fp32 round-off against the element-wise form.
              uint2 ow[4];
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                const uint2 xw = __builtin_bit_cast(uint2, xv[k]);
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                  const unsigned xd = pp ? xw.y : xw.x;
                  const unsigned rb = __builtin_bit_cast(unsigned, __builtin_convertvector(
                      f32x2_t{acc[4 * dtl + mb][4 * k + 2 * pp], acc[4 * dtl + mb][4 * k + 2 * pp + 1]}, bf16x2));
                  const f32x2_t rf = {__uint_as_float(rb << 16), __uint_as_float(rb & 0xFFFF0000u)};
                  const f32x2_t xf = {__uint_as_float(xd << 16), __uint_as_float(xd & 0xFFFF0000u)};
                  f32x2_t e2, d2;
                  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(e2) : "v"(rf), "v"(xf));
                  asm("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(plain2) : "v"(e2));
                  asm("v_pk_add_f32 %0, %1, %1" : "=v"(d2) : "v"(e2));
                  const unsigned ob = __builtin_bit_cast(unsigned, __builtin_convertvector(d2, bf16x2));
                  if (pp) ow[k].y = ob; else ow[k].x = ob;
                }
              }
#pragma unroll
              for (int k = 0; k < 4; ++k) *reinterpret_cast<uint2*>(srow + (dbase + 8 * k) * 2) = ow[k];
              continue;
            }
#endif
            bf16x4 o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float xf = (FF2_KO & 8) ? 0.0f : (float)xv[k][q];
                const float e = bf16_round(acc[4 * dtl + mb][4 * k + q]) - xf;
                if (!(FF2_KO & 1)) plain = __builtin_fmaf(e, e, plain);
                o[k][q] = (bf16_t)(e + e);
              }
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<bf16x4*>(srow + (dbase + 8 * k) * 2) = o[k];
            continue;
          }
#ifndef FF2_MASK_BRANCHY
          // A block WITH masked entries (real padding; or 16-bit data that did not come through the loader's guard, host_convert.c:5-7,
          // and hits -1.0 by rounding: N(0,1) rounded to bf16 does so in 0.14 % of its entries, i.e. in three of four 16 x 64 groups):
          // selects, no branch -- four more instructions per element than the form above.  Round 4's form (-DFF2_MASK_BRANCHY) asked the
          // wave per group of 16 whether any lane met a -1.0 and then took the entries out one by one under exec masks: right when that is
          // rare, 17.7 k cycles for this phase on the bench's own data before its generator got the loader's guard.
          bf16x4 o[4];
          unsigned nm_i = 0u;
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float xf = (FF2_KO & 8) ? 0.0f : (float)xv[k][q];
              const float e = bf16_round(acc[4 * dtl + mb][4 * k + q]) - xf;
              const bool mk = !(FF2_KO & 4) && (xf == -1.0f);
              const float em = mk ? e : 0.0f, ek = mk ? 0.0f : e;
              if (!(FF2_KO & 1)) plain = __builtin_fmaf(e, e, plain);
              msq = __builtin_fmaf(em, em, msq);
              nm_i += mk ? 1u : 0u;
              o[k][q] = (bf16_t)(ek + ek);
            }
          nmask += (float)nm_i;
#else
          bool any_m = false;
          bf16x4 o[4];
          float e[16];
#pragma unroll
          for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float xf = (FF2_KO & 8) ? 0.0f : (float)xv[k][q];
              e[4 * k + q] = bf16_round(acc[4 * dtl + mb][4 * k + q]) - xf;
              if (!(FF2_KO & 1)) plain = __builtin_fmaf(e[4 * k + q], e[4 * k + q], plain);
              if (!(FF2_KO & 4)) any_m |= (xf == -1.0f);
              o[k][q] = (bf16_t)(e[4 * k + q] + e[4 * k + q]);
            }
          if (__builtin_amdgcn_ballot_w64(any_m) != 0ull) {        // wave-uniform, almost never taken
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
              for (int q = 0; q < 4; ++q)
                if ((float)xv[k][q] == -1.0f) {
                  nmask += 1.0f;
                  msq = __builtin_fmaf(e[4 * k + q], e[4 * k + q], msq);
                  o[k][q] = (bf16_t)0.0f;
                }
          }
#endif
          if (FF2_KO & 2) {        // (keep the values alive without the LDS stores)
#pragma unroll
            for (int k = 0; k < 4; ++k) asm volatile("" :: "v"(o[k]));
            continue;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) *reinterpret_cast<bf16x4*>(srow + (dbase + 8 * k) * 2) = o[k];
          continue;
        }
#endif
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          bf16x4 o;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float xf = (float)xv[k][q];
            const float e = bf16_round(acc[4 * dtl + mb][4 * k + q]) - xf;
            const float e2 = e * e;
            plain += rmask * e2;
            const float keep = (xf != -1.0f) ? rmask : 0.f;
            nmask += rmask - keep;
            sq += keep * e2;
            o[q] = (bf16_t)(keep * (e * 2.0f));
          }
          *reinterpret_cast<bf16x4*>(srow + (dbase + 8 * k) * 2) = o;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          bf16x4 o;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int dd = dbase + 8 * k + q;
            const bool valid = dd < a.d;
            const float xf = (float)xrow[valid ? dd : a.d - 1];
            const float e = bf16_round(acc[4 * dtl + mb][4 * k + q]) - xf;
            const float e2 = valid ? e * e : 0.f;
            plain += rmask * e2;
            const float keep = (valid && xf != -1.0f) ? rmask : 0.f;
            nmask += (valid ? rmask : 0.f) - keep;
            sq += keep * e2;
            o[q] = (bf16_t)(keep * (e * 2.0f));
          }
          *reinterpret_cast<bf16x4*>(srow + (dbase + 8 * k) * 2) = o;
        }
      }
    }
  }
  };
  if (X_VIA_LDS && FF2_PRESCAN && vec_ok && !blk_masked) residual(std::true_type{}); else residual(std::false_type{});
  if (STAMP) clk_e2 = __builtin_amdgcn_s_memtime();      // residual arithmetic done, dx_hat in the staging image (not yet published)
  __syncthreads();
  {   // the wave's 32 rows of dx_hat are one contiguous 24 KiB block: 24 fully coalesced 16-byte stores per lane
    char* gblk = reinterpret_cast<char*>(a.dxh + m0 * FF_D);
#pragma unroll
    for (int pc = 0; pc < 24; ++pc) {
      const int off = pc * 1024 + lane * 16, r = off / (FF_D * 2), cb = off - r * (FF_D * 2);
      *reinterpret_cast<u32x4*>(gblk + off) = *reinterpret_cast<const u32x4*>(stg + (32 * w + r) * FF_DXH_PITCH + cb);
    }
  }
  plain += plain2[0] + plain2[1];
#ifndef FF2_TAIL_V1
  // Round 5: the four block sums in ONE exchange -- wave sums, one LDS-only barrier, thread 0 adds the four waves' values in
  // block_sum_256's order (bit-identical) -- through the bias ring (idle since the tile loop; the staging image above is still being
  // read by the dx_hat stores).  Round 4's tail stood behind a __syncthreads (= a drain of the 24 dx_hat stores per lane) and took four
  // block sums of two more such barriers each: the stamps count the epilogue's cycles one for one (-DFF2_TAIL_V1 = that form).
#ifndef FF2_EPI_V1
  if (!PAD && vec_ok) sq = plain - msq;          // (vec_ok is block-uniform)
#endif
  float* red = reinterpret_cast<float*>(smem + FF_FIXED_LDS);
  {
    const float v0 = wave_sum(l1_acc), v1 = wave_sum(sq), v2 = wave_sum(plain), v3 = wave_sum(nmask);
    if (lane == 0) *reinterpret_cast<f32x4*>(red + 4 * w) = f32x4{v0, v1, v2, v3};
  }
  lds_barrier();
  const float l1s = ((red[0] + red[4]) + red[8]) + red[12];
  const float sqs = ((red[1] + red[5]) + red[9]) + red[13];
  const float pls = ((red[2] + red[6]) + red[10]) + red[14];
  const float nms = ((red[3] + red[7]) + red[11]) + red[15];
#else
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);
  const float l1s = block_sum_256(l1_acc, red);
#ifndef FF2_EPI_V1
  if (!PAD && vec_ok) sq = plain - msq;          // (vec_ok is block-uniform)
#endif
  const float sqs = block_sum_256(sq, red + 8);
  const float pls = block_sum_256(plain, red + 16);
  const float nms = block_sum_256(nmask, red + 24);
#endif
  if (STAMP && lane == 0) {
    unsigned long long* o = a.stamps + ((int64_t)wg * 4 + w) * 8;
    const unsigned long long te = __builtin_amdgcn_s_memtime();
    o[7] = te - clk_k0;
    // epilogue phases (bench.py --dbg 65): [0] loop end -> final half iteration + latent drain, [1] x staging, [2] residual arithmetic;
    // what is left of the epilogue = dx_hat publication + stores + sums: whole - prologue - loop - [0] - [1] - [2]
    o[0] = clk_e0 - clk_t1; o[1] = clk_e1 - clk_e0; o[2] = clk_e2 - clk_e1;
#ifdef FF2_WAITSTAMP
    o[0] = ws_vm; o[1] = ws_bar; o[2] = ws_gap;      // sums over the tile loop (bench.py divides by the iterations)
#endif
  }
  if (t == 0) {
    a.cnt_part[wg] = nms;
    a.l1_part[wg] = l1s;
    a.sq_part[2 * wg] = sqs;
    a.sq_part[2 * wg + 1] = pls;
  }
}
