// Fused forward of the tied-weight L1 SAE for d_model (padded) == 384 on gfx950: the decomposition of fwd_fused2.h (decoder split
// along d, c^T exchanged through the latent staging image) on the OTHER bf16 MFMA shape, v_mfma_f32_16x16x32_bf16.
// Same arithmetic, outputs and arguments as fwd_fused_d384_kernel (reference src/models/l1autoencoder.py:69-95, mse_loss :29-36).
//
// Why: these kernels sit on the chip's power cap, not on an issue or LDS bound (fwd_fused2 took 17 % of the
// cycles out of the tile loop and got 1 % of the time back, the rest went into a lower clock).  What the cap leaves is energy per
// FLOP, and the chip holds a markedly higher clock on the 16x16x32 shape: 1.97 against 1.70 PFLOP/s in the bare loop, +6...9 %
// with LDS operand streams and VALU in the gaps (tools/mfma_clock_probe.hip, profiles/r01_mfma_clock_probe.jsonl).  The price is
// twice the MFMA instructions (96 per dictionary tile of 32 columns); fwd_fused2's decomposition pays part of it back, because a
// fragment read from LDS now feeds more MFMAs: an encoder W^T fragment 2, a decoder W fragment 8, a c^T fragment 6.
//
// Layouts of v_mfma_f32_16x16x32_bf16 (lq = lane & 15, kg = lane >> 4):  A[16 x 32]: row lq, k = 8 kg .. 8 kg + 7;
// B[32 x 16]: column lq, k = 8 kg .. + 7;  D[16 x 16]: column lq, rows 4 kg .. 4 kg + 3.  Per wave (row block of 32 rows):
//   encoder  S^T[32 n x 32 m] = 2 x 2 tiles (nh, mh), 12 k-steps of 32:  A = W^T rows (ds_read_b128 of the tile image),
//            B = x fragments (24 x 16 bytes per lane, in registers for the whole kernel);
//   latent   lane holds c for m = 16 mh + lq, n = 16 nh + 4 kg + (0..3): four consecutive columns -> 8-byte staging writes in
//            NATURAL column order (row m of the pair buffer = 64 latents of tiles 2t, 2t+1; 16-byte chunk index XOR (m & 7));
//   decoder  x_hat^T[96 d x 128 m] (the wave's d-slice, all rows of the workgroup) = 6 x 8 tiles (dt, mt), ONE k-step (32 n):
//            A = W[16 d x 32 n] by transposed reads of the W^T image (two ds_read_b64_tr_b16 per fragment), B = c^T fragment
//            (ds_read_b128 of the staging row: the lane's 8 consecutive latents).
// Pipeline: 96 MFMA slots per tile, even = decoder of tile j-1, odd = encoder of tile j+1; bias / ReLU / L1 of tile j in the
// gaps; one barrier per tile (slot 80) publishes c(j) and the DMA'd W^T tile j+2, as in fwd_fused2.h.
//
// STATUS (round 3): an experiment, not part of the product build.  Wired into engine.hip as a third forward variant it passes
// tests/test_engine_gpu.py (24 tests) -- the layouts and the pipeline are right -- but as hipcc compiles it (it needs
// -mllvm -pragma-unroll-threshold=400000 to unroll the 96-slot body at all) it keeps 256 + 256 registers, spills 85 more and
// reloads some of them from scratch inside the tile loop: 6395 cycles per tile at 2.42 GHz against fwd_fused2's 2088 at 2.03 GHz,
// forward 0.515 against 0.253 ms.  The shape question stays open until the register allocation is done by hand (drop the
// lo / hi pointer pairs, keep S in AGPRs); to try it again: copy next to fwd_fused2.h, include it from engine.hip and add the
// variant to the FREUD_FWD switch.
#pragma once
#include "fwd_fused2.h"

template <typename T, bool PAD, bool STAMP = false>
__global__ __launch_bounds__(256, 1) void fwd_fused3_d384_kernel(FwdFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int lq = lane & 15, kg = lane >> 4;
  const int wg = blockIdx.x + a.block_offset;
  const int64_t m0 = (int64_t)wg * FF_BM + 32 * w;             // first row of this wave's row block
  char* cst = smem + FF_RING_BYTES;                             // two pair buffers [128 rows][128 B]
  float* bias_s = reinterpret_cast<float*>(smem + FF_FIXED_LDS);
  unsigned long long clk_k0 = 0, clk_t0 = 0, clk_r0 = 0;
  if (STAMP) clk_k0 = __builtin_amdgcn_s_memtime();

  // x fragments: B[k = d][col = m]: lane (lq, kg) holds xb[m0 + 16 mh + lq][32 ks + 8 kg .. + 8]   (index 12 mh + ks)
  bf16x8 xfrag[24];
  bool row_ok[2];
#pragma unroll
  for (int mh = 0; mh < 2; ++mh) {
    const int64_t mrow = m0 + 16 * mh + lq;
    row_ok[mh] = mrow < a.M;
    const bf16_t* xp = a.xb + mrow * FF_D + 8 * kg;
#pragma unroll
    for (int ks = 0; ks < 12; ++ks) xfrag[12 * mh + ks] = *reinterpret_cast<const bf16x8*>(xp + 32 * ks);
  }
  if (t < 2 * FF_BN) bias_s[t] = a.bias[t];
  for (int i = t; i < FF_WT_BYTES / 16; i += 256) reinterpret_cast<u32x4*>(smem + 3 * FF_WT_BYTES)[i] = u32x4{0u, 0u, 0u, 0u};
  for (int i = t; i < FF_CST_BYTES / 16; i += 256) reinterpret_cast<u32x4*>(cst)[i] = u32x4{0u, 0u, 0u, 0u};

  f32x4 acc[48];                        // acc[8 dt + mt]: rows d = 96 w + 16 dt + 4 kg + r, column m = 16 mt + lq
#pragma unroll
  for (int i = 0; i < 48; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

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
  const int last = a.ntiles - 1;
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int p = 0; p < 3; ++p) dma_pair(p, q <= last ? q : last, q);
#pragma unroll
  for (int kk = 0; kk < 24; ++kk) asm volatile("" : "+v"(xfrag[kk]));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- loop-invariant per-lane LDS offsets
  // encoder A fragment (nh, ks): row n = 16 nh + lq of the W^T image, 16-byte chunk 4 ks + kg of its 768-byte row
  //   = sub-tile (4 ks + kg) >> 4, chunk (4 ks + kg) & 15: ks = 4 s + u -> sub-tile s, chunk 4 u + kg: eoff[nh][u] + 8192 s
  int eoff[2][4];
#pragma unroll
  for (int nh = 0; nh < 2; ++nh)
#pragma unroll
    for (int u = 0; u < 4; ++u) eoff[nh][u] = dual_off(16 * nh + lq, 4 * u + kg);
  // decoder A fragment dt: W rows d = 96 w + 16 dt + lq, k = n = 8 kg .. + 7: 16-lane group kg reads the 8 x 16 block
  // (rows 8 kg .. + 7, columns d0 .. d0 + 15) of the image with two transposing 8-byte reads (rows + 0..3 / + 4..7)
  int aoff0[6], aoff1[6];
  {
    const int q = lq >> 2, p = lq & 3;
#pragma unroll
    for (int dt = 0; dt < 6; ++dt) {
      const int d0 = 96 * w + 16 * dt, sub = d0 >> 7, colw = (d0 & 127) + 4 * p;
      aoff0[dt] = sub * 8192 + dual_off(8 * kg + q, colw >> 3) + (colw & 7) * 2;
      aoff1[dt] = sub * 8192 + dual_off(8 * kg + 4 + q, colw >> 3) + (colw & 7) * 2;
    }
  }
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  auto tr_pair = [&](const char* p0, const char* p1) -> bf16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, p1));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  };
  // staging: row m (128 B: tiles 2t | 2t+1), 16-byte chunk c8 = 4 hf + (n >> 3), chunk index XOR (m & 7)
  //   reader (c^T fragment mt, tile half hf): row 16 mt + lq, chunk 4 hf + kg                      -> boff[hf] + 2048 mt
  //   writer (tile (nh, mh), half hf): row 32 w + 16 mh + lq, chunk 4 hf + 2 nh + (kg >> 1), + 8 (kg & 1) bytes
  int boff[2], woff[2][2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    boff[hf] = lq * 128 + (((4 * hf + kg) ^ (lq & 7)) << 4);
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) woff[hf][nh] = (32 * w + lq) * 128 + (((4 * hf + 2 * nh + (kg >> 1)) ^ (lq & 7)) << 4) + 8 * (kg & 1);
  }
  const int wrow = w * 4096;
  const int drow_l = lane >> 3, dch = lane & 7;
  const int doff = drow_l * 128 + ((dch ^ drow_l) << 4);           // drain: row 8 p + lane / 8 of the row block, chunk lane % 8
  bf16_t* cdrain = a.c + (m0 + drow_l) * a.n_p + dch * 8;

  float l1_acc = 0.f;

  // ---- S(0): encoder product of tile 0 (outside the pipeline).  S[2 nh + mh]
  f32x4 SA[4], SB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) SA[i] = SB[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ks = 0; ks < 12; ++ks)
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      const bf16x8 fa = *reinterpret_cast<const bf16x8*>(smem + (ks >> 2) * 8192 + eoff[nh][ks & 3]);
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
        SA[2 * nh + mh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, xfrag[12 * mh + ks], SA[2 * nh + mh], 0, 0, 0);
    }

  // ---- pointers: ring slots 0,1 from *_lo, slots 2,3 from *_hi (ds_read immediates are 16 bit)
  const char *elo[2][4], *ehi[2][4], *a0lo[6], *a1lo[6], *a0hi[6], *a1hi[6], *bptr[2];
#pragma unroll
  for (int nh = 0; nh < 2; ++nh)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      elo[nh][u] = smem + eoff[nh][u];
      ehi[nh][u] = smem + 2 * FF_WT_BYTES + eoff[nh][u];
    }
#pragma unroll
  for (int dt = 0; dt < 6; ++dt) {
    a0lo[dt] = smem + aoff0[dt];
    a1lo[dt] = smem + aoff1[dt];
    a0hi[dt] = smem + 2 * FF_WT_BYTES + aoff0[dt];
    a1hi[dt] = smem + 2 * FF_WT_BYTES + aoff1[dt];
  }
  bptr[0] = cst + boff[0];
  bptr[1] = cst + boff[1];
  auto dec_a = [&](int slot, int dt) -> bf16x8 {
    const int off = (slot & 1) * FF_WT_BYTES;
    return tr_pair((slot < 2 ? a0lo : a0hi)[dt] + off, (slot < 2 ? a1lo : a1hi)[dt] + off);
  };
  auto enc_a = [&](int slot, int af) -> bf16x8 {          // encoder fragment af = 2 ks + nh
    const int ks = af >> 1, nh = af & 1;
    const int off = (slot & 1) * FF_WT_BYTES + (ks >> 2) * 8192;
    return *reinterpret_cast<const bf16x8*>((slot < 2 ? elo : ehi)[nh][ks & 3] + off);
  };
  auto dec_b = [&](int pb, int hf, int mt) -> bf16x8 { return *reinterpret_cast<const bf16x8*>(bptr[hf] + pb * 16384 + mt * 2048); };

  bf16x8 ering[4], Afr[6], Bq[2];
  // decoder of "tile -1": zeros (cleared staging) times the cleared ring slot 3; the first three encoder fragments of tile 1
#pragma unroll
  for (int dt = 0; dt < 6; ++dt) Afr[dt] = dec_a(3, dt);
  Bq[0] = dec_b(1, 1, 0);
#pragma unroll
  for (int af = 0; af < 3; ++af) ering[af] = enc_a(1, af);

  u32x4 dr[2];
  bf16_t* const dummy_line = a.c + (int64_t)a.c_rows * a.n_p + lane * 8;
  auto body = [&](auto ph_tag, int j) {
    constexpr int PH = decltype(ph_tag)::value;          // == j % 4
    constexpr int SLOT_DN = PH, SLOT_E = (PH + 1) & 3, SLOT_EN = (PH + 2) & 3, SLOT_DMA = (PH + 3) & 3;
    constexpr int PB_W = (PH >> 1) & 1, HF_W = PH & 1;               // where c(j) goes
    constexpr int PB_D = (((PH + 3) & 3) >> 1) & 1, HF_D = (PH + 3) & 1;   // where c(j-1) is read (this iteration's decoder)
    f32x4(&Scur)[4] = (PH & 1) ? SB : SA;
    f32x4(&Snxt)[4] = (PH & 1) ? SA : SB;
    const int jt = j + 3 <= last ? j + 3 : last;
    const float* bj = bias_s + (j & (FF_BIAS_RING_TILES - 1)) * FF_BN;
    const char* cst_r = cst + (PB_W ^ 1) * 16384 + wrow;
    bf16_t* dst_pair = j >= 2 ? cdrain + 64 * ((j >> 1) - 1) : dummy_line;
    const int64_t dst_rstride = j >= 2 ? (int64_t)a.n_p : 0;
    f32x4 bq[2];
    float l1_it = 0.f;
    bf16x4 cw;
    f32x4 S[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) S[i] = Scur[i];

    // 96 MFMA slots of 16 cycles: EVEN slot i = decoder MFMA m = i / 2 (mt = m / 6, dt = m % 6), ODD slot i = encoder MFMA
    // e = i / 2 (ks = e / 4, nh = (e / 2) & 1, mh = e & 1).  Fragments are requested ~12 slots ahead.
#pragma unroll
    for (int i = 0; i < 96; ++i) {
      // ---- fragment prefetch
      // encoder fragment af (used by slots 4 af + 1, 4 af + 3) at gap 4 af - 11: af >= 3 of this tile, af < 3 of the next one
      if (i >= 1 && i <= 81 && (i - 1) % 4 == 0) ering[((i + 11) / 4) & 3] = enc_a(SLOT_E, (i + 11) / 4);
      if (i == 85 || i == 89 || i == 93) ering[(i - 85) / 4] = enc_a(SLOT_EN, (i - 85) / 4);
      // c^T fragment mt (used by slots 12 mt .. 12 mt + 10) at gap 12 (mt - 1); mt = 0 of the next tile after the barrier
      if (i <= 72 && i % 12 == 0) Bq[(i / 12 + 1) & 1] = dec_b(PB_D, HF_D, i / 12 + 1);
      if (i == 84) Bq[0] = dec_b(PB_W, HF_W, 0);
      // decoder W fragment dt of the NEXT tile right after its last use (decoder MFMA 42 + dt = slot 84 + 2 dt)
      if (i >= 85 && (i - 85) % 2 == 0) Afr[(i - 85) / 2] = dec_a(SLOT_DN, (i - 85) / 2);
      if (i < 2) bq[i] = *reinterpret_cast<const f32x4*>(bj + 16 * i + 4 * kg);      // bias of n = 16 nh + 4 kg + (0..3), nh = i
      if (i == 80) {
        // <= 1 VMEM operation outstanding (this iteration's first latent store): the six DMA pieces of tile j+2 are done;
        // <= 2 LDS operations outstanding (four fragment reads follow the last staging write of gap 66): that write is done
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(1) lgkmcnt(2)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      if (i == 81 || i == 83 || i == 87) dma_pair(i == 81 ? 0 : (i == 83 ? 1 : 2), jt, SLOT_DMA);
      if ((PH & 1) == 0 && i == 91 && w == 0) {
        const int jb = j + 2 <= a.ntiles - 2 ? j + 2 : a.ntiles - 2;
        glds4(a.bias + (int64_t)jb * FF_BN, (unsigned)(lane * 4),
              (unsigned)__builtin_amdgcn_readfirstlane((int)(smem_base + FF_FIXED_LDS + ((j + 2) & (FF_BIAS_RING_TILES - 1)) * FF_BN * 4)));
      }
      // latent element e = 4 (2 nh + mh) + r at gap 6 + 4 e: rounded to bf16 BEFORE the fp32 bias add, as CPU autocast does
      if (i >= 6 && i <= 66 && (i - 6) % 4 == 0) {
        const int e = (i - 6) / 4, tl = e >> 2, nh = tl >> 1, mh = tl & 1, r = e & 3;
        float cv = fmaxf(bf16_round(S[tl][r]) + bq[nh][r], 0.f);
        if (PAD) cv = row_ok[mh] ? cv : 0.f;
        l1_it += cv;
        cw[r] = (bf16_t)cv;
        if (r == 3) *reinterpret_cast<bf16x4*>(cst + PB_W * 16384 + woff[HF_W][nh] + mh * 2048) = cw;
      }
      // two full-line pieces of the finished pair per iteration: LDS read in one gap, global store 16 gaps later
      if (i == 57 || i == 65) dr[i == 65] = *reinterpret_cast<const u32x4*>(cst_r + (2 * (PH & 1) + (i == 65)) * 1024 + doff);
      if (i == 73 || i == 82) {
        const int p = 2 * (PH & 1) + (i == 82);
        __builtin_nontemporal_store(dr[i == 82], reinterpret_cast<u32x4*>(dst_pair + (int64_t)(8 * p) * dst_rstride));
      }
      __builtin_amdgcn_sched_barrier(0);
      if ((i & 1) == 0) {
        const int m = i / 2, mt = m / 6, dt = m % 6;
        acc[8 * dt + mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Afr[dt], Bq[mt & 1], acc[8 * dt + mt], 0, 0, 0);
      } else {
        const int e = i / 2, ks = e >> 2, nh = (e >> 1) & 1, mh = e & 1;
        const f32x4 c0 = ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : Snxt[2 * nh + mh];
        Snxt[2 * nh + mh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ering[(e >> 1) & 3], xfrag[12 * mh + ks], c0, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    l1_acc += l1_it;
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
    if (lane == 0) {
      unsigned long long* o = a.stamps + ((int64_t)wg * 4 + w) * 8;
      o[0] = o[1] = o[2] = 0; o[3] = (unsigned long long)a.ntiles; o[4] = t1 - clk_t0; o[5] = r1 - clk_r0; o[6] = clk_t0 - clk_k0;
    }
  }
  // ---- final half iteration: decoder of the last tile (slot 3, pair buffer 1, half 1); its W fragments and its first c^T
  // fragment were requested at the end of the last iteration (after its barrier)
#pragma unroll
  for (int m = 0; m < 48; ++m) {
    if (m % 6 == 0 && m / 6 + 1 <= 7) Bq[(m / 6 + 1) & 1] = dec_b(1, 1, m / 6 + 1);
    __builtin_amdgcn_sched_barrier(0);
    const int mt = m / 6, dt = m % 6;
    acc[8 * dt + mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Afr[dt], Bq[mt & 1], acc[8 * dt + mt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  // drain the last pair of latent tiles (this wave's rows; published by the last iteration's barrier)
  {
    const char* cst_r = cst + (((a.ntiles >> 1) - 1) & 1) * 16384 + wrow;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(cst_r + p * 1024 + doff),
                                  reinterpret_cast<u32x4*>(cdrain + (int64_t)(8 * p) * a.n_p + 64 * ((a.ntiles >> 1) - 1)));
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- epilogue: acc[8 dt + mt][r] <-> d = 96 w + 16 dt + 4 kg + r, row 128 wg + 16 mt + lq.  The [128 x 384] x block is staged
  // in the (now idle) rings, transformed in place into dx_hat by the four waves, and leaves as coalesced 16-byte stores.
  float sq = 0.f, plain = 0.f, nmask = 0.f;
  typedef __attribute__((ext_vector_type(4))) T Tx4;
  const bool vec_ok = (a.d == FF_D) && ((reinterpret_cast<uintptr_t>(a.x) & (sizeof(T) * 4 - 1)) == 0);
  constexpr int FF_DXH_PITCH = FF_D * 2 + 16;                    // 784 B
  static_assert(128 * FF_DXH_PITCH <= FF_FIXED_LDS, "the dx_hat staging must fit the idle LDS");
  char* stg = smem;
  constexpr bool X_VIA_LDS = !PAD && sizeof(T) == 2;
  if (X_VIA_LDS && vec_ok) {
    const char* xblk = reinterpret_cast<const char*>(reinterpret_cast<const T*>(a.x) + m0 * FF_D);
#pragma unroll
    for (int p0 = 0; p0 < 24; p0 += 8) {
      u32x4 xr[8];
#pragma unroll
      for (int pc = 0; pc < 8; ++pc) xr[pc] = *reinterpret_cast<const u32x4*>(xblk + (p0 + pc) * 1024 + lane * 16);
#pragma unroll
      for (int pc = 0; pc < 8; ++pc) {
        const int off = (p0 + pc) * 1024 + lane * 16, r = off / (FF_D * 2), cb = off - r * (FF_D * 2);
        *reinterpret_cast<u32x4*>(stg + (32 * w + r) * FF_DXH_PITCH + cb) = xr[pc];
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int mt = 0; mt < 8; ++mt) {
    const int64_t grow = (int64_t)wg * FF_BM + 16 * mt + lq;
    const bool ok = !PAD || grow < a.M;
    const float rmask = ok ? 1.f : 0.f;
    const T* xrow = reinterpret_cast<const T*>(a.x) + (ok ? grow : a.M - 1) * a.d;
    char* srow = stg + (16 * mt + lq) * FF_DXH_PITCH;
#pragma unroll
    for (int dt = 0; dt < 6; ++dt) {
      const int dbase = 96 * w + 16 * dt + 4 * kg;
      bf16x4 o;
      if (vec_ok) {
        const Tx4 xv = X_VIA_LDS ? *reinterpret_cast<const Tx4*>(srow + dbase * 2) : *reinterpret_cast<const Tx4*>(xrow + dbase);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float xf = (float)xv[q];
          const float e = bf16_round(acc[8 * dt + mt][q]) - xf;
          const float e2 = e * e;
          plain += rmask * e2;
          const float keep = (xf != -1.0f) ? rmask : 0.f;
          nmask += rmask - keep;
          sq += keep * e2;
          o[q] = (bf16_t)(keep * (e * 2.0f));
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int dd = dbase + q;
          const bool valid = dd < a.d;
          const float xf = (float)xrow[valid ? dd : a.d - 1];
          const float e = bf16_round(acc[8 * dt + mt][q]) - xf;
          const float e2 = valid ? e * e : 0.f;
          plain += rmask * e2;
          const float keep = (valid && xf != -1.0f) ? rmask : 0.f;
          nmask += (valid ? rmask : 0.f) - keep;
          sq += keep * e2;
          o[q] = (bf16_t)(keep * (e * 2.0f));
        }
      }
      *reinterpret_cast<bf16x4*>(srow + dbase * 2) = o;
    }
  }
  __syncthreads();
  {
    char* gblk = reinterpret_cast<char*>(a.dxh + m0 * FF_D);
#pragma unroll
    for (int pc = 0; pc < 24; ++pc) {
      const int off = pc * 1024 + lane * 16, r = off / (FF_D * 2), cb = off - r * (FF_D * 2);
      *reinterpret_cast<u32x4*>(gblk + off) = *reinterpret_cast<const u32x4*>(stg + (32 * w + r) * FF_DXH_PITCH + cb);
    }
  }
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);
  const float l1s = block_sum_256(l1_acc, red);
  const float sqs = block_sum_256(sq, red + 8);
  const float pls = block_sum_256(plain, red + 16);
  const float nms = block_sum_256(nmask, red + 24);
  if (STAMP && lane == 0) a.stamps[((int64_t)wg * 4 + w) * 8 + 7] = __builtin_amdgcn_s_memtime() - clk_k0;
  if (t == 0) {
    a.cnt_part[wg] = nms;
    a.l1_part[wg] = l1s;
    a.sq_part[2 * wg] = sqs;
    a.sq_part[2 * wg + 1] = pls;
  }
}
