// Fused forward of the tied-weight L1 SAE for d_model (padded) == 384 on gfx950.
//
//   c = relu(bf16(x W) + b)          (l1autoencoder.py:74)        -> stored once as bf16 for the backward
//   x_hat = bf16(c W^T)              (l1autoencoder.py:84)
//   l1 = sum c ; masked squared error, UNSCALED dx_hat = bf16(2 (x_hat - x)[x != -1])   (:85-86, :29-36);
//   the factor alpha / count (count = unmasked entries, known only after every workgroup has looked at its x
//   rows) is applied by the backward's epilogue, so no separate counting pass over x is needed.
//
// One workgroup = 4 waves (one per SIMD, whole register file) = 128 activation rows; wave w owns rows
// 32w..32w+31 for the whole dictionary sweep:
//   - its x rows stay in registers as 24 MFMA B-fragments (K = d = 384);
//   - its x_hat^T tile [384 x 32] stays in 192 accumulator registers (12 MFMA tiles);
//   - the dictionary is swept in 32-column tiles.  Per tile, S^T[32 cols x 32 rows] = W^T tile . x^T
//     (24 MFMAs, A = W^T rows from LDS) -> bias + ReLU on the accumulator registers -> those registers ARE the
//     B operand of x_hat^T += W tile . c^T (24 MFMAs, A = W read from the SAME W^T tile image by transposed LDS
//     reads in the accumulator's permuted k order) -- c never round-trips through LDS or HBM between the two GEMMs;
//     it is written to HBM once (staged per wave in LDS into full 128-byte lines) for the backward.
//   - software pipeline across tiles: the encoder MFMAs of tile j+1 run first, with the bias/ReLU/store work of
//     tile j in their gaps, then the decoder MFMAs of tile j.  W tiles arrive by LDS-DMA two / one tile ahead.
#pragma once
#include <type_traits>
// gaps (of the 48 per iteration, after the barrier at gap 12) in which the three DMA pairs of tile j+2 are issued
#ifndef FF_DMA_G0
#define FF_DMA_G0 15
#define FF_DMA_G1 19
#define FF_DMA_G2 23
#endif

#include "bwd_fused.h"

// Timing experiments (tools/build_variant.sh <name> -DFF_KO=<bits>; results become WRONG): 1 = no bias / ReLU / L1 arithmetic,
// 2 = no latent staging / stores, 4 = no LDS-DMA inside the loop, 8 = no decoder fragment reads inside the loop,
// 16 = latent = S cast to bf16 (no rounding-then-bias, no ReLU, no L1 sum: the price of that arithmetic with live data)
#ifndef FF_KO
#define FF_KO 0
#endif

constexpr int FF_D = 384;
constexpr int FF_BM = 128;                        // rows per workgroup
constexpr int FF_BN = 32;                         // dictionary columns per tile
constexpr int FF_WT_BYTES = FF_BN * FF_D * 2;     // 24576: W^T tile [32 n][384 d] (3 dual-use sub-tiles)
constexpr int FF_DEPTH = 4;                       // W^T ring: tile j (decoder), j+1 (encoder), j+2 / j+3 landing
constexpr int FF_RING_BYTES = FF_DEPTH * FF_WT_BYTES;          // 98304
constexpr int FF_CST_BYTES = 4 * 2 * 4096;        // per wave two [32 rows][64 cols] bf16 latent staging buffers
constexpr int FF_FIXED_LDS = FF_RING_BYTES + FF_CST_BYTES;     // + the 1 KiB bias ring
constexpr int FF_BIAS_RING_TILES = 8;                          // bias of tiles j .. j+7 (32 floats each)
constexpr int FF_BIAS_RING_BYTES = FF_BIAS_RING_TILES * FF_BN * 4;
constexpr int FF_LDS_BYTES = FF_FIXED_LDS + FF_BIAS_RING_BYTES;

struct FwdFusedArgs {
  const bf16_t* xb;      // [M_p][384]  bf16 GEMM operand
  const void* x;         // original activations [M][d] (dtype T) for the residual
  const bf16_t* Wt;      // [n_p][384]
  const float* bias;     // [n_p]
  float* cnt_part;       // [M_p/128] masked-entry (x == -1) counts per workgroup
  bf16_t* c;             // [M_p][n_p]
  bf16_t* dxh;           // [M_p][384]
  float* l1_part;        // [M_p/128]
  float* sq_part;        // [M_p/128][2]
  int64_t M;
  int d, n_p, ntiles;    // ntiles = n_p / 32 (even)
  int block_offset;      // first workgroup index of this launch
  unsigned long long* stamps;   // STAMP build only
  int64_t c_rows;        // M_p: a dummy 1-KiB line lives at c[M_p][0..] (engine allocates the slack)
};

// PAD: the workgroup may contain rows >= M (only the last, ragged workgroup is launched with PAD = true).
// STAMP: diagnostic build (bench.py --dbg 65): s_memtime stamps around the halves of an iteration, summed per wave into
// a.stamps[wg][wave][8] (+ loop s_memtime / s_memrealtime deltas in [4], [5]); the production instantiations contain no stamp.
// The encoder bias streams through an 8-tile LDS ring (any dictionary size, e.g. the reference's default expansion_factor
// 32 = 12 288 latents at d = 384; holding all of it in LDS capped n_p at 8160 and was not faster): two tiles (one 256-byte
// LDS-DMA piece, issued by wave 0 right after the W^T pieces of the same iteration, i.e. older than the two latent
// stores the counted vmcnt leaves open) every second iteration, two iterations ahead of their use.
template <typename T, bool PAD, bool STAMP = false>
__global__ __launch_bounds__(256, 1) void fwd_fused_d384_kernel(FwdFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int arow = lane & 31, ah = lane >> 5;
  const int wg = blockIdx.x + a.block_offset;
  const int64_t m0 = (int64_t)wg * FF_BM + 32 * w;             // first row of this wave
  const int64_t mrow = m0 + arow;                               // the row this lane's accumulator column is
  const bool row_ok = mrow < a.M;
  char* cst = smem + FF_RING_BYTES + w * 8192;                  // this wave's latent staging (2 x 4 KB)
  float* bias_s = reinterpret_cast<float*>(smem + FF_FIXED_LDS);
  unsigned long long clk_k0 = 0;
  if (STAMP) clk_k0 = __builtin_amdgcn_s_memtime();

  // x fragments: B[k = d][col = m] -> lane (m = lane&31, h) holds xb[m0 + m][16 kk + 8 h ..+8]
  bf16x8 xfrag[24];
  {
    const bf16_t* xp = a.xb + mrow * FF_D + 8 * ah;
#pragma unroll
    for (int kk = 0; kk < 24; ++kk) xfrag[kk] = *reinterpret_cast<const bf16x8*>(xp + 16 * kk);
  }
  if (t < 2 * FF_BN) bias_s[t] = a.bias[t];                     // tiles 0 and 1; the loop brings tiles j+2, j+3 at even j
  // ring slot 3 is read (times zero) by the first iteration's decoder phase: make it finite
  for (int i = t; i < FF_WT_BYTES / 16; i += 256)
    reinterpret_cast<u32x4*>(smem + 3 * FF_WT_BYTES)[i] = u32x4{0u, 0u, 0u, 0u};

  f32x16 acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // ---- LDS-DMA plan: a W^T tile is 3 sub-tiles [32 rows][256 B] with the dual-use swizzle applied to the source
  // chunk; 24 pieces of 1 KB (4 rows each), 6 per wave, issued as 3 two-piece statements per iteration.
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
  auto dma_pair = [&](int p, int jt, int st) {     // pieces 2p, 2p+1 of W^T tile `jt` into ring slot `st`
    const bf16_t* src = a.Wt + (int64_t)jt * FF_BN * FF_D;
    const unsigned dst = smem_base + st * FF_WT_BYTES;
    glds16_x2(src, src, voff_t[2 * p], voff_t[2 * p + 1], (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + loff_t[2 * p])),
              (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + loff_t[2 * p + 1])));
  };
  const int last = a.ntiles - 1;
#pragma unroll
  for (int q = 0; q < 2; ++q)                       // prologue: W^T tiles 0, 1 into slots 0, 1
#pragma unroll
    for (int p = 0; p < 3; ++p) dma_pair(p, q <= last ? q : last, q);
#pragma unroll
  for (int kk = 0; kk < 24; ++kk) asm volatile("" : "+v"(xfrag[kk]));   // retire hipcc-tracked loads before the loop
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- loop-invariant per-lane LDS offsets
  int roff[8], toff[8];      // row reads (encoder A operand) / transposed reads (decoder A operand) of a W^T tile
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
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  auto tr_pair = [&](const char* p0, const char* p1) -> bf16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, p1));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  };
  // latent staging [32 rows][64 cols] bf16, 128-B rows, 16-B chunk index XORed with (row & 7):
  //   write: this lane's row `arow`, 8 bytes at column 32 (j&1) + 8 k + 4 h
  //   drain: piece p (0..3) = rows 8p..8p+7, lane -> row 8p + lane/8, chunk lane%8 (16 B) -> full 128-B lines
  const int drow_l = lane >> 3, dch = lane & 7;
  bf16_t* cdrain = a.c + (m0 + drow_l) * a.n_p + dch * 8;     // + 8 p rows, + 64 * pair columns

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

  // ---- skewed software pipeline.  Iteration j (0..ntiles-1) issues 48 MFMAs:
  //   i in [ 0,24): decoder of tile j-1:  x_hat^T[dt] += W tile (transposed reads of slot (j-1)%4) . c^T(j-1)
  //   i in [24,48): encoder of tile j+1:  S(j+1)      += W^T tile (row reads of slot (j+1)%4)      . x^T
  // while the bias/ReLU/L1/staging work that turns S(j) into c(j) is spread over the 48 gaps (one latent element per
  // ~3 MFMAs).  A-fragments are requested DIST MFMAs ahead through a 12-deep register ring that is carried ACROSS
  // iterations (the next decoder tile is long resident), so no LDS latency is exposed at iteration boundaries.
  // One barrier per iteration at gap 12: by then the DMA of tile j+1 (issued in iteration j-1) has had a full
  // iteration to land (counted vmcnt); the DMA of tile j+2 is issued after it into the slot whose tile j-2 was last
  // read before this barrier.
  // The loop is unrolled x4 (template parameter PH = j % 4): ring slots, the S / c^T ping-pong registers and the
  // staging halves are then compile-time constants, every LDS address is a loop-invariant base register plus an
  // instruction immediate, and no per-iteration address arithmetic or register copies remain (measured with
  // s_memtime stamps: that loop-top work cost ~300 of ~2560 cycles per iteration).
  // (an 8-deep fragment ring: with the bias ring's extra live values a 12-deep one spills 7 registers to scratch, and 8
  // measured the same at the C2 shape)
  constexpr int RING = 8, DIST = RING - 1;
  // per-lane LDS base pointers: *_lo serves slots 0,1 and *_hi slots 2,3 (ds_read immediates are 16-bit)
  const char *rlo[8], *rhi[8], *tlo[8], *thi[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    rlo[i] = smem + roff[i];
    rhi[i] = smem + 2 * FF_WT_BYTES + roff[i];
    tlo[i] = smem + toff[i];
    thi[i] = smem + 2 * FF_WT_BYTES + toff[i];
  }
  bf16x8 ring[RING];
  f32x16 SA = Sn, SB;          // S(j) lives in SA for even j, SB for odd j
  bf16x8 CA[2], CB[2];         // c^T(j) lives in CA for even j, CB for odd j; c^T(-1) = 0
#pragma unroll
  for (int q = 0; q < 8; ++q) CA[0][q] = CA[1][q] = CB[0][q] = CB[1][q] = (bf16_t)0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) SB[r] = 0.f;
  auto dec_frag = [&](int slot, int i) -> bf16x8 {       // decoder fragment of MFMA i (0..23) from ring slot `slot`
    const int dt = i >> 1, sk = i & 1;
    const int off = (slot & 1) * FF_WT_BYTES + (dt >> 2) * 8192 + sk * 4096;
    const char* b0 = (slot < 2 ? tlo : thi)[2 * (dt & 3)] + off;
    const char* b1 = (slot < 2 ? tlo : thi)[2 * (dt & 3) + 1] + off;
    return tr_pair(b0, b1);
  };
  auto enc_frag = [&](int slot, int kk) -> bf16x8 {      // encoder fragment of k-step kk (0..23)
    const int off = (slot & 1) * FF_WT_BYTES + (kk >> 3) * 8192;
    return *reinterpret_cast<const bf16x8*>((slot < 2 ? rlo : rhi)[kk & 7] + off);
  };
#pragma unroll
  for (int i = 0; i < DIST; ++i) ring[i] = dec_frag(3, i);

  u32x4 dr[2];
  bf16_t* const dummy_line = a.c + (int64_t)a.c_rows * a.n_p + lane * 8;
  unsigned long long tsum[4] = {0, 0, 0, 0};
  auto stamp = [&]() -> unsigned long long {
    unsigned long long tt;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return tt;
  };
  auto body = [&](auto ph_tag, int j) {
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    if (STAMP) t0 = stamp();
    constexpr int PH = decltype(ph_tag)::value;          // == j % 4
    constexpr int SLOT_D = (PH + 3) & 3, SLOT_DN = PH, SLOT_E = (PH + 1) & 3, SLOT_DMA = (PH + 2) & 3;
    f32x16& Scur = (PH & 1) ? SB : SA;
    f32x16& Snxt = (PH & 1) ? SA : SB;
    bf16x8(&cfn)[2] = (PH & 1) ? CB : CA;
    bf16x8(&cfp)[2] = (PH & 1) ? CA : CB;
    const int jt = j + 2 <= last ? j + 2 : last;                     // DMA source (clamped in the tail)
    const float* bj = bias_s + (j & (FF_BIAS_RING_TILES - 1)) * FF_BN;
    char* cst_w = cst + ((PH >> 1) & 1) * 4096;                      // staging buffer filled by tiles 2t, 2t+1
    const char* cst_r = cst + (((PH >> 1) & 1) ^ 1) * 4096;          // staging buffer drained (tiles 2t-2, 2t-1)
    // the first two iterations have no finished pair to drain: their two stores go to a dummy line past the latent
    // (keeps the loop branch-free and the number of memory operations per iteration constant for the counted wait)
    bf16_t* dst_pair = j >= 2 ? cdrain + 64 * ((j >> 1) - 1) : dummy_line;
    const int64_t dst_rstride = j >= 2 ? (int64_t)a.n_p : 0;
    f32x4 bq[4];
    float l1_it = 0.f;
    // a VGPR copy of S(j) taken once at the top: reading the accumulator registers element by element made hipcc
    // cluster all 16 elements' VALU work (~85 instructions) into a single MFMA gap
    const f32x16 S = Scur;

#pragma unroll
    for (int i = 0; i < 48; ++i) {
      if (STAMP && i == 12) t1 = stamp();
      if (STAMP && i == 24) t2 = stamp();
      // ---- fragment prefetch (ring carried across iterations)
      {
        const int g = i + DIST;
        if (g < 24) { if (!(FF_KO & 8)) ring[g % RING] = dec_frag(SLOT_D, g); }
        else if (g < 48) ring[g % RING] = enc_frag(SLOT_E, g - 24);
        else { if (!(FF_KO & 8)) ring[g % RING] = dec_frag(SLOT_DN, g - 48); }
      }
      if (i < 4) bq[i] = *reinterpret_cast<const f32x4*>(bj + 8 * i + 4 * ah);    // bias of S rows 8 i + 4 h + (0..3)
      if (i == 12) {
        // everything older than the previous iteration's 2 latent stores has completed once <= 2 operations are
        // outstanding: in particular the 6 DMA pieces of tile j+1.  Raw barrier (no fence).
        __builtin_amdgcn_sched_barrier(0);
        if (FF_KO & 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!(FF_KO & 4) && (i == FF_DMA_G0 || i == FF_DMA_G1 || i == FF_DMA_G2)) dma_pair(i == FF_DMA_G0 ? 0 : (i == FF_DMA_G1 ? 1 : 2), jt, SLOT_DMA);
      if (!(FF_KO & 4) && (PH & 1) == 0 && i == FF_DMA_G2 + 2 && w == 0) {
        // bias of tiles j+2, j+3 (64 floats) -> ring slots (j+2) % 8, (j+3) % 8; past the end the last pair is re-copied
        const int jb = j + 2 <= a.ntiles - 2 ? j + 2 : a.ntiles - 2;
        glds4(a.bias + (int64_t)jb * FF_BN, (unsigned)(lane * 4),
              (unsigned)__builtin_amdgcn_readfirstlane((int)(smem_base + FF_FIXED_LDS + ((j + 2) & (FF_BIAS_RING_TILES - 1)) * FF_BN * 4)));
      }
      // latent element e at gap 6 + 5 (e >> 1) + 2 (e & 1): gaps 6, 8, 11, 13, ..., 41, 43 (computed from the unrolled
      // loop index itself: a lookup table made hipcc emit all 16 elements' work in ONE gap)
      if (!(FF_KO & 1) && i >= 6 && i < 46 && ((i - 6) % 5 == 0 || (i - 6) % 5 == 2)) {
        const int e = 2 * ((i - 6) / 5) + ((i - 6) % 5 == 2);   // S register e <-> column n = (e&3) + 8 (e>>2) + 4 h
        // pre-activation rounded to bf16 BEFORE the fp32 bias add, as CPU autocast does (l1autoencoder.py:74)
        float cv = (FF_KO & 16) ? S[e] : fmaxf(bf16_round(S[e]) + bq[e >> 2][e & 3], 0.f);
        if (PAD) cv = row_ok ? cv : 0.f;
        if (!(FF_KO & 16)) l1_it += cv;
        cfn[e >> 3][e & 7] = (bf16_t)cv;
        if (!(FF_KO & 2) && (e & 3) == 3) {   // 4 consecutive columns ready: 8 bytes into the staging image
          const int k = e >> 2;
          const bf16x4 o = {cfn[e >> 3][(e & 7) - 3], cfn[e >> 3][(e & 7) - 2], cfn[e >> 3][(e & 7) - 1], cfn[e >> 3][e & 7]};
          const int chunk = 4 * (PH & 1) + k;              // 16-B chunk of the 128-B row; +8 bytes for h = 1
          // (+ row bit 3 on the 8-byte half: rows r and r + 8 of a 16-lane ds_write_b64 group otherwise share a bank pair,
          // the 2-way conflict behind SQ_LDS_BANK_CONFLICT = 7.7 % of this kernel's LDS cycles in round 1)
          *reinterpret_cast<bf16x4*>(cst_w + arow * 128 + ((chunk ^ (arow & 7)) << 4) + 8 * (ah ^ ((arow >> 3) & 1))) = o;
        }
      }
      // two full-line pieces of the finished pair per iteration: LDS read in one gap, global store 8 gaps later
      if (!(FF_KO & 2) && (i == 29 || i == 34)) {
        const int r = 8 * (2 * (PH & 1) + (i == 34)) + drow_l;
        dr[i == 34] = *reinterpret_cast<const u32x4*>(cst_r + r * 128 + ((dch ^ (r & 7)) << 4));
      }
      if (!(FF_KO & 2) && (i == 37 || i == 42)) {
        const int p = 2 * (PH & 1) + (i == 42);
        // written once, read once by the backward after 400 MB more of it: non-temporal, it must not evict W^T / x lines
        // odd pieces hold rows with bit 3 set: their 8-byte halves were stored swapped (see the staging write)
        const u32x4 dv = dr[i == 42];
        const u32x4 ov = (i == 42) ? u32x4{dv[2], dv[3], dv[0], dv[1]} : dv;
        __builtin_nontemporal_store(ov, reinterpret_cast<u32x4*>(dst_pair + (int64_t)(8 * p) * dst_rstride));
      }
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8 fa = ring[i % RING];
      if (i < 24) {
        acc[i >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, cfp[i & 1], acc[i >> 1], 0, 0, 0);
      } else if (i == 24) {      // S(j+1) starts from zero (the MFMA's C operand is the literal 0)
        f32x16 zero;
#pragma unroll
        for (int r = 0; r < 16; ++r) zero[r] = 0.f;
        Snxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, xfrag[0], zero, 0, 0, 0);
      } else {
        Snxt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, xfrag[i - 24], Snxt, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    l1_acc += l1_it;
    if (STAMP) {
      t3 = stamp();
      tsum[0] += t1 - t0; tsum[1] += t2 - t1; tsum[2] += t3 - t2; tsum[3] += 1;
    }
  };
  unsigned long long clk_t0 = 0, clk_r0 = 0;
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
#pragma unroll
      for (int q = 0; q < 4; ++q) a.stamps[((int64_t)wg * 4 + w) * 8 + q] = tsum[q];
      a.stamps[((int64_t)wg * 4 + w) * 8 + 4] = t1 - clk_t0;
      a.stamps[((int64_t)wg * 4 + w) * 8 + 5] = r1 - clk_r0;
      a.stamps[((int64_t)wg * 4 + w) * 8 + 6] = clk_t0 - clk_k0;      // prologue cycles
    }
  }
  // ---- final half iteration: decoder of the last tile (slot 3, c^T in CB); its first DIST fragments are in the ring
#pragma unroll
  for (int i = 0; i < 24; ++i) {
    if (i + DIST < 24) ring[(i + DIST) % RING] = dec_frag(3, i + DIST);
    __builtin_amdgcn_sched_barrier(0);
    acc[i >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[i % RING], CB[i & 1], acc[i >> 1], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  // drain the last pair of latent tiles
  {
    const char* cst_r = cst + (((a.ntiles >> 1) - 1) & 1) * 4096;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = 8 * p + drow_l;
      const u32x4 dv = *reinterpret_cast<const u32x4*>(cst_r + r * 128 + ((dch ^ (r & 7)) << 4));
      const u32x4 v = (p & 1) ? u32x4{dv[2], dv[3], dv[0], dv[1]} : dv;       // rows with bit 3 set: halves stored swapped
      __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(cdrain + (int64_t)(8 * p) * a.n_p + 64 * ((a.ntiles >> 1) - 1)));
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- epilogue: x_hat^T accumulators -> residual / dx_hat / squared-error sums.
  // acc[dt][r] <-> d = 32 dt + (r&3) + 8 (r>>2) + 4 h, row mrow.
  float sq = 0.f, plain = 0.f, nmask = 0.f;
  // every value of this lane belongs to ONE activation row (mrow): rows past M read row M-1 (valid memory) and the
  // row mask is applied once at the end.  With d == 384 every 4-column group is whole -> vector loads, no branches.
  const int64_t lrow = row_ok ? mrow : a.M - 1;
  const T* xrow = reinterpret_cast<const T*>(a.x) + lrow * a.d;
  bf16_t* drow = a.dxh + mrow * FF_D;
  typedef __attribute__((ext_vector_type(4))) T Tx4;
  const bool vec_ok = (a.d == FF_D) && ((reinterpret_cast<uintptr_t>(a.x) & (sizeof(T) * 4 - 1)) == 0);
  const float rmask = row_ok ? 1.f : 0.f;
  constexpr int FF_DXH_PITCH = FF_D * 2 + 16;                    // 784 B: 16-byte aligned rows, at most 2-way bank conflicts
  char* stg = smem + w * 32768;                                  // per-wave [32 rows][784 B] inside rings + latent staging
  static_assert(32 * FF_DXH_PITCH <= 32768 && 4 * 32768 <= FF_FIXED_LDS, "dx_hat staging must fit the idle LDS");
  // 2-byte activations of full workgroups: the wave's 32 x rows are ONE contiguous 24 KiB block, fetched with 24
  // coalesced 16-byte loads per lane into the staging image; each lane then reads its row from LDS and overwrites the
  // 8 bytes it consumed with the 8 bytes of dx_hat it produced (same position, same size).
  constexpr bool X_VIA_LDS = !PAD && sizeof(T) == 2;
  if (X_VIA_LDS && vec_ok) {
    const char* xblk = reinterpret_cast<const char*>(reinterpret_cast<const T*>(a.x) + m0 * FF_D);
#pragma unroll
    for (int p0 = 0; p0 < 24; p0 += 8) {        // 8 loads in flight per lane (register budget)
      u32x4 xr[8];
#pragma unroll
      for (int pc = 0; pc < 8; ++pc) xr[pc] = *reinterpret_cast<const u32x4*>(xblk + (p0 + pc) * 1024 + lane * 16);
#pragma unroll
      for (int pc = 0; pc < 8; ++pc) {
        const int off = (p0 + pc) * 1024 + lane * 16, r = off / (FF_D * 2), cb = off - r * (FF_D * 2);
        *reinterpret_cast<u32x4*>(stg + r * FF_DXH_PITCH + cb) = xr[pc];
      }
    }
  }
  if (vec_ok) {
#pragma unroll
    for (int dt = 0; dt < 12; ++dt) {
      Tx4 xv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        xv[k] = X_VIA_LDS ? *reinterpret_cast<const Tx4*>(stg + arow * FF_DXH_PITCH + (32 * dt + 8 * k + 4 * ah) * 2)
                          : *reinterpret_cast<const Tx4*>(xrow + 32 * dt + 8 * k + 4 * ah);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        bf16x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float xf = (float)xv[k][q];
          const float e = bf16_round(acc[dt][4 * k + q]) - xf;
          const float e2 = e * e;
          plain += e2;
          const float keep = (xf != -1.0f) ? rmask : 0.f;
          nmask += rmask - keep;
          sq += keep * e2;
          o[q] = (bf16_t)(keep * (e * 2.0f));
        }
        // staged in LDS (rings and latent staging are idle: the barrier above retired every DMA): one row per lane
        // straight to HBM would be 48 scattered 8-byte stores per lane, a store-ISSUE-bound tail
        *reinterpret_cast<bf16x4*>(stg + arow * FF_DXH_PITCH + (32 * dt + 8 * k + 4 * ah) * 2) = o;
      }
    }
    // the wave's 32 rows are one contiguous 24 KiB block of dx_hat: 24 fully coalesced 16-byte stores per lane.  Default
    // cache policy on purpose: dx_hat is re-read by the 24 column-tile workgroups of the backward (non-temporal stores
    // here cost +6 % in the forward and +8 % in the backward, same-box A/B)
    char* gblk = reinterpret_cast<char*>(a.dxh + m0 * FF_D);
#pragma unroll
    for (int pc = 0; pc < 24; ++pc) {
      const int off = pc * 1024 + lane * 16, r = off / (FF_D * 2), cb = off - r * (FF_D * 2);
      *reinterpret_cast<u32x4*>(gblk + off) = *reinterpret_cast<const u32x4*>(stg + r * FF_DXH_PITCH + cb);
    }
  } else {
    for (int dt = 0; dt < 12; ++dt)
      for (int k = 0; k < 4; ++k) {
        const int d0 = 32 * dt + 8 * k + 4 * ah;
        bf16x4 o;
        for (int q = 0; q < 4; ++q) {
          const bool valid = d0 + q < a.d;
          const float xf = (float)xrow[valid ? d0 + q : a.d - 1];
          float accv = 0.f;
#pragma unroll
          for (int dd = 0; dd < 12; ++dd)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr)
              if (dd == dt && rr == 4 * k + q) accv = acc[dd][rr];
          const float e = bf16_round(accv) - xf;
          const float e2 = valid ? e * e : 0.f;
          plain += e2;
          const float keep = (valid && xf != -1.0f) ? rmask : 0.f;
          nmask += (valid ? rmask : 0.f) - keep;
          sq += keep * e2;
          o[q] = (bf16_t)(keep * (e * 2.0f));
        }
        *reinterpret_cast<bf16x4*>(drow + d0) = o;
      }
  }
  plain *= rmask;
  float* red = reinterpret_cast<float*>(smem);      // the W rings are idle now
  const float l1s = block_sum_256(l1_acc, red);
  const float sqs = block_sum_256(sq, red + 8);
  const float pls = block_sum_256(plain, red + 16);
  const float nms = block_sum_256(nmask, red + 24);
  if (STAMP && lane == 0) a.stamps[((int64_t)wg * 4 + w) * 8 + 7] = __builtin_amdgcn_s_memtime() - clk_k0;   // whole kernel
  if (t == 0) {
    a.cnt_part[wg] = nms;
    a.l1_part[wg] = l1s;
    a.sq_part[2 * wg] = sqs;
    a.sq_part[2 * wg + 1] = pls;
  }
}
