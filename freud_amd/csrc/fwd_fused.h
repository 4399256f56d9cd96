// Fused forward of the tied-weight L1 SAE for d_model (padded) == 384 on gfx950.
//
//   c = relu(bf16(x W) + b)          (l1autoencoder.py:74)        -> stored once as bf16 for the backward
//   x_hat = bf16(c W^T)              (l1autoencoder.py:84)
//   l1 = sum c ; masked squared error, dx_hat = bf16(2 alpha (x_hat - x)[x != -1] / count)   (:85-86, :29-36)
//
// One workgroup = 4 waves (one per SIMD, whole register file) = 128 activation rows; wave w owns rows
// 32w..32w+31 for the whole dictionary sweep:
//   - its x rows stay in registers as 24 MFMA B-fragments (K = d = 384);
//   - its x_hat^T tile [384 x 32] stays in 192 accumulator registers (12 MFMA tiles);
//   - the dictionary is swept in 32-column tiles.  Per tile, S^T[32 cols x 32 rows] = W^T tile . x^T
//     (24 MFMAs, A = W^T rows from LDS) -> bias + ReLU on the accumulator registers -> those registers ARE the
//     B operand of x_hat^T += W tile . c^T (24 MFMAs, A = W rows from LDS in the accumulator's permuted k
//     order, prepared by normalize_cast_kernel as the `Wp` copy) -- c never round-trips through LDS or HBM
//     between the two GEMMs; it is written to HBM once, straight from registers, for the backward.
//   - software pipeline across tiles: the encoder MFMAs of tile j+1 run first, with the bias/ReLU/store work of
//     tile j in their gaps, then the decoder MFMAs of tile j.  W tiles arrive by LDS-DMA two / one tile ahead.
#pragma once
#include "bwd_fused.h"

constexpr int FF_D = 384;
constexpr int FF_BM = 128;                        // rows per workgroup
constexpr int FF_BN = 32;                         // dictionary columns per tile
constexpr int FF_WT_BYTES = FF_BN * FF_D * 2;     // 24576: W^T tile [32 n][384 d]
constexpr int FF_WP_BYTES = FF_D * FF_BN * 2;     // 24576: W tile   [384 d][32 n (permuted)]
constexpr int FF_DEPTH = 3;                        // LDS ring slots per operand: one being read, two landing
constexpr int FF_RING_BYTES = FF_DEPTH * (FF_WT_BYTES + FF_WP_BYTES);   // 147456

struct FwdFusedArgs {
  const bf16_t* xb;      // [M_p][384]  bf16 GEMM operand
  const void* x;         // original activations [M][d] (dtype T) for the residual
  const bf16_t* Wt;      // [n_p][384]
  const bf16_t* Wp;      // [n_p/32][384][32 permuted]
  const float* bias;     // [n_p]
  const float* scal;     // scal[1] = alpha / count
  bf16_t* c;             // [M_p][n_p]
  bf16_t* dxh;           // [M_p][384]
  float* l1_part;        // [M_p/128]
  float* sq_part;        // [M_p/128][2]
  int64_t M;
  int d, n_p, ntiles;    // ntiles = n_p / 32
  int dbg;               // timing experiments only: 1 = skip latent stores, 2 = skip in-loop DMA
};

template <typename T>
__global__ __launch_bounds__(256, 1) void fwd_fused_d384_kernel(FwdFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int arow = lane & 31, ah = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.x * FF_BM + 32 * w;     // first row of this wave
  const int64_t mrow = m0 + arow;                               // the row this lane's accumulator column is
  const bool row_ok = mrow < a.M;
  float* bias_s = reinterpret_cast<float*>(smem + FF_RING_BYTES);

  // x fragments: B[k = d][col = m] -> lane (m = lane&31, h) holds xb[m0 + m][16 kk + 8 h ..+8]
  bf16x8 xfrag[24];
  {
    const bf16_t* xp = a.xb + mrow * FF_D + 8 * ah;
#pragma unroll
    for (int kk = 0; kk < 24; ++kk) xfrag[kk] = *reinterpret_cast<const bf16x8*>(xp + 16 * kk);
  }
  for (int i = t; i < a.n_p; i += 256) bias_s[i] = a.bias[i];

  f32x16 acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // ---- LDS-DMA plan.  W^T tile: 3 sub-tiles [32][256 B] (dual_off swizzle on the source chunk), 24 pieces of
  // 1 KB (4 rows each).  W tile: 384 rows x 64 B, chunk XOR ((d>>2)&3), 24 pieces (16 rows each).  6 + 6 per wave.
  unsigned voff_t[6], loff_t[6], voff_p[6], loff_p[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int inst = w + 4 * i;
    {
      const int sub = inst >> 3, row = 4 * (inst & 7) + (lane >> 4), pc = lane & 15;
      const int ch = pc ^ (((row & 3) << 2) | ((row >> 2) & 3));
      voff_t[i] = (unsigned)(row * (FF_D * 2) + (sub * 16 + ch) * 16);
      loff_t[i] = (unsigned)__builtin_amdgcn_readfirstlane(sub * 8192 + (inst & 7) * 1024);
    }
    {
      const int row = 16 * inst + (lane >> 2), pc = lane & 3;
      const int ch = pc ^ ((row >> 2) & 3);
      voff_p[i] = (unsigned)(row * 64 + ch * 16);
      loff_p[i] = (unsigned)__builtin_amdgcn_readfirstlane(inst * 1024);
    }
  }
  typedef __attribute__((address_space(3))) char* lptr_t;
  const unsigned smem_base = (unsigned)(uintptr_t)(lptr_t)smem;
  // pair p (0..5): one piece of W^T tile `jt` into Wt slot `st`, one piece of W tile `jp` into Wp slot `sp`
  auto dma_pair = [&](int p, int jt, int st, int jp, int sp) {
    const unsigned dst_t = smem_base + st * FF_WT_BYTES + loff_t[p];
    const unsigned dst_p = smem_base + FF_DEPTH * FF_WT_BYTES + sp * FF_WP_BYTES + loff_p[p];
    glds16_x2(a.Wt + (int64_t)jt * FF_BN * FF_D, a.Wp + (int64_t)jp * FF_BN * FF_D, voff_t[p], voff_p[p], dst_t, dst_p);
  };
  const int last = a.ntiles - 1;
  // prologue: W^T tiles 0, 1, 2 and W tiles 0, 1
#pragma unroll
  for (int p = 0; p < 6; ++p) dma_pair(p, 0, 0, 0, 0);
#pragma unroll
  for (int p = 0; p < 6; ++p) dma_pair(p, last < 1 ? last : 1, 1, last < 1 ? last : 1, 1);
#pragma unroll
  for (int p = 0; p < 6; ++p) dma_pair(p, last < 2 ? last : 2, 2, last < 1 ? last : 1, 1);
#pragma unroll
  for (int kk = 0; kk < 24; ++kk) asm volatile("" : "+v"(xfrag[kk]));   // retire hipcc-tracked loads before the loop
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // loop-invariant per-lane LDS read offsets
  int roff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) roff[i] = dual_off(arow, 2 * i + ah);
  int poff[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) poff[s] = arow * 64 + (((2 * s + ah) ^ ((arow >> 2) & 3)) << 4);

  // per-lane output pointer for c: row mrow, columns 4 h + 8 k (k = 0..3) of the current tile
  bf16_t* cptr = a.c + mrow * a.n_p + 4 * ah;
  float l1_acc = 0.f;

  // ---- S for tile 0 (no overlap partner yet)
  f32x16 S;
#pragma unroll
  for (int r = 0; r < 16; ++r) S[r] = 0.f;
  {
    const char* img_t = smem;   // slot 0
#pragma unroll
    for (int kk = 0; kk < 24; ++kk) {
      const bf16x8 fa = *reinterpret_cast<const bf16x8*>(img_t + (kk >> 3) * 8192 + roff[kk & 7]);
      S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, xfrag[kk], S, 0, 0, 0);
    }
  }

  constexpr int DIST = 8;
  int s0 = 0, s1 = 1, s2 = 2;                               // j % 3, (j+1) % 3, (j+2) % 3
  for (int j = 0; j < a.ntiles; ++j) {
    // iteration j: encoder MFMAs of tile j+1 (S_next) with the epilogue of tile j in their gaps, then decoder
    // MFMAs of tile j.  Reads W^T slot (j+1)%3 and W slot j%3.  DMA issued now: W^T tile j+3 -> slot j%3 (its
    // tile j was consumed in iteration j-1) and W tile j+2 -> slot (j+2)%3; they are only required to have
    // landed by the END of iteration j+1 (counted vmcnt), so a copy has more than one iteration to arrive.
    const int jn = j + 2 <= last ? j + 2 : last;           // clamped sources: the tail re-copies valid tiles
    const int jt = j + 3 <= last ? j + 3 : last;
    const char* img_t = smem + s1 * FF_WT_BYTES;
    const char* img_p = smem + FF_DEPTH * FF_WT_BYTES + s0 * FF_WP_BYTES;
    const float* bj = bias_s + j * FF_BN;

    auto load_frag = [&](int i) -> bf16x8 {
      if (i < 24) return *reinterpret_cast<const bf16x8*>(img_t + (i >> 3) * 8192 + roff[i & 7]);
      const int tt = i - 24, dt = tt >> 1, s = tt & 1;
      return *reinterpret_cast<const bf16x8*>(img_p + dt * 2048 + poff[s]);
    };
    bf16x8 ring[DIST + 1];
#pragma unroll
    for (int i = 0; i < DIST; ++i) ring[i] = load_frag(i);

    f32x16 Sn;
#pragma unroll
    for (int r = 0; r < 16; ++r) Sn[r] = 0.f;
    bf16x8 cf[2];
    f32x4 bq[4];

#pragma unroll
    for (int i = 0; i < 48; ++i) {
      if (i + DIST < 48) ring[(i + DIST) % (DIST + 1)] = load_frag(i + DIST);
      if (i % 4 == 1 && i / 4 < 6 && !(a.dbg & 2)) dma_pair(i / 4, jt, s0, jn, s2);   // slots idle in this iteration
      if (i < 4) bq[i] = *reinterpret_cast<const f32x4*>(bj + 8 * i + 4 * ah);    // bias of S rows 8 i + 4 h + (0..3)
      if (i >= 4 && i < 20) {     // one latent element per gap: S register e <-> column n = (e&3) + 8 (e>>2) + 4 h
        const int e = i - 4;
        float cv = fmaxf(bf16_round(S[e]) + bq[e >> 2][e & 3], 0.f);
        cv = row_ok ? cv : 0.f;
        l1_acc += cv;
        cf[e >> 3][e & 7] = (bf16_t)cv;
        if ((e & 3) == 3) {       // 4 consecutive columns ready: 8-B store straight from registers
          const int k = e >> 2;
          bf16x4 o = {cf[e >> 3][(e & 7) - 3], cf[e >> 3][(e & 7) - 2], cf[e >> 3][(e & 7) - 1], cf[e >> 3][e & 7]};
          if (!(a.dbg & 1)) *reinterpret_cast<bf16x4*>(cptr + 8 * k) = o;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8 fa = ring[i % (DIST + 1)];
      if (i < 24) {
        Sn = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, xfrag[i], Sn, 0, 0, 0);
      } else {
        acc[(i - 24) >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, cf[i & 1], acc[(i - 24) >> 1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    S = Sn;
    cptr += FF_BN;
    { const int tmp = s0; s0 = s1; s1 = s2; s2 = tmp; }
    // this iteration issued 12 LDS-DMA pieces + 4 latent stores per wave: everything older has completed once at
    // most 16 operations are outstanding.  Raw barrier (no fence): __syncthreads() would drain vmcnt to 0.
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- epilogue: x_hat^T accumulators -> residual / dx_hat / squared-error sums.
  // acc[dt][r] <-> d = 32 dt + (r&3) + 8 (r>>2) + 4 h, row mrow.
  const float scale = a.scal[1];
  float sq = 0.f, plain = 0.f;
  // rows past M read row M-1 (valid memory) and are masked; with d == 384 every 4-column group is whole, so the
  // original activations come in as vector loads issued back to back (no per-element branches / waits)
  const int64_t lrow = row_ok ? mrow : a.M - 1;
  const T* xrow = reinterpret_cast<const T*>(a.x) + lrow * a.d;
  bf16_t* drow = a.dxh + mrow * FF_D;
  typedef __attribute__((ext_vector_type(4))) T Tx4;
  const bool vec_ok = (a.d == FF_D) && ((reinterpret_cast<uintptr_t>(a.x) & (sizeof(T) * 4 - 1)) == 0);
#pragma unroll
  for (int dt = 0; dt < 12; ++dt) {
    float xv[4][4];
    if (vec_ok) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const Tx4 v = *reinterpret_cast<const Tx4*>(xrow + 32 * dt + 8 * k + 4 * ah);
#pragma unroll
        for (int q = 0; q < 4; ++q) xv[k][q] = (float)v[q];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int dd = 32 * dt + 8 * k + 4 * ah + q;
          xv[k][q] = (float)xrow[dd < a.d ? dd : a.d - 1];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int d0 = 32 * dt + 8 * k + 4 * ah;
      bf16x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool valid = row_ok && (d0 + q < a.d);
        const float e = bf16_round(acc[dt][4 * k + q]) - xv[k][q];
        const float e2 = valid ? e * e : 0.f;
        plain += e2;
        const bool keep = valid && (xv[k][q] != -1.0f);
        sq += keep ? e2 : 0.f;
        o[q] = (bf16_t)(keep ? (e * 2.0f) * scale : 0.f);
      }
      *reinterpret_cast<bf16x4*>(drow + d0) = o;
    }
  }
  float* red = reinterpret_cast<float*>(smem);      // the W rings are idle now
  const float l1s = block_sum_256(l1_acc, red);
  const float sqs = block_sum_256(sq, red + 8);
  const float pls = block_sum_256(plain, red + 16);
  if (t == 0) {
    a.l1_part[blockIdx.x] = l1s;
    a.sq_part[2 * blockIdx.x] = sqs;
    a.sq_part[2 * blockIdx.x + 1] = pls;
  }
}
