// Stand-alone timing of the 256x256 GEMM skeleton (freud_amd/csrc/gemm256.h) at the shapes of the d = 1280 / d = 768 paths,
// for A/B experiments on the K loop (compile-time switches G2X_*).  Build here (hipcc cross-compiles), run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DG2X_...] -o build/kbench/gemm_bench tools/kbench/gemm_bench.hip
//   build/kbench/gemm_bench M N K [mode: 0 row (8-wave kernel), 1 kmajor, 2 row (4-wave gemm256w4.h), 3 compare 0 against 2] [splits]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>

#include "gemm256w4.h"
#include "../../freud_amd/csrc/l1_kernels.h"
#include "../../freud_amd/csrc/gemm256s.h"

struct EpiBf16 {          // plain bf16 store (the encoder epilogue without bias / ReLU)
#ifdef KB_BF16EPI
  static constexpr bool ROUNDS_BF16_FIRST = true;      // the product's one-step bf16 epilogue (the default here is the fp32 two-pass one)
#endif
#ifdef KB_A3
  static constexpr bool DEEP_A_RING = true;      // three-deep A ring (the decoder's form)
#endif
  bf16_t* out;
  int64_t ld;
  __device__ void tile_begin(int, int, int) {}
  struct Pre {};
  __device__ Pre prefetch(int, int) const { return Pre{}; }
  __device__ void apply(int row, int col, f32x4 v, const Pre&) {
    bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    __builtin_nontemporal_store(o, reinterpret_cast<bf16x4*>(out + (int64_t)row * ld + col));
  }
  __device__ void tile_end(float*) {}
  // streaming form (gemm256s.h; mode 4)
  static constexpr bool STREAM = true;
  struct SPre {};
  __device__ void s_begin() {}
  __device__ void s_tile(int, int) {}
  __device__ SPre s_prefetch(int, int) const { return SPre{}; }
  __device__ int64_t s_rows() const { return (int64_t)1 << 62; }
  template <bool PARTIAL>
  __device__ void s_apply(int row, int col, f32x4 v0, f32x4 v1, const SPre&) {
    bf16x8 o = {(bf16_t)v0[0], (bf16_t)v0[1], (bf16_t)v0[2], (bf16_t)v0[3], (bf16_t)v1[0], (bf16_t)v1[1], (bf16_t)v1[2], (bf16_t)v1[3]};
    __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(out + (int64_t)row * ld + col));
  }
  __device__ void s_tile_end(int, int) {}
  __device__ void s_end(float*) {}
};

#ifdef KB_A3
struct EpiSlabK : EpiSlab { static constexpr bool DEEP_A_RING = true; };
#else
typedef EpiSlab EpiSlabK;
#endif

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
  const int64_t M = argc > 1 ? atoll(argv[1]) : 65536, N = argc > 2 ? atoll(argv[2]) : 40960, K = argc > 3 ? atoll(argv[3]) : 1280;
  int kmajor = argc > 4 ? atoi(argv[4]) : 0;
  const int splits = argc > 5 ? atoi(argv[5]) : 1;
  const bool compare = kmajor == 3 || kmajor == 5;      // 5: the 8-wave tile form against the streaming form (mode 4)
  const bool compare_stream = kmajor == 5;
  if (compare) kmajor = 0;
  bf16_t *A, *B, *C;
  float* slab = nullptr;
  CK(hipMalloc(&A, M * K * 2));
  CK(hipMalloc(&B, N * K * 2));
  CK(hipMalloc(&C, M * N * 2));
  std::vector<unsigned short> h(1 << 20);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3C00 + (rand() & 0x3FF) + ((rand() & 1) << 15));   // ~ +-[0.0078, 0.03]
  for (int64_t off = 0; off < M * K; off += (int64_t)h.size()) CK(hipMemcpy(A + off, h.data(), (size_t)std::min<int64_t>(h.size(), M * K - off) * 2, hipMemcpyHostToDevice));
  for (int64_t off = 0; off < N * K; off += (int64_t)h.size()) CK(hipMemcpy(B + off, h.data(), (size_t)std::min<int64_t>(h.size(), N * K - off) * 2, hipMemcpyHostToDevice));
  GemmArgs g{};
  g.A0 = A; g.B0 = B; g.splits = splits;
  if (kmajor == 6) {      // mixed (experiment): A row-major [M][K] (K contiguous), B k-major [K][N]; weight-gradient shape with the d-side operand transposed
    g.lda = K; g.ldb = N; g.nbm = (int)(M / 256); g.nbn = (int)(N / 256); g.ktiles0 = g.ktiles = (int)(K / 64);
    CK(hipMalloc(&slab, (size_t)splits * M * N * 4));
  } else if (kmajor != 1) {      // C[M][N] = A[M][K] B[N][K]^T
    g.lda = K; g.ldb = K; g.nbm = (int)(M / 256); g.nbn = (int)(N / 256); g.ktiles0 = g.ktiles = (int)(K / 64);
  } else {                // weight-gradient shape: C[M][N] = sum_k A[k][M] B[k][N], K rows
    g.lda = M; g.ldb = N; g.nbm = (int)(M / 256); g.nbn = (int)(N / 256); g.ktiles0 = g.ktiles = (int)(K / 64);
    CK(hipMalloc(&slab, (size_t)splits * M * N * 4));
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int mode = kmajor;
  auto run_w4 = [&](bf16_t* out) {
    EpiBf16 e{out, N};
#ifdef KB_M32
    auto kern = gemm256w4m_bf16_kernel<EpiBf16>;
#else
    auto kern = gemm256w4_bf16_kernel<EpiBf16>;
#endif
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(g.nbm * g.nbn * g.splits), dim3(256), G2_LDS_BYTES, 0, g, e);
  };
  auto run_stream = [&](bf16_t* out) {
    EpiBf16 e{out, N};
    auto kern = gemm256s_bf16_kernel<EpiBf16>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G2S_LDS_BYTES));
#ifndef KB_STREAM_GRID
#define KB_STREAM_GRID 512
#endif
    hipLaunchKernelGGL(kern, dim3(KB_STREAM_GRID), dim3(512), G2S_LDS_BYTES, 0, g, e);
  };
  auto run = [&]() {
    if (mode == 4) {
      run_stream(C);
    } else if (mode == 2) {
      run_w4(C);
    } else if (mode == 6) {
      EpiSlabK e{};
      e.slab = slab; e.slab_stride = M * N; e.ld = (int)N;
      auto kern = gemm256_bf16_kernel<OP_ROW, OP_KMAJOR, EpiSlabK>;
      constexpr int lds = epi_deep_a_ring<EpiSlabK>::value ? G2_A3_LDS_BYTES : G2_LDS_BYTES;
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      hipLaunchKernelGGL(kern, dim3(g.nbm * g.nbn * g.splits), dim3(512), lds, 0, g, e);
    } else if (!kmajor) {
      EpiBf16 e{C, N};
#ifdef KB_PERSIST
      auto kern = gemm256_bf16_kernel<OP_ROW, OP_ROW, EpiBf16, true>;       // the tile-loop instantiation on KB_PERSIST workgroups
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS_BYTES));
      hipLaunchKernelGGL(kern, dim3(KB_PERSIST), dim3(512), G2_LDS_BYTES, 0, g, e);
#else
      auto kern = gemm256_bf16_kernel<OP_ROW, OP_ROW, EpiBf16>;
      constexpr int lds = epi_deep_a_ring<EpiBf16>::value ? G2_A3_LDS_BYTES : (epi_rounds_first<EpiBf16>::value && G2_BF16_LDS_BYTES > G2_LDS_BYTES) ? G2_BF16_LDS_BYTES : G2_LDS_BYTES;
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      hipLaunchKernelGGL(kern, dim3(g.nbm * g.nbn * g.splits), dim3(512), lds, 0, g, e);
#endif
    } else {
      EpiSlabK e{};
      e.slab = slab; e.slab_stride = M * N; e.ld = (int)N;
      auto kern = gemm256_bf16_kernel<OP_KMAJOR, OP_KMAJOR, EpiSlabK>;
      constexpr int lds = epi_deep_a_ring<EpiSlabK>::value ? G2_A3_LDS_BYTES : G2_LDS_BYTES;
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      hipLaunchKernelGGL(kern, dim3(g.nbm * g.nbn * g.splits), dim3(512), lds, 0, g, e);
    }
  };
  if (compare) {
    bf16_t* C2;
    CK(hipMalloc(&C2, M * N * 2));
    CK(hipMemset(C, 0, M * N * 2));
    CK(hipMemset(C2, 0xFF, M * N * 2));
    run();
    if (compare_stream) run_stream(C2); else run_w4(C2);
    CK(hipDeviceSynchronize());
    std::vector<unsigned short> c1((size_t)M * N), c2((size_t)M * N);
    CK(hipMemcpy(c1.data(), C, (size_t)M * N * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(c2.data(), C2, (size_t)M * N * 2, hipMemcpyDeviceToHost));
    auto f = [](unsigned short b) { unsigned u = (unsigned)b << 16; float x; memcpy(&x, &u, 4); return x; };
    double maxd = 0, maxv = 0;
    size_t bad = 0, diff = 0;
    for (size_t i = 0; i < c1.size(); ++i) {
      const double a = f(c1[i]), b = f(c2[i]);
      if (!(fabs(a - b) <= 0.01 * fabs(a) + 1e-6)) ++bad;     // one bf16 ulp is 0.4-0.8 %
      if (c1[i] != c2[i]) ++diff;
      if (fabs(a - b) > maxd) maxd = fabs(a - b);
      if (fabs(a) > maxv) maxv = fabs(a);
    }
    printf("compare 8-wave vs 4-wave: %zu of %zu elements differ (bf16 bits), %zu beyond 1 %%, max |diff| %.3g, max |value| %.3g\n", diff,
           c1.size(), bad, maxd, maxv);
    return bad ? 1 : 0;
  }
  for (int i = 0; i < 3; ++i) run();
  CK(hipDeviceSynchronize());
  // hold the clocks with the workload itself for ~0.3 s, then time
  float ms = 0;
  int reps = 0;
  do {
    CK(hipEventRecord(e0));
    for (int i = 0; i < 5; ++i) run();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    reps += 5;
  } while (reps < 40);
#ifdef G2X_STAMP
  {
    const size_t ntile = (size_t)g.nbm * g.nbn * g.splits;
    unsigned long long* dbuf;
    CK(hipMalloc(&dbuf, ntile * 32));
    CK(hipMemset(dbuf, 0, ntile * 32));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g2x_stamps), &dbuf, sizeof(dbuf)));
    run();
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> hs(ntile * 4);
    CK(hipMemcpy(hs.data(), dbuf, ntile * 32, hipMemcpyDeviceToHost));
    double a = 0, b = 0, c = 0;
    if (mode == 4) {       // streaming form: per WORKGROUP {tiles, K-loop cycles, epilogue cycles, whole-kernel cycles} (wave 0)
      double tl = 0, lp = 0, ep = 0, tot = 0;
      for (size_t i = 0; i < (size_t)KB_STREAM_GRID && i < ntile; ++i) { tl += hs[4 * i]; lp += hs[4 * i + 1]; ep += hs[4 * i + 2]; tot += hs[4 * i + 3]; }
      printf("streaming form, per tile (shader cycles, wave 0): K loop %.0f (%.0f per K tile)  epilogue %.0f  other %.0f  -> shares %.1f %% / %.1f %% / %.1f %%\n",
             lp / tl, lp / tl / g.ktiles, ep / tl, (tot - lp - ep) / tl, 100 * lp / tot, 100 * ep / tot, 100 * (tot - lp - ep) / tot);
    }
    for (size_t i = 0; i < ntile; ++i) { a += hs[4 * i]; b += hs[4 * i + 1]; c += hs[4 * i + 2]; }
    if (mode != 4)
    printf("per tile (s_memtime cycles, 100 MHz-independent counter units): prologue %.0f  K loop %.0f  epilogue %.0f  -> shares %.1f %% / %.1f %% / %.1f %%\n",
           a / ntile, b / ntile, c / ntile, 100 * a / (a + b + c), 100 * b / (a + b + c), 100 * c / (a + b + c));
  }
#endif
  if (kmajor == 1 && getenv("KB_HASH")) {      // FNV-1a over the fp32 slab bits: two builds of the k-major kernel must print the same value
    std::vector<unsigned> hs((size_t)splits * M * N);
    CK(hipMemcpy(hs.data(), slab, hs.size() * 4, hipMemcpyDeviceToHost));
    unsigned long long hsh = 1469598103934665603ull;
    for (unsigned v : hs) { hsh ^= v; hsh *= 1099511628211ull; }
    printf("slab hash %016llx\n", hsh);
  }
  const double t = ms / 5 * 1e-3, flops = 2.0 * M * N * K;
  printf("%s M=%lld N=%lld K=%lld splits=%d: %.3f ms  %.1f TFLOP/s (%.1f %% of 2.5 PF)\n", mode == 6 ? "row x kmajor" : mode == 4 ? "row-stream" : mode == 2 ? "row-w4" : (kmajor ? "kmajor" : "row"), (long long)M, (long long)N,
         (long long)K, splits, t * 1e3, flops / t / 1e12, flops / t / 2.5e15 * 100);
  return 0;
}
