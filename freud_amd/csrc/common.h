// Shared device/host helpers for the gfx950 SAE engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

__device__ __forceinline__ float bf16_round(float v) { return (float)(bf16_t)v; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// max over the 32 lanes of a wave half (lanes 0-31 / 32-63), valid in the lanes 16-31 / 48-63 of the half: DPP only (quad
// swaps, row_half_mirror / row_mirror bring in the other quad pair / the other half row, row_bcast15 hands the first row's
// result to the second) -- five ds_bpermute shuffles cost the TopK encoder epilogue +28 % of the whole GEMM, this +5 %
__device__ __forceinline__ float half_wave_max_hi(float v) {
  int x = __float_as_int(v);
  x = __float_as_int(fmaxf(__int_as_float(x), __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false))));    // quad_perm [1,0,3,2]
  x = __float_as_int(fmaxf(__int_as_float(x), __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false))));    // quad_perm [2,3,0,1]
  x = __float_as_int(fmaxf(__int_as_float(x), __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false))));   // row_half_mirror
  x = __float_as_int(fmaxf(__int_as_float(x), __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false))));   // row_mirror
  // rows 1 and 3 (lanes 16-31, 48-63) take lane 15 of the row before them; the other rows keep their own value
  x = __float_as_int(fmaxf(__int_as_float(x), __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x142, 0xA, 0xF, false))));   // row_bcast15
  return __int_as_float(x);
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding GLOBAL access of the wave
// (s_waitcnt vmcnt(0)), and in a GEMM epilogue that is a drain of the tile's non-temporal stores (~1-2 us) at each of its
// barriers.  Use where the threads exchange data through LDS and nothing through global memory.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
}

// block_sum_256 with LDS-only barriers (GEMM epilogue functors)
__device__ __forceinline__ float block_sum_256_lds(float v, float* red /* >= 4 floats of LDS */);

// Sum over a group of 256 consecutive threads (4 waves); result valid in the group's first thread.  blockDim.x is 256
// (one group) or 512 (two groups, each with its own `red`); every thread of the block must call it.
__device__ __forceinline__ float block_sum_256(float v, float* red /* >= 4 floats of LDS */) {
  v = wave_sum(v);
  const int w = (threadIdx.x >> 6) & 3;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_sum_256_lds(float v, float* red) {
  v = wave_sum(v);
  const int w = (threadIdx.x >> 6) & 3;
  lds_barrier();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  lds_barrier();
  return red[0] + red[1] + red[2] + red[3];
}

// LDS-DMA issued from inline asm so that hipcc does not track it (with the builtin it waits vmcnt(0) before the next LDS
// read, and a copy can never overlap the MFMAs that follow it): the caller waits with its own s_waitcnt vmcnt + barrier.
// M0 carries the wave-uniform LDS byte address.  Two pieces per statement (one M0 save/restore): piece k copies
// 64 x 16 B from sbase_k + voff_k (per lane) to LDS [dst_k, dst_k + 1024).
// (Round 6: M0 is declared CLOBBERED instead of being saved and restored around every statement -- nothing else in these kernels lives in
// M0 (gfx9 LDS instructions do not read it), so the save / restore pair was two scalar instructions per statement on the one issue
// port a single-wave-per-SIMD kernel has: 6 of the fused forward's ~28 scalar instructions per iteration.  -DGLDS_KEEP_M0 = the old form.)
#ifdef GLDS_KEEP_M0
__device__ __forceinline__ void glds16_x2(const void* sbase0, const void* sbase1, unsigned voff0, unsigned voff1,
                                          unsigned dst0, unsigned dst1) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
      "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(sbase0), "s"(sbase1), "v"(voff0), "v"(voff1), "s"(dst0), "s"(dst1)
      : "memory");
}
__device__ __forceinline__ void glds16_x2_nt(const void* sbase0, const void* sbase1, unsigned voff0, unsigned voff1,
                                             unsigned dst0, unsigned dst1) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1 nt\n\t"
      "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %2 nt\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(sbase0), "s"(sbase1), "v"(voff0), "v"(voff1), "s"(dst0), "s"(dst1)
      : "memory");
}
__device__ __forceinline__ void glds4(const void* sbase, unsigned voff, unsigned dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %2, %1\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(sbase), "v"(voff), "s"(dst)
      : "memory");
}
#else
__device__ __forceinline__ void glds16_x2(const void* sbase0, const void* sbase1, unsigned voff0, unsigned voff1,
                                          unsigned dst0, unsigned dst1) {
  asm volatile(
      "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0\n\t"
      "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1"
      :
      : "s"(sbase0), "s"(sbase1), "v"(voff0), "v"(voff1), "s"(dst0), "s"(dst1)
      : "memory", "m0");
}

// Same, with the non-temporal cache policy: for streams that are read exactly once (the latent in the backward)
__device__ __forceinline__ void glds16_x2_nt(const void* sbase0, const void* sbase1, unsigned voff0, unsigned voff1,
                                             unsigned dst0, unsigned dst1) {
  asm volatile(
      "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %0 nt\n\t"
      "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1 nt"
      :
      : "s"(sbase0), "s"(sbase1), "v"(voff0), "v"(voff1), "s"(dst0), "s"(dst1)
      : "memory", "m0");
}

// One 256-byte piece: lane l copies 4 B from sbase + voff (per lane) to LDS [dst + 4 l, dst + 4 l + 4).
__device__ __forceinline__ void glds4(const void* sbase, unsigned voff, unsigned dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %0" : : "s"(sbase), "v"(voff), "s"(dst) : "memory", "m0");
}
#endif

// One 1 KiB piece (the single form of glds16_x2)
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" : : "s"(sbase), "v"(voff), "s"(dst) : "memory", "m0");
}

// A loop whose index is a TEMPLATE constant.  `#pragma unroll` gives up silently above LLVM's size threshold; the loop then stays
// rolled and every register array it indexes goes to the stack (fwd_fused2.h's slot loop in round 5: 1.8 KB of scratch per lane after
// a few added lines; topk_select_reg_kernel<44>: its key arrays, 1 KB per lane).
#include <type_traits>
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

static inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// Balanced form of the fused d = 384 backward (bwd_fused.h): workgroup k takes quanta [k m, (k + 1) m) of the tile-major list of
// (column tile, 1/32 of the rows) quanta.  The pieces of column tile j = the workgroups whose quanta meet [32 j, 32 j + 32).
__host__ __device__ __forceinline__ int bal_first_wg(int j, int m) { return (32 * j) / m; }
__host__ __device__ __forceinline__ int bal_pieces(int j, int m) { return (32 * j + 31) / m - (32 * j) / m + 1; }

