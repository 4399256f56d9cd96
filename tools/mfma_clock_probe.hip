// Diagnostic only (not part of the product): what FLOP/s and in-kernel clock does MI355X hold on bare bf16 MFMA loops of
// the two shapes, one wave per SIMD, random operands, with and without an LDS operand stream and with VALU filler in the
// MFMA gaps?  Used to put a measured ceiling next to the fused SAE kernels.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_clock_probe.hip -o build/mfma_clock_probe && build/mfma_clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cstdint>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Stamp { unsigned long long t0, t1, r0, r1; };

// SHAPE 0: 32x32x16, 12 accumulators (192 regs) ; SHAPE 1: 16x16x32, 48 accumulators (192 regs)
// LDS: 0 = both operands in registers; 1 = A operand re-read from LDS per MFMA (32x32) / per 2 MFMAs (16x16) so LDS bytes match
// VALU: number of v_fma filler instructions per 32 MFMA-cycles
template <int SHAPE, int LDS, int VALU>
__global__ __launch_bounds__(256, 1) void probe(const u32x4* __restrict__ src, float* __restrict__ out, Stamp* st, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4* l4 = (u32x4*)smem;
    for (int i = threadIdx.x; i < 24 * 64 * 4; i += 256) l4[i] = src[(i * 7 + blockIdx.x) & 16383];
    __syncthreads();
    u32x4 bfrag[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) bfrag[i] = src[(i * 64 + lane + wave * 1536 + blockIdx.x * 31) & 16383];
    u32x4 areg[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) areg[i] = src[(i * 64 + lane + 5000) & 16383];
    float fill[4] = {1.f, 2.f, 3.f, 4.f};
    const u32x4* lw = l4 + wave * 24 * 64 + lane;
    unsigned long long t0 = 0, r0 = 0;
    if (SHAPE == 0) {
        f32x16 acc[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] = (f32x16)(0.f);
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
            u32x4 a[24];
            if (LDS) {
#pragma unroll
                for (int i = 0; i < 24; ++i) a[i] = lw[i * 64];
            }
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                u32x4 av = LDS ? a[i] : areg[i & 3];
                acc[i % 12] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bfrag[i]), acc[i % 12], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < VALU; ++v) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(fill[v & 3]) : "v"(1.0001f));
                if (VALU) __builtin_amdgcn_sched_barrier(0);
            }
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        float s = fill[0] + fill[1] + fill[2] + fill[3];
#pragma unroll
        for (int i = 0; i < 12; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) s += acc[i][j];
        out[blockIdx.x * 256 + threadIdx.x] = s;
        if (threadIdx.x == 0) st[blockIdx.x] = Stamp{t0, t1, r0, r1};
    } else {
        f32x4 acc[48];
#pragma unroll
        for (int i = 0; i < 48; ++i) acc[i] = (f32x4)(0.f);
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int it = 0; it < iters; ++it) {
            u32x4 a[24];
            if (LDS) {
#pragma unroll
                for (int i = 0; i < 24; ++i) a[i] = lw[i * 64];
            }
#pragma unroll
            for (int i = 0; i < 48; ++i) {
                u32x4 av = LDS ? a[i >> 1] : areg[i & 3];
                u32x4 bv = bfrag[i >> 1];
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(av), "v"(bv));
                if (VALU) {
#pragma unroll
                    for (int v = 0; v < VALU / 2; ++v) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(fill[(v + 2 * (i & 1)) & 3]) : "v"(1.0001f));
                }
            }
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        float s = fill[0] + fill[1] + fill[2] + fill[3];
#pragma unroll
        for (int i = 0; i < 48; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        out[blockIdx.x * 256 + threadIdx.x] = s;
        if (threadIdx.x == 0) st[blockIdx.x] = Stamp{t0, t1, r0, r1};
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int SHAPE, int LDS, int VALU>
void run(const char* name, const u32x4* src, float* out, Stamp* st, int zero) {
    const int grid = 256, iters = 3000;
    const size_t smem = 24 * 64 * 4 * 16;   // 96 KiB -> one workgroup per CU
    CK(hipFuncSetAttribute((const void*)probe<SHAPE, LDS, VALU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // >= 2 s of back-to-back launches so that the clock settles, then time the last 20
    float ms = 0; int warm = 0;
    CK(hipEventRecord(e0));
    probe<SHAPE, LDS, VALU><<<grid, 256, smem>>>(src, out, st, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    warm = std::max(20, (int)(2000.f / ms));
    for (int i = 0; i < warm; ++i) probe<SHAPE, LDS, VALU><<<grid, 256, smem>>>(src, out, st, iters);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) probe<SHAPE, LDS, VALU><<<grid, 256, smem>>>(src, out, st, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 20;
    std::vector<Stamp> h(grid); CK(hipMemcpy(h.data(), st, grid * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> clk, cyc;
    for (auto& s : h) { clk.push_back(double(s.t1 - s.t0) / double(s.r1 - s.r0) * 100.0); cyc.push_back(double(s.t1 - s.t0)); }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    double flops = 2.0 * 32 * 32 * 16 * 24.0 * iters * 4 * grid;     // both shapes: same flops per iteration per wave
    double ideal = 24.0 * 32 * iters;
    printf("{\"probe\": \"%s\", \"data\": \"%s\", \"ms\": %.4f, \"tflops\": %.1f, \"clock_mhz_median\": %.0f, \"loop_cycles_median\": %.0f, \"mfma_issue_frac\": %.3f}\n",
           name, zero ? "zeros" : "random", ms, flops / (ms * 1e-3) / 1e12, clk[grid / 2], cyc[grid / 2], ideal / cyc[grid / 2]);
    fflush(stdout);
}

int main() {
    u32x4* src; float* out; Stamp* st;
    CK(hipMalloc(&src, 16384 * 16)); CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&st, 256 * sizeof(Stamp)));
    std::vector<uint16_t> h(16384 * 8);
    for (int zero = 0; zero < 2; ++zero) {
        uint32_t s = 12345;
        for (auto& v : h) {
            s = s * 1664525u + 1013904223u;
            // random bf16 in [-2, 2): random sign/mantissa, exponent 125..128
            v = zero ? 0 : (uint16_t)(((s >> 16) & 0x807F) | ((125 + ((s >> 8) & 3)) << 7));
        }
        CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        run<0, 0, 0>("32x32x16 regs", src, out, st, zero);
        run<1, 0, 0>("16x16x32 regs", src, out, st, zero);
        run<0, 1, 0>("32x32x16 lds", src, out, st, zero);
        run<1, 1, 0>("16x16x32 lds", src, out, st, zero);
        run<0, 1, 4>("32x32x16 lds valu4", src, out, st, zero);
        run<1, 1, 4>("16x16x32 lds valu4", src, out, st, zero);
        run<0, 1, 8>("32x32x16 lds valu8", src, out, st, zero);
        run<1, 1, 8>("16x16x32 lds valu8", src, out, st, zero);
    }
    return 0;
}
