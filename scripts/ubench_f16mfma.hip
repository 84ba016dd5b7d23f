// Lab: issue rates of the f16 matrix-core instructions the coarse-gated scan uses (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_f16mfma scripts/ubench_f16mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 v4f16 __attribute__((ext_vector_type(4)));
typedef _Float16 v8f16 __attribute__((ext_vector_type(8)));
typedef float v4f32 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e__), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc)
{
    v8f16 a8, b8;
    v4f16 a4, b4;
    for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(threadIdx.x * 0.001f + i); b8[i] = (_Float16)(i * 0.5f); }
    for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
    v4f32 acc[4] = {{0, 0, 0, 0}, {1, 1, 1, 1}, {2, 2, 2, 2}, {3, 3, 3, 3}};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {          // 4 independent K = 32
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[q], 0, 0, 0);
        } else if (MODE == 1) {   // 4 independent legacy K = 16
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[q], 0, 0, 0);
        } else if (MODE == 2) {   // the scan's tile: 4 x K32 then 4 x K16 on the same accumulators
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else {                  // 8 x K32 (the third product as a zero-padded K = 32)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b8, a8, acc[q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
int run(const char* what, int waves_per_simd, int per_iter)
{
    float* out; long long* cyc;
    CK(hipMalloc((void**)&out, 256 * 8 * 256 * 4)); CK(hipMalloc((void**)&cyc, 8));
    const int iters = 20000, blocks = 256 * waves_per_simd;     // 256-thread blocks: one wave per SIMD each
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, 100, cyc);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, cyc); CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    // per SIMD: waves_per_simd waves x iters x per_iter instructions
    printf("%-62s %d wave(s)/SIMD: %.1f shader clocks per instruction (one wave's view: %.1f), kernel %.3f ms\n", what, waves_per_simd,
           (double)c / ((double)iters * per_iter * waves_per_simd), (double)c / ((double)iters * per_iter), ms);
    (void)hipFree(out); (void)hipFree(cyc);
    return 0;
}

int main()
{
    for (int w = 1; w <= 2; ++w) {
        if (run<0>("v_mfma_f32_16x16x32_f16, 4 independent accumulators", w, 4)) return 1;
        if (run<1>("v_mfma_f32_16x16x16_f16 (legacy), 4 independent accumulators", w, 4)) return 1;
        if (run<2>("4 x K32 then 4 x K16 on the same accumulators (the scan's tile)", w, 8)) return 1;
        if (run<3>("4 x K32 then 4 x K32 on the same accumulators", w, 8)) return 1;
    }
    return 0;
}
