// Does anything overlap with v_mfma_f64_16x16x4_f64 on gfx950?  (decides how much the scan epilogue costs)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void k(float* out, int iters)
{
    v4f64 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    double d0 = a + 1, d1 = a + 2, d2 = a + 3, d3 = a + 4;
    for (int i = 0; i < iters; ++i) {
        if (MODE & 1) {   // 4 MFMA
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
        if (MODE & 2) {   // 32 f32 FMA
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f0 = fmaf(f0, 1.0001f, 0.5f); f1 = fmaf(f1, 1.0001f, 0.5f); f2 = fmaf(f2, 1.0001f, 0.5f); f3 = fmaf(f3, 1.0001f, 0.5f);
                f4 = fmaf(f4, 1.0001f, 0.5f); f5 = fmaf(f5, 1.0001f, 0.5f); f6 = fmaf(f6, 1.0001f, 0.5f); f7 = fmaf(f7, 1.0001f, 0.5f);
            }
        }
        if (MODE & 4) {   // 8 cvt_f32_f64 + 8 rcp_f32 (the scan epilogue's per-output pair), kept live
            f0 += __builtin_amdgcn_rcpf((float)d0); f1 += __builtin_amdgcn_rcpf((float)d1);
            f2 += __builtin_amdgcn_rcpf((float)d2); f3 += __builtin_amdgcn_rcpf((float)d3);
            d0 += 1.0; d1 += 1.0; d2 += 1.0; d3 += 1.0;
            f4 += __builtin_amdgcn_rcpf((float)d0); f5 += __builtin_amdgcn_rcpf((float)d1);
            f6 += __builtin_amdgcn_rcpf((float)d2); f7 += __builtin_amdgcn_rcpf((float)d3);
        }
        if (MODE & 8) {   // 16 int ops
#pragma unroll
            for (int u = 0; u < 16; ++u) { asm volatile("v_xor_b32 %0, %0, %1" : "+v"(f0) : "v"(f1)); }
        }
    }
    v4f64 c = c0 + c1 + c2 + c3;
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xFFFFF] = (float)(c[0] + c[1] + c[2] + c[3] + d0 + d1 + d2 + d3) + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}

template <int MODE>
void run(const char* name, float* out, int cus, int wpc)
{
    const int iters = 20000;
    int blocks = cus * wpc / 4;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s waves/CU %2d: %.3f ms -> %.1f cycles per loop trip per SIMD (@2.4 GHz)\n", name, wpc, ms,
           ms * 1e-3 * 2.4e9 / (1.0 * iters * wpc / 4));
}

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    float* out; (void)hipMalloc(&out, sizeof(float) * (1 << 20));
    for (int wpc : {4, 8, 16}) {
        run<1>("4 mfma_f64", out, p.multiProcessorCount, wpc);
        run<2>("32 fma_f32", out, p.multiProcessorCount, wpc);
        run<3>("4 mfma_f64 + 32 fma_f32", out, p.multiProcessorCount, wpc);
        run<4>("8 cvt_f32_f64 + 8 rcp_f32 (+adds)", out, p.multiProcessorCount, wpc);
        run<5>("4 mfma_f64 + 8 cvt + 8 rcp", out, p.multiProcessorCount, wpc);
        run<8>("16 v_xor_b32", out, p.multiProcessorCount, wpc);
        run<9>("4 mfma_f64 + 16 v_xor_b32", out, p.multiProcessorCount, wpc);
    }
    return 0;
}
