// Per-instruction issue cost of the scan epilogue's instruction mix on gfx950 (independent chains, no MFMA).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 16
template <int MODE>
__global__ void k(double* out, int iters)
{
    double d[N]; float f[N]; unsigned u[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { d[i] = threadIdx.x * 1.0 + i; f[i] = threadIdx.x + i * 0.5f; u[i] = threadIdx.x * 7 + i; }
    const double lim = 123456.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (MODE == 0) asm volatile("v_min_f64 %0, %0, %1" : "+v"(d[i]) : "v"(lim));
            if (MODE == 1) asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[i]) : "v"(lim));
            if (MODE == 2) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
            if (MODE == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
            if (MODE == 4) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) % N]), "v"(u[(i + 2) % N]));
            if (MODE == 5) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(lim));
            if (MODE == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) % N]));
            if (MODE == 7) asm volatile("v_cmp_lt_f64 vcc, %0, %1" ::"v"(d[i]), "v"(lim) : "vcc");
            if (MODE == 8) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(u[(i + 1) % N]));
            if (MODE == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) % N]) : "vcc");
        }
    }
    double acc = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) acc += d[i] + f[i] + u[i];
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xFFFFF] = acc;
}
template <int MODE>
void run(const char* name, double* out, int cus, int wpc)
{
    const int iters = 4000;
    int blocks = cus * wpc / 4;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-16s waves/CU %2d: %.2f cycles per wave-instruction per SIMD (@2.4 GHz)\n", name, wpc, ms * 1e-3 * 2.4e9 / ((double)iters * N * wpc / 4));
}
int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    double* out; (void)hipMalloc(&out, 8 << 20);
    for (int wpc : {4, 16}) {
        run<0>("v_min_f64", out, p.multiProcessorCount, wpc);
        run<1>("v_max_f64", out, p.multiProcessorCount, wpc);
        run<2>("v_cvt_f32_f64", out, p.multiProcessorCount, wpc);
        run<3>("v_rcp_f32", out, p.multiProcessorCount, wpc);
        run<4>("v_and_or_b32", out, p.multiProcessorCount, wpc);
        run<5>("v_fma_f64", out, p.multiProcessorCount, wpc);
        run<6>("v_add_f32", out, p.multiProcessorCount, wpc);
        run<7>("v_cmp_lt_f64", out, p.multiProcessorCount, wpc);
        run<8>("v_mov_b32", out, p.multiProcessorCount, wpc);
        run<9>("v_cndmask_b32", out, p.multiProcessorCount, wpc);
    }
    return 0;
}
