// What the int8 matrix core leaves for the vector unit on gfx950: cycles per v_mfma_i32_16x16x64_i8 with K filler instructions
// of one kind behind every MFMA, 1 / 2 / 3 waves per SIMD (every wave runs the same stream).  Shader cycles by s_memtime.
// Round 4: the int8 scan's tile arithmetic (15 MFMAs + ~55 vector instructions) takes 0.57 ms where the MFMAs alone would
// take 0.27 -- which of its vector instructions hide beside an MFMA, and which serialise with it?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_i8_fill scripts/ubench_i8_fill.hip && /tmp/ubench_i8_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i32 __attribute__((ext_vector_type(4)));

template <int MODE, int K, bool MFMA>
__global__ __launch_bounds__(256) void k(double* out, long long* cyc, int iters)
{
    v4i32 a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)threadIdx.x, 8};
    v4i32 acc[5] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double d[8]; float f[8]; int u[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i] = threadIdx.x * 1.0 + i; f[i] = threadIdx.x + i * 0.5f + 1.0f; u[i] = threadIdx.x * 7 + i; }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 15; ++j) {
            if (MFMA) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[j % 5]) : "v"(a), "v"(b));
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const int i = (j * K + q) & 7;
                if (MODE == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 3) & 7]));
                if (MODE == 2) asm volatile("v_lshl_add_u32 %0, %0, 8, %1" : "+v"(u[i]) : "v"(u[(i + 3) & 7]));
                if (MODE == 3) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(u[i]));
                if (MODE == 4) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(d[(i + 3) & 7]));
                if (MODE == 5) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
                if (MODE == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
                if (MODE == 7) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[i]) : "v"(u[i]));
                if (MODE == 8) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(f[(i + 3) & 7]));
                if (MODE == 9) asm volatile("v_cmp_le_f32 vcc, %0, %1" ::"v"(f[i]), "v"(f[(i + 3) & 7]) : "vcc");
                if (MODE == 10) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 3) & 7]));
                if (MODE == 11) asm volatile("v_ashrrev_i32 %0, 8, %1" : "=v"(u[i]) : "v"(u[(i + 3) & 7]));
                if (MODE == 12) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 3) & 7]));
            }
        }
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += d[i] + f[i] + u[i];
#pragma unroll
    for (int i = 0; i < 5; ++i) s += acc[i][0] + acc[i][3];
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xFFFFF] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE, int K, bool MFMA>
double run(double* out, long long* cyc, int cus, int wps)
{
    const int iters = 400, blocks = cus * wps;
    hipLaunchKernelGGL((k<MODE, K, MFMA>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((k<MODE, K, MFMA>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    std::vector<long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double sum = 0;
    for (long long v : h) sum += (double)v;
    return sum / h.size() / (iters * 15.0) / wps;       // SIMD cycles per (MFMA + K fillers) of ONE wave's stream
}

template <int MODE>
void row(const char* name, double* out, long long* cyc, int cus)
{
    for (int wps : {1, 2, 3}) {
        printf("%-16s %d wave(s)/SIMD: per MFMA", name, wps);
        printf("  K=1 %5.1f", run<MODE, 1, true>(out, cyc, cus, wps));
        printf("  K=2 %5.1f", run<MODE, 2, true>(out, cyc, cus, wps));
        printf("  K=3 %5.1f", run<MODE, 3, true>(out, cyc, cus, wps));
        printf("  K=4 %5.1f", run<MODE, 4, true>(out, cyc, cus, wps));
        printf("  K=6 %5.1f", run<MODE, 6, true>(out, cyc, cus, wps));
        printf("   | without the MFMA: K=4 %5.1f (= %.1f per filler)\n", run<MODE, 4, false>(out, cyc, cus, wps), run<MODE, 4, false>(out, cyc, cus, wps) / 4.0);
    }
}

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    double* out; (void)hipMalloc(&out, 8 << 20);
    long long* cyc; (void)hipMalloc(&cyc, 1 << 20);
    const int cus = p.multiProcessorCount;
    printf("SIMD cycles (s_memtime) per v_mfma_i32_16x16x64_i8 + K fillers, one wave's stream, divided by the waves sharing the SIMD\n");
    for (int wps : {1, 2, 3}) printf("bare MFMA stream, %d wave(s)/SIMD: %.1f cycles per MFMA\n", wps, run<0, 1, true>(out, cyc, cus, wps));
    row<1>("v_add_f32", out, cyc, cus);
    row<2>("v_lshl_add_u32", out, cyc, cus);
    row<11>("v_ashrrev_i32", out, cyc, cus);
    row<7>("v_cvt_f32_i32", out, cyc, cus);
    row<8>("v_fma_f32", out, cyc, cus);
    row<9>("v_cmp_le_f32", out, cyc, cus);
    row<6>("v_rcp_f32", out, cyc, cus);
    row<3>("v_cvt_f64_i32", out, cyc, cus);
    row<4>("v_fma_f64", out, cyc, cus);
    row<10>("v_mul_f64", out, cyc, cus);
    row<12>("v_add_f64", out, cyc, cus);
    row<5>("v_cvt_f32_f64", out, cyc, cus);
    return 0;
}
