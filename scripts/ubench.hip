// Micro-benchmarks that decide the scan-kernel design on gfx950 (not part of the product):
//   1. v_fma_f64 throughput (VALU fp64)
//   2. v_mfma_f64_16x16x4_f64 and v_mfma_f64_4x4x4_4b_f64 throughput (matrix fp64)
//   3. scalar-load streaming of a 460 KB table (s_load_dwordx16) -- what the lane=item scan does
//   4. vector-load streaming of the same table through L1/L2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

__global__ void k_fma64(double* out, int iters)
{
    double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double b = 1.0000001, c = 0.5;
    for (int i = 0; i < iters; ++i) {
        a0 = fma(a0, b, c); a1 = fma(a1, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c);
        a4 = fma(a4, b, c); a5 = fma(a5, b, c); a6 = fma(a6, b, c); a7 = fma(a7, b, c);
    }
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xFFFFF] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ void k_mfma64_16(double* out, int iters)
{
    v4f64 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    v4f64 c = c0 + c1 + c2 + c3;
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xFFFFF] = c[0] + c[1] + c[2] + c[3];
}

__global__ void k_mfma64_4(double* out, int iters)
{
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    }
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xFFFFF] = c0 + c1 + c2 + c3;
}

// mixed: MFMA + independent VALU fp64 fma in the same wave (do the pipes overlap?)
__global__ void k_mix(double* out, int iters)
{
    v4f64 c0 = {0, 0, 0, 0}, c1 = c0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    double f0 = a, f1 = a + 1, f2 = a + 2, f3 = a + 3;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        f0 = fma(f0, b, 0.5); f1 = fma(f1, b, 0.5); f2 = fma(f2, b, 0.5); f3 = fma(f3, b, 0.5);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        f0 = fma(f0, b, 0.5); f1 = fma(f1, b, 0.5); f2 = fma(f2, b, 0.5); f3 = fma(f3, b, 0.5);
    }
    v4f64 c = c0 + c1;
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xFFFFF] = c[0] + c[1] + c[2] + c[3] + f0 + f1 + f2 + f3;
}

// each wave streams the whole table with scalar loads (16 doubles per "bin"), 16 fma per bin
__global__ void k_sload(const double* __restrict__ F, int nbins, double* out, int fmas)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    double q = threadIdx.x * 1e-3, acc0 = 0, acc1 = 0;
    for (int b = wave; b < nbins; b += nw) {
        const double* __restrict__ Fb = F + (size_t)b * 16;
        if (fmas) {
#pragma unroll
            for (int e = 0; e < 16; e += 2) { acc0 = fma(q, Fb[e], acc0); acc1 = fma(q, Fb[e + 1], acc1); }
        } else {
            acc0 += Fb[0] + Fb[15];
        }
    }
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xFFFFF] = acc0 + acc1;
}

// same bytes through the vector path: every lane loads the same 16 doubles (broadcast from L1)
__global__ void k_vload(const double* __restrict__ F, int nbins, double* out)
{
    const int wave = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6;
    const int lane = threadIdx.x & 63;
    double acc = 0;
    // lanes cooperatively load 64 doubles = 4 bins per instruction
    for (int b = wave * 4; b < nbins; b += nw * 4) acc += F[(size_t)b * 16 + lane];
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xFFFFF] = acc;
}

template <typename Fn>
float time_ms(Fn fn, int reps = 5)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    fn();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) fn();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device %s CUs %d clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    const int CUS = p.multiProcessorCount;
    double* out; CK(hipMalloc(&out, sizeof(double) * 4096 * 1024));
    const int iters = 20000;
    for (int wpc : {4, 8, 16}) {   // waves per CU
        int blocks = CUS * wpc / 4;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_fma64, dim3(blocks), dim3(256), 0, 0, out, iters); });
        double fl = 2.0 * 8 * iters * (double)blocks * 256;
        printf("fma64   waves/CU %2d: %.3f ms  %.1f TFLOP/s  (%.2f cyc/wave-instr/SIMD @2.4GHz)\n", wpc, ms, fl / ms / 1e9,
               ms * 1e-3 * 2.4e9 / (8.0 * iters * wpc / 4));
        ms = time_ms([&] { hipLaunchKernelGGL(k_mfma64_16, dim3(blocks), dim3(256), 0, 0, out, iters); });
        fl = 2.0 * 16 * 16 * 4 * 4 * iters * (double)blocks * 4;
        printf("mfma16  waves/CU %2d: %.3f ms  %.1f TFLOP/s  (%.2f cyc/mfma/SIMD @2.4GHz)\n", wpc, ms, fl / ms / 1e9,
               ms * 1e-3 * 2.4e9 / (4.0 * iters * wpc / 4));
        ms = time_ms([&] { hipLaunchKernelGGL(k_mfma64_4, dim3(blocks), dim3(256), 0, 0, out, iters); });
        fl = 2.0 * 4 * 4 * 4 * 4 * 4 * iters * (double)blocks * 4;
        printf("mfma4b  waves/CU %2d: %.3f ms  %.1f TFLOP/s  (%.2f cyc/mfma/SIMD @2.4GHz)\n", wpc, ms, fl / ms / 1e9,
               ms * 1e-3 * 2.4e9 / (4.0 * iters * wpc / 4));
        ms = time_ms([&] { hipLaunchKernelGGL(k_mix, dim3(blocks), dim3(256), 0, 0, out, iters); });
        printf("mix 2mfma+8fma waves/CU %2d: %.3f ms  (%.2f cyc per loop-iter/SIMD; sum-of-parts would be 2*mfma+8*fma)\n", wpc, ms,
               ms * 1e-3 * 2.4e9 / (1.0 * iters * wpc / 4));
    }
    // table streaming
    const int nbins = 3600;
    std::vector<double> hF((size_t)nbins * 16, 1.0);
    double* dF; CK(hipMalloc(&dF, hF.size() * 8)); CK(hipMemcpy(dF, hF.data(), hF.size() * 8, hipMemcpyHostToDevice));
    // proper: table of nbins; each block's 4 waves split the bins; repeat R passes inside via grid size
    for (int bpc : {1, 2, 4, 8}) {
        int blocks = CUS * bpc * 16;   // 16 rounds of blocks
        for (int fm : {0, 1}) {
            float ms = time_ms([&] { hipLaunchKernelGGL(k_sload, dim3(blocks), dim3(256), 0, 0, dF, nbins, out, fm); });
            double bytes = (double)blocks * nbins * 128.0;
            printf("sload fm=%d blocks %5d (%d/CU resident target): %.3f ms  %.2f TB/s scalar-path  (%.1f B/clk/CU)\n", fm, blocks, bpc, ms,
                   bytes / ms / 1e9, bytes / (ms * 1e-3 * 2.4e9) / CUS);
        }
    }
    {
        int blocks = CUS * 4 * 16;
        float ms = time_ms([&] { hipLaunchKernelGGL(k_vload, dim3(blocks), dim3(256), 0, 0, dF, nbins, out); });
        double bytes = (double)blocks * nbins * 128.0;
        printf("vload blocks %5d: %.3f ms  %.2f TB/s vector-path (%.1f B/clk/CU)\n", blocks, ms, bytes / ms / 1e9,
               bytes / (ms * 1e-3 * 2.4e9) / CUS);
    }
    return 0;
}
