// Lab harness (not product): can the scan's two resources -- SIMD issue time (fp64 MFMAs + epilogue) and the spectrum store
// path -- be busy at the same time when NOTHING couples them?  The compute-only form (ABL = 1: everything but the stores) and
// the store-only form (ABL = 8|2|4: stores + staging, 38 registers) of scan_mfma_kernel run (a) alone and (b) side by side
// as two launches on two streams, the compute launch padded with unused dynamic LDS to 3 / 2 workgroups per CU so that the
// store launch finds register file and wave slots beside it.  If (b) takes ~max of the two, a producer / consumer split of
// the scan (compute waves -> LDS ring -> store waves) has something to win; if it takes what the shipped kernel takes, the
// coupling is in the hardware (clocks, fabric), not in the waves.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -o scripts/scan_decouple_lab scripts/scan_decouple_lab.hip
#include "../gr_baz_amd/csrc/music_kernels.hip.h"
#include <cstdio>
#include <cmath>
#include <vector>
#include <algorithm>
using namespace bazmusic;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Args { const double* dQ; const double2* dFB; float* spec; double* cand; double* cand2; uint32_t batch, res; };

template <int ABL>
void launch(const Args& a, hipStream_t s, size_t pad, double* cand)
{
    constexpr int M = 4, NMAX = 2;
    const uint32_t nclass = 4, nsplit = 1;
    const uint32_t rpc = ((a.batch + nclass - 1) / nclass + 63) / 64 * 64;
    const uint32_t blocks = (nclass * (rpc / 16) / 4) * nsplit;
    ScanRefine rf; rf.Gs = nullptr; rf.TB = nullptr; rf.below = 0.0; rf.count = nullptr; rf.A2 = nullptr;
    hipLaunchKernelGGL((scan_mfma_kernel<M, NMAX, true, true, ABL, 19>), dim3(blocks), dim3(256), pad, s,
                       a.dQ, a.dFB, a.spec, cand, a.batch, a.res, a.batch, nsplit, nclass, rpc, 0xFFFF0000u, 2u, rf);
}

template <typename F>
float timeit(const char* name, double bytes, F&& f)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> t;
    for (int rep = 0; rep < 8; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        f();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 1) t.push_back(ms);
    }
    CK(hipGetLastError());
    std::sort(t.begin(), t.end());
    printf("%-92s %.3f ms (min %.3f)  %.2f TB/s of spectrum\n", name, t[t.size() / 2], t[0], bytes / (t[t.size() / 2] * 1e-3) / 1e12);
    fflush(stdout);
    return t[t.size() / 2];
}

int main()
{
    const uint32_t batch = 262144, res = 3600, nsteps = 57, KS = 4;
    std::vector<double> hQ((size_t)16 * batch), hFB((size_t)(nsteps + 2) * 2 * KS * 64 * 2);
    // smooth operands: d rises with the bin, so the top-n gate fires on the first step of a row only (as on a coherent stream)
    for (size_t e = 0; e < 16; ++e)
        for (size_t i = 0; i < batch; ++i) hQ[e * batch + i] = 0.5 + 0.01 * e;
    for (size_t sti = 0; sti < nsteps + 2; ++sti)
        for (size_t s = 0; s < KS; ++s)
            for (size_t t = 0; t < 4; ++t)
                for (size_t lane = 0; lane < 64; ++lane) {
                    const double bin = 64.0 * ((double)sti - 1.0) + 4.0 * (lane & 15) + t;
                    hFB[(((sti * 2 * KS + 2 * s + (t >> 1)) * 64 + lane) * 2) + (t & 1)] = 1.0 + 1e-3 * bin + 0.01 * (4 * s + (lane >> 4));
                }
    double *dQ, *cand, *cand2; double2* dFB; float* spec;
    CK(hipMalloc(&dQ, hQ.size() * 8)); CK(hipMalloc(&dFB, hFB.size() * 8));
    CK(hipMemcpy(dQ, hQ.data(), hQ.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dFB, hFB.data(), hFB.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&spec, (size_t)batch * res * 4 + 4096));
    CK(hipMalloc(&cand, (size_t)batch * 4 * 2 * 8)); CK(hipMalloc(&cand2, (size_t)batch * 4 * 2 * 8));
    Args a{dQ, dFB + (size_t)2 * KS * 64, spec, cand, cand2, batch, res};
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t f1, f2, j1, j2; CK(hipEventCreate(&f1)); CK(hipEventCreate(&f2)); CK(hipEventCreate(&j1)); CK(hipEventCreate(&j2));
    const double B = (double)batch * res * 4;
    const size_t pads[3] = {0, 24 * 1024, 56 * 1024};      // + 16 KiB static: 4 (register limit) / 3 / 2 workgroups per CU
    const char* padname[3] = {"4 per CU", "3 per CU", "2 per CU"};
    for (int rep = 0; rep < 2; ++rep) {
        timeit("shipped kernel (compute + stores in every wave)", B, [&] { launch<0>(a, 0, 0, cand); });
        for (int p = 0; p < 3; ++p) {
            char nm[128]; snprintf(nm, sizeof nm, "compute only (no stores), workgroups %s", padname[p]);
            timeit(nm, B, [&] { launch<1>(a, 0, pads[p], cand); });
        }
        timeit("stores only (38 registers)", B, [&] { launch<(8 | 2 | 4)>(a, 0, 0, cand2); });
        for (int p = 0; p < 3; ++p) {
            char nm[160]; snprintf(nm, sizeof nm, "compute only (%s) and stores only, two launches on two streams", padname[p]);
            timeit(nm, B, [&] {
                CK(hipEventRecord(f1, 0));
                CK(hipStreamWaitEvent(s1, f1, 0)); CK(hipStreamWaitEvent(s2, f1, 0));
                launch<1>(a, s1, pads[p], cand);
                launch<(8 | 2 | 4)>(a, s2, 0, cand2);
                CK(hipEventRecord(j1, s1)); CK(hipEventRecord(j2, s2));
                CK(hipStreamWaitEvent(0, j1, 0)); CK(hipStreamWaitEvent(0, j2, 0));
            });
        }
    }
    return 0;
}
