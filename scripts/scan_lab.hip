// Lab harness (not product): times scan_mfma_kernel ablations (ABL mask) on random data, cfg2 shape.
#include "../gr_baz_amd/csrc/music_kernels.hip.h"
#include <cstdio>
#include <cmath>
#include <vector>
using namespace bazmusic;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int ABL>
float run(const char* name, const double* dQ, const double2* dFB, float* spec, double* cand,
          uint32_t batch, uint32_t res, uint32_t nsteps, uint32_t nsplit)
{
    constexpr int M = 4, NMAX = 2;
    const uint32_t groups = (batch + 15) / 16;
    const uint32_t blocks = ((groups + 3) / 4) * nsplit;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((scan_mfma_kernel<M, NMAX, true, true, ABL>), dim3(blocks), dim3(256), 0, 0,
                           dQ, dFB, spec, cand, batch, res, batch, nsteps, nsplit, groups, 0xFFFF0000u);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    printf("%-40s nsplit=%2u blocks=%5u : %.3f ms\n", name, nsplit, blocks, best);
    return best;
}

int main()
{
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const uint32_t batch = 65536, res = 3600, nsteps = 57, KS = 4;
    std::vector<double> hQ((size_t)16 * batch), hFB((size_t)nsteps * 2 * KS * 64 * 2);
    for (size_t i = 0; i < hQ.size(); ++i) hQ[i] = 0.1 + 0.9 * ((i * 2654435761u) % 1000) / 1000.0;
    for (size_t i = 0; i < hFB.size(); ++i) hFB[i] = 0.1 + ((i * 40503u) % 997) / 997.0;
    double *dQ, *cand; double2* dFB; float* spec;
    CK(hipMalloc(&dQ, hQ.size() * 8)); CK(hipMalloc(&dFB, hFB.size() * 8));
    CK(hipMemcpy(dQ, hQ.data(), hQ.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dFB, hFB.data(), hFB.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&spec, (size_t)batch * res * 4));
    CK(hipMalloc(&cand, (size_t)batch * 64 * 2 * 8));
#define R(ABL, name, ns) run<ABL>(name, dQ, dFB, spec, cand, batch, res, nsteps, ns)
    R(0, "full", 8);
    R(1, "no stores", 8);
    R(2, "no top-n", 8);
    R(3, "no stores, no top-n", 8);
    R(7, "no stores/top-n/cvt-rcp", 8);
    R(8, "full, no MFMA", 8);
    R(16, "full, asm saddr stores (untracked)", 8);
    R(18, "asm saddr stores, no top-n", 8);
    R(0, "full ns=1", 1);
    R(0, "full ns=2", 2);
    R(0, "full ns=4", 4);
    R(0, "full ns=16", 16);
    return 0;
}
