// Lab harness (not product): times scan_mfma_kernel ablations (ABL mask) and spectrum-store cache policies (AUX) on
// synthetic data, cfg2 shape (res 3600 -> 4 row classes), 262,144 items.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -o scripts/scan_lab scripts/scan_lab.hip
#include "../gr_baz_amd/csrc/music_kernels.hip.h"
#include <cstdio>
#include <cmath>
#include <vector>
#include <algorithm>
using namespace bazmusic;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Args {
    const double* dQ; const double2* dFB; float* spec; double* cand; uint32_t batch, res, nsteps;
};

template <int ABL, int AUX>
float run(const char* name, const Args& a, uint32_t nsplit, uint32_t nclass)
{
    constexpr int M = 4, NMAX = 2;
    const uint32_t rpc = ((a.batch + nclass - 1) / nclass + 63) / 64 * 64;
    const uint32_t groups = nclass * (rpc / 16);
    const uint32_t blocks = (groups / 4) * nsplit;
    ScanRefine rf; rf.Gs = nullptr; rf.TB = nullptr; rf.below = 0.0; rf.count = nullptr;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> t;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((scan_mfma_kernel<M, NMAX, true, true, ABL, AUX>), dim3(blocks), dim3(256), 0, 0,
                           a.dQ, a.dFB, a.spec, a.cand, a.batch, a.res, a.batch, nsplit, nclass, rpc, 0xFFFF0000u, 2u, rf);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0) t.push_back(ms);
    }
    CK(hipGetLastError());
    std::sort(t.begin(), t.end());
    printf("%-58s nclass=%u nsplit=%u : %.3f ms (min %.3f)  %.2f TB/s of spectrum\n", name, nclass, nsplit, t[t.size() / 2], t[0],
           (double)a.batch * a.res * 4 / (t[t.size() / 2] * 1e-3) / 1e12);
    fflush(stdout);
    return t[t.size() / 2];
}

int main()
{
    const uint32_t batch = 262144, res = 3600, nsteps = 57, KS = 4;
    std::vector<double> hQ((size_t)16 * batch), hFB((size_t)(nsteps + 2) * 2 * KS * 64 * 2);
    // every item of a 64-item block gets the same q (coherent scene: the top-n gate then skips most steps), plus noise
    for (size_t e = 0; e < 16; ++e)
        for (size_t i = 0; i < batch; ++i) hQ[e * batch + i] = 0.1 + 0.9 * (((e * 7919u + (i / 4096) * 104729u) * 2654435761u) % 1000) / 1000.0 + 1e-6 * (i % 97);
    for (size_t i = 0; i < hFB.size(); ++i) hFB[i] = 0.1 + ((i * 40503u) % 997) / 997.0;
    double *dQ, *cand; double2* dFB; float* spec;
    CK(hipMalloc(&dQ, hQ.size() * 8)); CK(hipMalloc(&dFB, hFB.size() * 8));
    CK(hipMemcpy(dQ, hQ.data(), hQ.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dFB, hFB.data(), hFB.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&spec, (size_t)batch * res * 4 + 4096));
    CK(hipMalloc(&cand, (size_t)batch * 64 * 2 * 8));
    Args a{dQ, dFB + (size_t)2 * KS * 64, spec, cand, batch, res, nsteps};
#define R(ABL, AUX, name) run<ABL, AUX>(name, a, 2, 4)
    R(0, 19, "full, sc0 sc1 nt");
    R(0, 0, "full, plain stores");
    R(256, 0, "full, plain stores, table loads nt");
    R(256, 19, "full, sc0 sc1 nt stores, table loads nt");
    R(8 | 2 | 4, 19, "stores + staging + barriers only, sc0 sc1 nt");
    R(8 | 2 | 4, 0, "stores + staging + barriers only, plain");
    R(8 | 2 | 4 | 128, 0, "stores + barriers only (no table loads), plain");
    R(8 | 2 | 4 | 128, 19, "stores + barriers only (no table loads), sc0 sc1 nt");
    R(8 | 2 | 4 | 256, 0, "stores + staging (nt loads) + barriers, plain");
    R(2 | 256, 0, "no top-n, plain stores, table loads nt");
    R(2, 19, "no top-n, sc0 sc1 nt");
    return 0;
}
