// Lab micro-benchmark (not product; TIMING ONLY -- operands are filler, no result is checked): what would cfg3's scan
// (16,384 items x 36,000 bins, m = 8: d = <q(item), t(bin)> over 64 real terms, spectrum = 1 / d as float) cost if the
// bulk of its tiles ran on the bf16 matrix core with both operands split into three bf16 parts (six cross products:
// K = 6 x 64 = 384 per value, f32 accumulation; DESIGN.md 10 item 5, tests/lab/f32_bulk_study.py for the accuracy)?
// Today: scan_mfma_kernel on the fp64 matrix core, 1.24 ms (77 % of its peak).
//   wave  = 32 items x 32 bins per tile, v_mfma_f32_32x32x16_bf16 x NSTEP (24 for the three-part split, 16 for a
//           two-part f16-style split with all four cross products, 8 = one plain bf16 pass, for the slope)
//   A     = the items' parts, in registers for the whole bin range (4 VGPRs per step)
//   B     = the table's parts, [tile][step][lane] x 16 B, staged per tile through double-buffered LDS by the 4 waves of
//           a workgroup (they share the bins and differ in the items)
//   block = item block (128 items) x bin range; ranges are dealt so that a range stays on one XCD (blockIdx % 8) and its
//           slice of the table image (1.7 MB of 27.6) lives in that XCD's L2
//   store = v_rcp_f32 of the 16 accumulators, 16 dword stores per tile (two 128-B runs per instruction); STORE = false
//           leaves one conditional store so that the arithmetic stays
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/ubench_bf16x3_scan scripts/ubench_bf16x3_scan.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef short v8s __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

constexpr uint32_t ITEMS = 16384, RES = 36000, TILE = 32, NTILES = RES / TILE;      // 1,125 bin tiles
constexpr uint32_t RANGES = 16, WG_ITEMS = 128;

template <int NSTEP, bool STORE>
__global__ __launch_bounds__(256) void scan_bf16_parts(const v4u* __restrict__ A, const v4u* __restrict__ B, float* __restrict__ spec)
{
    extern __shared__ v4u lds[];                                   // 2 x NSTEP x 64 units of 16 B
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t range = blockIdx.x % RANGES, iblk = blockIdx.x / RANGES;
    const uint32_t t0 = (uint32_t)(((uint64_t)NTILES * range) / RANGES), t1 = (uint32_t)(((uint64_t)NTILES * (range + 1)) / RANGES);
    const uint32_t item0 = iblk * WG_ITEMS + wave * 32;
    v8s a[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) a[s] = __builtin_bit_cast(v8s, A[((size_t)(item0 / 32) * NSTEP + s) * 64 + lane]);
    constexpr int UNITS = NSTEP * 64;                              // 16-B units per tile
    auto stage = [&](uint32_t tile, int buf) {
        for (int u = threadIdx.x; u < UNITS; u += 256) lds[buf * UNITS + u] = B[(size_t)tile * UNITS + u];
    };
    stage(t0, 0);
    __syncthreads();
    for (uint32_t t = t0; t < t1; ++t) {
        const int buf = (t - t0) & 1;
        if (t + 1 < t1) stage(t + 1, buf ^ 1);
        v16f acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const v8s b = __builtin_bit_cast(v8s, lds[buf * UNITS + s * 64 + lane]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b, acc, 0, 0, 0);
        }
        // C/D: col = lane & 31 (bin), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (item)
        const uint32_t bin = t * TILE + (lane & 31);
        float sink = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t row = item0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const float v = __builtin_amdgcn_rcpf(acc[r]);
            if constexpr (STORE) __builtin_nontemporal_store(v, spec + (size_t)row * RES + bin);
            else sink += v;
        }
        if constexpr (!STORE)
            if (sink == 12345.678f) spec[(size_t)item0 * RES + bin] = sink;
        __syncthreads();
    }
}

template <typename F>
float timeit(const char* name, F&& f)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> t;
    for (int rep = 0; rep < 7; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        f();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 1) t.push_back(ms);
    }
    CK(hipGetLastError());
    std::sort(t.begin(), t.end());
    const double vals = (double)ITEMS * RES;
    printf("%-74s %.3f ms (min %.3f)  %.2f TB/s of spectrum, %.2e items/s\n", name, t[t.size() / 2], t[0],
           vals * 4 / (t[t.size() / 2] * 1e-3) / 1e12, ITEMS / (t[t.size() / 2] * 1e-3));
    fflush(stdout);
    return t[t.size() / 2];
}

template <int NSTEP>
void run(const v4u* dA, const v4u* dB, float* spec)
{
    const dim3 grid((ITEMS / WG_ITEMS) * RANGES), block(256);
    const size_t lds = (size_t)2 * NSTEP * 64 * 16;
    char nm[160];
    snprintf(nm, sizeof nm, "K = %3d (%2d x v_mfma_f32_32x32x16_bf16 per 32 x 32 tile), rcp, stores", NSTEP * 16, NSTEP);
    timeit(nm, [&] { hipLaunchKernelGGL((scan_bf16_parts<NSTEP, true>), grid, block, lds, 0, dA, dB, spec); });
    snprintf(nm, sizeof nm, "K = %3d (%2d x v_mfma_f32_32x32x16_bf16 per 32 x 32 tile), rcp, no stores", NSTEP * 16, NSTEP);
    timeit(nm, [&] { hipLaunchKernelGGL((scan_bf16_parts<NSTEP, false>), grid, block, lds, 0, dA, dB, spec); });
}

int main()
{
    constexpr int MAXSTEP = 24;
    const size_t nA = (size_t)(ITEMS / 32) * MAXSTEP * 64, nB = (size_t)NTILES * MAXSTEP * 64;      // 16-B units
    std::vector<uint32_t> hA(nA * 4), hB(nB * 4);
    uint32_t x = 12345u;
    auto bf = [&]() { x = x * 1664525u + 1013904223u; return (uint32_t)(0x3F00u + ((x >> 20) & 0x7Fu)); };     // bf16 in [0.5, 1)
    for (auto& w : hA) w = bf() | (bf() << 16);
    for (auto& w : hB) w = bf() | (bf() << 16);
    v4u *dA, *dB; float* spec;
    CK(hipMalloc((void**)&dA, nA * 16)); CK(hipMalloc((void**)&dB, nB * 16)); CK(hipMalloc((void**)&spec, (size_t)ITEMS * RES * 4));
    CK(hipMemcpy(dA, hA.data(), nA * 16, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), nB * 16, hipMemcpyHostToDevice));
    CK(hipMemset(spec, 0, (size_t)ITEMS * RES * 4));
    printf("# cfg3's scan shape on the bf16 matrix core, timing only: %u items x %u bins, %u workgroups of 128 items x 1/%u of the bins\n",
           ITEMS, RES, (ITEMS / WG_ITEMS) * RANGES, RANGES);
    run<24>(dA, dB, spec);
    run<16>(dA, dB, spec);
    run<8>(dA, dB, spec);
    float h[4]; CK(hipMemcpy(h, spec + 12345, 16, hipMemcpyDeviceToHost));
    printf("# sample outputs %g %g %g %g (filler operands)\n", h[0], h[1], h[2], h[3]);
    return 0;
}
