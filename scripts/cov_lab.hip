// Lab harness (not product): what bounds the m = 4 covariance stream (cov4_x4_kernel / cov4_evd_kernel's inner loop)?
// Variants of the chunk loop on synthetic data, 262,144 items x 8 KiB, with the shader clock observed by the kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -o scripts/cov_lab scripts/cov_lab.hip
#include "../gr_baz_amd/csrc/music_kernels.hip.h"
#include <cstdio>
#include <cmath>
#include <vector>
#include <algorithm>
using namespace bazmusic;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ unsigned long long lab_clk[4];

__device__ __forceinline__ float dpp_xor1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ float dpp_xor2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }   // quad_perm [2,3,0,1]
__device__ __forceinline__ float dpp_half_mirror(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)); }   // row_half_mirror

// V: 0 shipped loop; 1 no ds_write; 2 no ds_read; 3 fp32 staging; 4 DPP transpose (no LDS); 5 no MFMA; 6 fp32 staging, one buffer per ring slot (no second fence)
template <int V>
__global__ __launch_bounds__(256) void cov_lab(const float* __restrict__ in, double2* __restrict__ R, uint32_t batch, uint32_t K)
{
    constexpr int RSD = 34;
    __shared__ double stage[4][(V == 6) ? 9 : 2][8 * RSD];
    __shared__ double gram[4][2][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t chunks = K >> 5;
    const int wcol = lane >> 1, wrow = 4 * (lane & 1);
    const int ri = lane & 3, rh = (lane >> 2) & 1, rw = (lane >> 3) & 1, rk = lane >> 4;
    const int p_off = (4 * rh + ri) * RSD + 8 * rk + 4 * rw;
    const int q_off = (4 * (1 - rh) + ri) * RSD + 8 * rk + 4 * rw;
    // fp32 staging: row stride 36 floats (144 B: 16-B aligned operand reads)
    constexpr int RSF = 36;
    float* const stf = reinterpret_cast<float*>(&stage[wave][0][0]);
    const int pf_off = (4 * rh + ri) * RSF + 8 * rk + 4 * rw;
    const int qf_off = (4 * (1 - rh) + ri) * RSF + 8 * rk + 4 * rw;
    double* const g1 = gram[wave][0];
    double* const g2 = gram[wave][1];
    const double dK = (double)K;
    const uint32_t groups = chunks >> 3;
    const uint32_t stride = gridDim.x * 4;
    uint32_t item = blockIdx.x * 4 + wave;
    if (item >= batch) return;
    __shared__ double gram4[4][4][128];
    uint32_t nepi = 0;
    const double rK = 1.0 / dK;
    int eo[4];
    {   // offsets of the four Gram entries R_ab needs, in [D1 | D2] (see the epilogue below)
        const int a = (lane >> 2) & 3, b = lane & 3;
        auto GO = [&](int x, int y) -> int {
            if ((x >> 2) == (y >> 2)) return (y & 3) + 4 * (x >> 2) + 16 * (x & 3);
            if (x > y) { const int t = x; x = y; y = t; }
            return 64 + (y - 4) + 16 * x;
        };
        eo[0] = GO(2 * a, 2 * b); eo[1] = GO(2 * a + 1, 2 * b + 1); eo[2] = GO(2 * a + 1, 2 * b); eo[3] = GO(2 * a, 2 * b + 1);
    }
    unsigned long long c0 = 0, w0 = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) { c0 = __builtin_readcyclecounter(); w0 = wall_clock64(); }
    // DPP variant: lane = 4g + x loads column 4*(g>>1) + x, half g&1
    const int dg = lane >> 2, dx = lane & 3;
    const int lane_off = (V == 4) ? (2 * (4 * (dg >> 1) + dx) + (dg & 1)) : lane;
    const v4f32* __restrict__ src = reinterpret_cast<const v4f32*>(in + (size_t)item * K * 8) + lane_off;
    v4f32 pf[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) pf[u] = __builtin_nontemporal_load(src + (size_t)u * 64);
    for (; item < batch; item += stride) {
        const uint32_t nitem = (item + stride < batch) ? item + stride : item;
        const v4f32* __restrict__ nsrc = reinterpret_cast<const v4f32*>(in + (size_t)nitem * K * 8) + lane_off;
        double a1 = 0.0, b1 = 0.0, a2 = 0.0, b2 = 0.0;
        for (uint32_t cg = 0; cg < groups; ++cg) {
            const v4f32* __restrict__ rearm = (cg + 1 < groups) ? src + (size_t)(cg + 1) * 8 * 64 : nsrc;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v4f64 P, Q;
                if constexpr (V == 7 || V == 8) {
                    for (int j = 0; j < 4; ++j) { P[j] = (double)pf[u][j]; Q[j] = (double)pf[u][3 - j]; }
                    asm volatile("" ::: "memory");
                    pf[u] = __builtin_nontemporal_load(rearm + (size_t)u * 64);
                } else if constexpr (V == 3 || V == 6) {
                    float* __restrict__ T = stf + ((V == 6) ? u : (u & 1)) * (8 * RSF);
#pragma unroll
                    for (int j = 0; j < 4; ++j) T[(wrow + j) * RSF + wcol] = pf[u][j];
                    asm volatile("" ::: "memory");
                    pf[u] = __builtin_nontemporal_load(rearm + (size_t)u * 64);
                    wave_lds_fence();
                    const v4f32 p = *reinterpret_cast<const v4f32*>(T + pf_off);
                    const v4f32 q = *reinterpret_cast<const v4f32*>(T + qf_off);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { P[j] = (double)p[j]; Q[j] = (double)q[j]; }
                } else if constexpr (V == 4) {
                    float f0 = pf[u][0], f1 = pf[u][1], f2 = pf[u][2], f3 = pf[u][3];
                    asm volatile("" ::: "memory");
                    pf[u] = __builtin_nontemporal_load(rearm + (size_t)u * 64);
                    const bool o1 = dx & 1, o2 = dx & 2;
                    {   // stage A: 2x2 transposes between lanes x and x^1
                        const float s01 = o1 ? f0 : f1, s23 = o1 ? f2 : f3;
                        const float r01 = dpp_xor1(s01), r23 = dpp_xor1(s23);
                        if (o1) { f0 = r01; f2 = r23; } else { f1 = r01; f3 = r23; }
                    }
                    {   // stage B: between lanes x and x^2
                        const float s02 = o2 ? f0 : f2, s13 = o2 ? f1 : f3;
                        const float r02 = dpp_xor2(s02), r13 = dpp_xor2(s13);
                        if (o2) { f0 = r02; f1 = r13; } else { f2 = r02; f3 = r13; }
                    }
                    P[0] = (double)f0; P[1] = (double)f1; P[2] = (double)f2; P[3] = (double)f3;
                    Q[0] = (double)dpp_half_mirror(f0); Q[1] = (double)dpp_half_mirror(f1);
                    Q[2] = (double)dpp_half_mirror(f2); Q[3] = (double)dpp_half_mirror(f3);
                } else {
                    double* __restrict__ T = stage[wave][u & 1];
                    if constexpr (V != 1) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) T[(wrow + j) * RSD + wcol] = (double)pf[u][j];
                    }
                    v4f32 keep = pf[u];
                    asm volatile("" ::: "memory");
                    pf[u] = __builtin_nontemporal_load(rearm + (size_t)u * 64);
                    wave_lds_fence();
                    if constexpr (V == 2) {
                        for (int j = 0; j < 4; ++j) { P[j] = (double)keep[j]; Q[j] = (double)keep[3 - j]; }
                    } else {
                        P = *reinterpret_cast<const v4f64*>(T + p_off);
                        Q = *reinterpret_cast<const v4f64*>(T + q_off);
                        if constexpr (V == 1) { P[0] += (double)keep[0]; Q[1] += (double)keep[1]; P[2] += (double)keep[2]; Q[3] += (double)keep[3]; }
                    }
                }
                if constexpr (V == 5 || V == 7 || V == 8) {
                    a1 += P[0] * Q[1]; b1 += P[2] * Q[3]; a2 += P[1] * Q[0]; b2 += Q[2] * P[3];
                } else {
                    a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(P[0], P[0], a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f64_4x4x4f64(P[0], Q[0], a2, 0, 0, 0);
                    b1 = __builtin_amdgcn_mfma_f64_4x4x4f64(P[1], P[1], b1, 0, 0, 0);
                    b2 = __builtin_amdgcn_mfma_f64_4x4x4f64(P[1], Q[1], b2, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(P[2], P[2], a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f64_4x4x4f64(P[2], Q[2], a2, 0, 0, 0);
                    b1 = __builtin_amdgcn_mfma_f64_4x4x4f64(P[3], P[3], b1, 0, 0, 0);
                    b2 = __builtin_amdgcn_mfma_f64_4x4x4f64(P[3], Q[3], b2, 0, 0, 0);
                }
                if constexpr (V != 4 && V != 6 && V != 7 && V != 8) wave_lds_fence();
            }
            if constexpr (V == 6) wave_lds_fence();
        }
        src = nsrc;
        if constexpr (V == 8) { if (a1 + b1 + a2 + b2 == 1.2345) R[item] = make_double2(a1, b1); continue; }
        if constexpr (V == 9 || V == 10 || V == 11 || V >= 100) {
            constexpr int NS = (V == 9) ? 1 : 4;          // items per epilogue
            double* const gg = &gram4[wave][0][0];
            const uint32_t slot = (V == 9) ? 0u : (nepi & 3u);
            gg[slot * 128 + lane] = a1 + b1;
            gg[slot * 128 + 64 + lane] = a2 + b2;
            ++nepi;
            if (slot == NS - 1 || item + stride >= batch) {
                wave_lds_fence();
                const uint32_t sl = (V == 9) ? 0u : (uint32_t)(lane >> 4);
                if ((V == 9) ? (lane < 16) : (sl <= slot)) {
                    const double* gs = gg + sl * 128;
                    const double re = (gs[eo[0]] + gs[eo[0] + 8]) + (gs[eo[1]] + gs[eo[1] + 8]);
                    const double im = (gs[eo[2]] + gs[eo[2] + 8]) - (gs[eo[3]] + gs[eo[3] + 8]);
                    const uint32_t it_s = item - (slot - sl) * stride;
                    if constexpr (V == 11) { if (re == 1.2345) R[(size_t)it_s * 16 + (lane & 15)] = make_double2(re * rK, im * rK); }
                    else if constexpr (V >= 100) {
                        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)R, 0, 0x7FFFFFFF, 0x00020000);
                        v4u32 d; double2 val = make_double2(re * rK, im * rK);
                        __builtin_memcpy(&d, &val, 16);
                        __builtin_amdgcn_raw_buffer_store_b128(d, rs, (it_s * 16 + (lane & 15)) * 16, 0, V - 100);
                    }
                    else R[(size_t)it_s * 16 + (lane & 15)] = make_double2(re * rK, im * rK);
                }
                wave_lds_fence();
            }
            continue;
        }
        g1[lane] = a1 + b1;
        g2[lane] = a2 + b2;
        wave_lds_fence();
        if (lane < 16) {
            const int a = lane >> 2, b = lane & 3;
            auto G = [&](int x, int y) -> double {
                if ((x >> 2) == (y >> 2)) {
                    const int hh = x >> 2, o = (y & 3) + 4 * hh + 16 * (x & 3);
                    return g1[o] + g1[o + 8];
                }
                if (x > y) { const int t = x; x = y; y = t; }
                const int yy = (V == 4) ? 3 - (y - 4) : (y - 4);     // DPP variant: the mirror hands row 3-j to lane j
                const int o = yy + 16 * x;
                return g2[o] + g2[o + 8];
            };
            const double re = G(2 * a, 2 * b) + G(2 * a + 1, 2 * b + 1);
            const double im = G(2 * a + 1, 2 * b) - G(2 * a, 2 * b + 1);
            R[(size_t)item * 16 + lane] = make_double2(re / dK, im / dK);
        }
        wave_lds_fence();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        lab_clk[0] = __builtin_readcyclecounter() - c0;
        lab_clk[1] = wall_clock64() - w0;
    }
}

template <int V>
void run(const char* name, const float* din, double2* dR, uint32_t batch, uint32_t K, int blocks_per_cu, std::vector<double2>* out = nullptr)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> t;
    const uint32_t blocks = 256u * blocks_per_cu;
    for (int rep = 0; rep < 42; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(cov_lab<V>, dim3(blocks), dim3(256), 0, 0, din, dR, batch, K);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 1) t.push_back(ms);
    }
    CK(hipGetLastError());
    unsigned long long clk[4];
    CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(lab_clk), sizeof(clk)));
    std::sort(t.begin(), t.end());
    printf("%-44s blocks/CU %d : %.3f ms (min %.3f)  %.2f TB/s   cycle counter / 100 MHz wall clock = %.2f\n", name, blocks_per_cu,
           t[t.size() / 2], t[0], (double)batch * K * 32 / (t[t.size() / 2] * 1e-3) / 1e12, (double)clk[0] / (double)clk[1]);
    fflush(stdout);
    if (out) { out->resize((size_t)batch * 16); CK(hipMemcpy(out->data(), dR, out->size() * sizeof(double2), hipMemcpyDeviceToHost)); }
}

double maxdiff(const std::vector<double2>& a, const std::vector<double2>& b)
{
    double m = 0;
    for (size_t i = 0; i < a.size(); ++i) m = std::max(m, std::max(fabs(a[i].x - b[i].x), fabs(a[i].y - b[i].y)));
    return m;
}

// pure stream with the covariance kernel's load structure (8-deep ring of 1-KiB wave loads, nt): a wave reads runs of
// `cpw` consecutive chunks; run r of wave gw starts at chunk (r * nwaves + gw) * cpw.  cpw = 8: cov4_x4_kernel (one item
// per run), cpw = 512: cov4_evd_kernel (64 items per run), cpw = 1: the grid-stride order of scripts/ubench_hbm.hip.
__global__ __launch_bounds__(256) void stream_lab(const v4f32* __restrict__ in, uint32_t per_wave, uint32_t cpw_log2, float* sink)
{
    const int lane = threadIdx.x & 63;
    const uint32_t gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const uint32_t cmask = (1u << cpw_log2) - 1u;
    auto chunk = [&](uint32_t s) -> size_t {
        s = s < per_wave ? s : per_wave - 1;
        return ((size_t)((s >> cpw_log2) * nw + gw) << cpw_log2) + (s & cmask);
    };
    v4f32 pf[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) pf[u] = __builtin_nontemporal_load(in + chunk(u) * 64 + lane);
    v4f32 acc = {0, 0, 0, 0};
    for (uint32_t s = 0; s < per_wave; s += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc += pf[u];
            asm volatile("" ::: "memory");
            pf[u] = __builtin_nontemporal_load(in + chunk(s + u + 8) * 64 + lane);
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) sink[0] = 1.0f;
}

void run_stream(const float* din, float* sink, size_t nchunks, uint32_t blocks, uint32_t cpw_log2)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> t;
    const uint32_t per_wave = (uint32_t)(nchunks / (blocks * 4));
    for (int rep = 0; rep < 32; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(stream_lab, dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const v4f32*>(din), per_wave, cpw_log2, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 1) t.push_back(ms);
    }
    CK(hipGetLastError());
    std::sort(t.begin(), t.end());
    printf("stream: %5u waves, runs of %4u KiB per wave : %.3f ms (min %.3f)  %.2f TB/s\n", blocks * 4, 1u << cpw_log2, t[t.size() / 2], t[0],
           (double)nchunks * 1024 / (t[t.size() / 2] * 1e-3) / 1e12);
    fflush(stdout);
}

int main()
{
    const uint32_t batch = 262144, K = 256;
    const size_t nfl = (size_t)batch * K * 8;
    float* din; double2* dR;
    CK(hipMalloc(&din, nfl * 4)); CK(hipMalloc(&dR, (size_t)batch * 16 * sizeof(double2)));
    {
        std::vector<float> h(1 << 22);
        uint32_t s = 12345;
        for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
        for (size_t o = 0; o < nfl; o += h.size()) CK(hipMemcpy(din + o, h.data(), std::min(h.size(), nfl - o) * 4, hipMemcpyHostToDevice));
    }
    std::vector<double2> r0, r3, r4, r6;
    float* sink; CK(hipMalloc(&sink, 64));
    printf("filled\n"); fflush(stdout);
    for (int i = 0; i < 1200; ++i) hipLaunchKernelGGL(cov_lab<0>, dim3(256), dim3(256), 0, 0, din, dR, batch, K);   // clock ramp
    CK(hipDeviceSynchronize());
    printf("ramped\n"); fflush(stdout);
    for (int rep = 0; rep < 2; ++rep)
        for (uint32_t blocks : {256u, 512u, 1024u})
            for (uint32_t l2 : {3u, 9u}) if ((size_t)batch * 8 / (blocks * 4) >= (1u << l2)) run_stream(din, sink, (size_t)batch * 8, blocks, l2);
    std::vector<double2> r9, r10;
    for (int pc : {2, 1, 2}) {
        if (pc == 1) run<9>("9 shipped loop, precomputed offsets, * 1/K", din, dR, batch, K, pc, &r9); else run<9>("9 shipped loop, precomputed offsets, * 1/K", din, dR, batch, K, pc);
        if (pc == 1) run<10>("10 = 9 + epilogue once per 4 items (64 lanes)", din, dR, batch, K, pc, &r10); else run<10>("10 = 9 + epilogue once per 4 items", din, dR, batch, K, pc);
        run<11>("11 = 10 without the R store", din, dR, batch, K, pc);
        run<100>("10 with buffer store aux 0 (plain)", din, dR, batch, K, pc);
        run<101>("10 with buffer store aux 1 (sc0)", din, dR, batch, K, pc);
        run<102>("10 with buffer store aux 2 (nt)", din, dR, batch, K, pc);
        run<103>("10 with buffer store aux 3 (sc0 nt)", din, dR, batch, K, pc);
        run<116>("10 with buffer store aux 16 (sc1)", din, dR, batch, K, pc);
        run<117>("10 with buffer store aux 17 (sc0 sc1)", din, dR, batch, K, pc);
        run<118>("10 with buffer store aux 18 (sc1 nt)", din, dR, batch, K, pc);
        run<119>("10 with buffer store aux 19 (sc0 sc1 nt)", din, dR, batch, K, pc);
        run<8>("8 no LDS, no MFMA, no epilogue", din, dR, batch, K, pc);
        if (pc == 1) run<0>("0 shipped loop", din, dR, batch, K, pc, &r0); else run<0>("0 shipped loop", din, dR, batch, K, pc);
    }
    printf("max |R - R_shipped|: V9 %.3g V10 %.3g\n", maxdiff(r0, r9), maxdiff(r0, r10));
    //printf("max |R - R_shipped|: fp32 staging %.3g, 8-buffer %.3g, DPP %.3g   (|R| ~ %.3g)\n", maxdiff(r0, r3), maxdiff(r0, r6), maxdiff(r0, r4), fabs(r0[0].x));
    return 0;
}
