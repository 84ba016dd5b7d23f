// Lab harness (not product; run twice in round 3: profiles/r03_frontend_fused_lab.txt -- bit-identical, and NOT faster):
// the config-5 front-end with the resampler folded INTO the AGC's two tile passes.
//
// Today: resamp_kernel writes the resampled streams (read 8 B / 1.25 + write 8 B per output sample), then
// the AGC reads them twice (tile maps, apply) and writes the MUSIC items: 0.28 + 0.36 ms of the 1.17-ms step, 2.6 GB of
// HBM traffic for 0.54 GB of items.  The resampled stream is a pure function of the raw input (output o = 8 taps over
// in[ii_o ..], ii_o and the tap row from the 64.64 phase P_o), so both AGC passes can evaluate it on the fly from the
// raw input: the intermediate stream is never written or read (raw input read twice, items written once: 1.4 GB).
// Bit-exactness is by construction -- resamp_sample() below is resamp_kernel's arithmetic, the rest is
// agc_tile_kernel<0> / <1> -- and the harness checks it: the fused path's items against resamp_kernel ->
// agc_tile_kernel<0> -> agc_carry_kernel -> agc_tile_kernel<1>, bit for bit, then times both.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/frontend_fused_lab scripts/frontend_fused_lab.hip
//   scripts/frontend_fused_lab [items=16384] [iterations=20]
#include "../gr_baz_amd/csrc/agc_kernels.hip.h"
#include "../gr_baz_amd/csrc/resamp_kernels.hip.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace bazagc;
using namespace bazresamp;

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e__), __LINE__); exit(1); } } while (0)

// output sample o of one stream: resamp_kernel's body (resamp_kernels.hip.h), operation for operation
__device__ __forceinline__ float2 resamp_sample(const float2* __restrict__ xs, const PhaseParams& p, uint32_t o,
                                                const float* __restrict__ st)
{
    uint64_t ii, frac;
    phase_of(p, o, ii, frac);
    const float mu = __ull2float_rn(frac) * 5.42101086242752217e-20f;
    const int imu = __float2int_rn(mu * (float)RS_NSTEPS);
    const float* t = st + imu * RS_NTAPS;
    const float2* x = xs + ii;
    float re = 0.0f, im = 0.0f;
    {
#pragma clang fp contract(off)
#pragma unroll
        for (int k = 0; k < RS_NTAPS; ++k) {
            const float2 v = x[k];
            const float w = t[RS_NTAPS - 1 - k];
            re = re + v.x * w;
            im = im + v.y * w;
        }
    }
    return make_float2(re, im);
}

// the same arithmetic with the 8 input samples taken from an LDS copy of the tile's input window (win[0] = sample w0)
__device__ __forceinline__ float2 resamp_sample_win(const float2* __restrict__ win, uint64_t w0, const PhaseParams& p, uint32_t o,
                                                    const float* __restrict__ st)
{
    uint64_t ii, frac;
    phase_of(p, o, ii, frac);
    const float mu = __ull2float_rn(frac) * 5.42101086242752217e-20f;
    const int imu = __float2int_rn(mu * (float)RS_NSTEPS);
    const float* t = st + imu * RS_NTAPS;
    const float2* x = win + (uint32_t)(ii - w0);
    float re = 0.0f, im = 0.0f;
    {
#pragma clang fp contract(off)
#pragma unroll
        for (int k = 0; k < RS_NTAPS; ++k) {
            const float2 v = x[k];
            const float w = t[RS_NTAPS - 1 - k];
            re = re + v.x * w;
            im = im + v.y * w;
        }
    }
    return make_float2(re, im);
}

constexpr int FW_MAX = 448;       // samples of input window per stream and tile the LDS form holds (ratio <= ~1.7)

// state "before sample 0" of every stream when the call starts the stream (agc_carry_kernel's first != 0 rule), from
// the resampled sample 0: lets agc_carry_kernel run unchanged with first = 0
__global__ void fused_first_kernel(const float2* __restrict__ raw, uint64_t raw_stride, PhaseParams p,
                                   const float* __restrict__ taps, double* __restrict__ env_state, uint32_t nstreams)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < nstreams) env_state[s] = agc_mag(resamp_sample(raw + (size_t)s * raw_stride, p, 0u, taps));
}

// agc_tile_kernel<MODE> (MODE 0: tile maps, MODE 1: apply + interleave; workgroup = the S streams of one tile) with its
// input samples computed by the resampler instead of loaded
// STAGED = false: lane l evaluates its own AGC_IE consecutive outputs (a wave-wide load then strides 5 input samples per
//                 lane: ~3x the L1 line requests of resamp_kernel);
// STAGED = true:  the wave evaluates its 256 outputs with lane = output mod 64 (resamp_kernel's coalesced pattern), parks
//                 them in a wave-private 2-KiB LDS row and reads back AGC_IE consecutive ones per lane.
// STAGE = 2: the wave first copies its tile's INPUT window (<= FW_MAX samples, 16-B loads, every input byte once) into LDS and
//            every lane then evaluates its own AGC_IE consecutive outputs from there: no 8-byte gathers through the L1
//            (resamp_kernel issues 8 of them per output: 64 addresses per instruction, the address path sets its pace).
template <int MODE, int STAGE>
__global__ __launch_bounds__(1024) void fused_tile_kernel(const float2* __restrict__ raw, uint64_t raw_stride, PhaseParams p,
                                                           const float* __restrict__ taps, uint64_t n, AgcParams P,
                                                           double2* __restrict__ chunk_pair, const double* __restrict__ carry_in,
                                                           uint32_t ntiles, float2* __restrict__ out, double* __restrict__ env_state,
                                                           uint32_t nstreams)
{
    extern __shared__ float2 tile[];                         // MODE 1: [nstreams][AGC_IT + 1], then STAGED: [nstreams][AGC_IT]
    __shared__ float st[(RS_NSTEPS + 1) * RS_NTAPS];
    for (int i = threadIdx.x; i < (RS_NSTEPS + 1) * RS_NTAPS; i += blockDim.x) st[i] = taps[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const uint32_t stream = threadIdx.x >> 6, t = blockIdx.x;
    const uint64_t base = (uint64_t)t * AGC_IT;
    const uint32_t valid = (uint32_t)((n - base < (uint64_t)AGC_IT) ? (n - base) : AGC_IT);
    const int i0 = lane * AGC_IE;
    const int cnt = ((int)valid - i0) < 0 ? 0 : (((int)valid - i0) > AGC_IE ? AGC_IE : ((int)valid - i0));
    const float2* __restrict__ xs = raw + (size_t)stream * raw_stride;
    float2 x[AGC_IE];
    double mag[AGC_IE];
    if constexpr (STAGE == 2) {
        float2* __restrict__ win = tile + (MODE == 1 ? (size_t)nstreams * (AGC_IT + 1) : 0) + (size_t)stream * FW_MAX;
        uint64_t ii0, ii1, fr;
        phase_of(p, (uint32_t)base, ii0, fr);
        phase_of(p, (uint32_t)(base + valid - 1), ii1, fr);
        const uint64_t w0 = ii0 & ~1ull;                                    // 16-B aligned start
        const uint32_t nq = (uint32_t)((ii1 + RS_NTAPS - w0 + 1) >> 1);     // float4s to copy (<= FW_MAX / 2)
        const float4* __restrict__ src4 = reinterpret_cast<const float4*>(xs + w0);
        float4* __restrict__ dst4 = reinterpret_cast<float4*>(win);
#pragma unroll
        for (int u = 0; u < (FW_MAX / 2 + 63) / 64; ++u) {
            const uint32_t i4 = (uint32_t)u * 64u + (uint32_t)lane;
            if (i4 < nq) dst4[i4] = src4[i4];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < AGC_IE; ++j)
            x[j] = (j < cnt) ? resamp_sample_win(win, w0, p, (uint32_t)(base + i0 + j), st) : make_float2(0.f, 0.f);
    } else if constexpr (STAGE == 1) {
        float2* __restrict__ park = tile + (MODE == 1 ? (size_t)nstreams * (AGC_IT + 1) : 0) + (size_t)stream * AGC_IT;
#pragma unroll
        for (int j = 0; j < AGC_IE; ++j) {
            const uint32_t q = (uint32_t)j * 64u + (uint32_t)lane;
            if (q < valid) park[q] = resamp_sample(xs, p, (uint32_t)(base + q), st);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int j = 0; j < AGC_IE; ++j) x[j] = (j < cnt) ? park[i0 + j] : make_float2(0.f, 0.f);
    } else {
#pragma unroll
        for (int j = 0; j < AGC_IE; ++j)
            x[j] = (j < cnt) ? resamp_sample(xs, p, (uint32_t)(base + i0 + j), st) : make_float2(0.f, 0.f);
    }
    double A = 1.0, S = 0.0;
#pragma unroll
    for (int j = 0; j < AGC_IE; ++j) {
        mag[j] = agc_mag(x[j]);
        if (j < cnt) { S = fma(P.a, S, P.b * mag[j]); A *= P.a; }
    }
    double Ai = A, Si = S;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double Ap = __shfl_up(Ai, d, 64), Sp = __shfl_up(Si, d, 64);
        if (lane >= d) compose(Ai, Si, Ap, Sp);
    }
    if (MODE == 0) {
        if (lane == 63) chunk_pair[(size_t)stream * ntiles + t] = make_double2(Ai, Si);
        return;
    }
    double Ae = __shfl_up(Ai, 1, 64), Se = __shfl_up(Si, 1, 64);
    if (lane == 0) { Ae = 1.0; Se = 0.0; }
    double e = fma(Ae, carry_in[(size_t)stream * ntiles + t], Se);
    float2* __restrict__ row = tile + (size_t)stream * (AGC_IT + 1);
#pragma unroll
    for (int j = 0; j < AGC_IE; ++j) {
        if (j < cnt) {
            e = agc_env_step(e, mag[j], P.a, P.b);
            const double gain = __ddiv_rn(P.reference, e);
            row[i0 + j] = agc_apply(x[j], gain);
        }
    }
    if (env_state && base + i0 + cnt == n && cnt > 0) env_state[stream] = e;
    __syncthreads();
    float2* __restrict__ ob = out + (size_t)base * nstreams;
    const uint32_t total = valid * nstreams;
    if ((nstreams & 1u) == 0 && (reinterpret_cast<uintptr_t>(ob) & 15u) == 0) {
        for (uint32_t q = threadIdx.x * 2; q < total; q += blockDim.x * 2) {
            const uint32_t tt = q / nstreams, ss = q - tt * nstreams;
            const float2 a = tile[(size_t)ss * (AGC_IT + 1) + tt], b = tile[(size_t)(ss + 1) * (AGC_IT + 1) + tt];
            *reinterpret_cast<float4*>(ob + q) = make_float4(a.x, a.y, b.x, b.y);
        }
    } else {
        for (uint32_t q = threadIdx.x; q < total; q += blockDim.x) {
            const uint32_t tt = q / nstreams, ss = q - tt * nstreams;
            ob[q] = tile[(size_t)ss * (AGC_IT + 1) + tt];
        }
    }
}

__global__ void fill_kernel(float2* __restrict__ x, size_t n, uint32_t seed)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ (uint32_t)(i >> 32) * 40503u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        const uint32_t g = h * 1103515245u + 12345u;
        x[i] = make_float2((float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f, (float)(int)(g >> 8) * (1.0f / 8388608.0f) - 1.0f);
    }
}

__global__ void diff_kernel(const uint2* __restrict__ a, const uint2* __restrict__ b, size_t n, unsigned long long* __restrict__ ndiff)
{
    unsigned long long local = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        local += (a[i].x != b[i].x) + (a[i].y != b[i].y);
    if (local) atomicAdd(ndiff, local);
}

int main(int argc, char** argv)
{
    const uint32_t items = argc > 1 ? (uint32_t)atoi(argv[1]) : 16384u;
    const int iters = argc > 2 ? atoi(argv[2]) : 20;
    const uint32_t S = 16, K = 256;                              // config 5: 16 antennas, 4096 samples per item
    const uint64_t n = (uint64_t)items * K;                      // output samples per antenna
    const double ratio = 1.25;                                   // input samples per output sample
    const uint64_t raw_n = (uint64_t)((double)n * ratio) + 16;   // per antenna, with the 8-tap tail
    const uint32_t ntiles = (uint32_t)((n + AGC_IT - 1) / AGC_IT);

    PhaseParams p;                                               // mu_0 = 0, inc = 1.25 in 64.64
    p.first_lo = 0; p.first_hi = 0;
    p.inc_hi = 1; p.inc_lo = 1ull << 62;
    p.base_hi = 1; p.base_lo = 1ull << 62;                       // P_1 = P_0 + inc
    AgcParams P;
    P.b = (double)1e-4f; P.a = 1.0 - P.b; P.reference = 1.0;

    std::vector<float> taps((RS_NSTEPS + 1) * RS_NTAPS);
    for (int s = 0; s <= RS_NSTEPS; ++s)                         // any table will do for a bit-for-bit comparison: a
        for (int k = 0; k < RS_NTAPS; ++k) {                     // windowed-sinc-like shape with irregular low bits
            const double xk = (double)(k - 3) - (double)s / RS_NSTEPS;
            const double w = 0.54 + 0.46 * cos(3.14159265358979 * xk / 4.5);
            taps[s * RS_NTAPS + (RS_NTAPS - 1 - k)] = (float)((fabs(xk) < 1e-12 ? 1.0 : sin(3.14159265358979 * xk) / (3.14159265358979 * xk)) * w);
        }

    float2 *raw, *mid, *items_ref, *items_fused;
    float* d_taps;
    double2* pair;
    double *carry, *env;
    unsigned long long* ndiff;
    CK(hipMalloc(&raw, (size_t)S * raw_n * 8));
    CK(hipMalloc(&mid, (size_t)S * n * 8));
    CK(hipMalloc(&items_ref, (size_t)S * n * 8));
    CK(hipMalloc(&items_fused, (size_t)S * n * 8));
    CK(hipMalloc(&d_taps, taps.size() * 4));
    CK(hipMalloc(&pair, (size_t)S * ntiles * 16));
    CK(hipMalloc(&carry, (size_t)S * ntiles * 8));
    CK(hipMalloc(&env, S * 8));
    CK(hipMalloc(&ndiff, 8));
    CK(hipMemcpy(d_taps, taps.data(), taps.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, raw, (size_t)S * raw_n, 12345u);
    CK(hipMemset(items_ref, 0, (size_t)S * n * 8));
    CK(hipMemset(items_fused, 0xFF, (size_t)S * n * 8));
    CK(hipDeviceSynchronize());

    const dim3 tile_block(64 * S);
    const size_t lds = (size_t)S * (AGC_IT + 1) * sizeof(float2);
    const dim3 rs_grid((uint32_t)((n + RS_BLOCK * RS_PER_THREAD - 1) / (RS_BLOCK * RS_PER_THREAD)), S);
    auto three_engines = [&]() {
        hipLaunchKernelGGL(resamp_kernel, rs_grid, dim3(RS_BLOCK), 0, 0, raw, raw_n, mid, n, (uint32_t)n, p, d_taps);
        hipLaunchKernelGGL((agc_tile_kernel<0>), dim3(ntiles), tile_block, 0, 0, mid, n, n, P, pair, (const double*)nullptr, ntiles,
                           (float2*)nullptr, (double*)nullptr, S, (float*)nullptr, (float*)nullptr);
        hipLaunchKernelGGL(agc_carry_kernel, dim3(S), dim3(AGC_CARRY_THREADS), 0, 0, mid, n, pair, carry, ntiles, env, 1);
        hipLaunchKernelGGL((agc_tile_kernel<1>), dim3(ntiles), tile_block, lds, 0, mid, n, n, P, pair, carry, ntiles, items_ref, env, S,
                           (float*)nullptr, (float*)nullptr);
    };
    const size_t park = (size_t)S * AGC_IT * sizeof(float2);
    const size_t winb = (size_t)S * FW_MAX * sizeof(float2);
    auto fused_win = [&]() {
        hipLaunchKernelGGL((fused_tile_kernel<0, 2>), dim3(ntiles), tile_block, winb, 0, raw, raw_n, p, d_taps, n, P, pair,
                           (const double*)nullptr, ntiles, (float2*)nullptr, (double*)nullptr, S);
        hipLaunchKernelGGL(fused_first_kernel, dim3(1), dim3(64), 0, 0, raw, raw_n, p, d_taps, env, S);
        hipLaunchKernelGGL(agc_carry_kernel, dim3(S), dim3(AGC_CARRY_THREADS), 0, 0, raw, raw_n, pair, carry, ntiles, env, 0);
        hipLaunchKernelGGL((fused_tile_kernel<1, 2>), dim3(ntiles), tile_block, lds + winb, 0, raw, raw_n, p, d_taps, n, P, pair, carry,
                           ntiles, items_fused, env, S);
    };
    auto fused = [&](bool staged) {
        if (staged)
            hipLaunchKernelGGL((fused_tile_kernel<0, 1>), dim3(ntiles), tile_block, park, 0, raw, raw_n, p, d_taps, n, P, pair,
                               (const double*)nullptr, ntiles, (float2*)nullptr, (double*)nullptr, S);
        else
            hipLaunchKernelGGL((fused_tile_kernel<0, 0>), dim3(ntiles), tile_block, 0, 0, raw, raw_n, p, d_taps, n, P, pair,
                               (const double*)nullptr, ntiles, (float2*)nullptr, (double*)nullptr, S);
        hipLaunchKernelGGL(fused_first_kernel, dim3(1), dim3(64), 0, 0, raw, raw_n, p, d_taps, env, S);
        hipLaunchKernelGGL(agc_carry_kernel, dim3(S), dim3(AGC_CARRY_THREADS), 0, 0, raw, raw_n, pair, carry, ntiles, env, 0);
        if (staged)
            hipLaunchKernelGGL((fused_tile_kernel<1, 1>), dim3(ntiles), tile_block, lds + park, 0, raw, raw_n, p, d_taps, n, P, pair, carry,
                               ntiles, items_fused, env, S);
        else
            hipLaunchKernelGGL((fused_tile_kernel<1, 0>), dim3(ntiles), tile_block, lds, 0, raw, raw_n, p, d_taps, n, P, pair, carry, ntiles,
                               items_fused, env, S);
    };

    three_engines();
    CK(hipGetLastError());
    unsigned long long h_ndiff = 0;
    printf("config-5 front-end, %u items (16 antennas x %llu output samples, ratio 1.25)\n", items, (unsigned long long)n);
    {
        const void* fn0 = reinterpret_cast<const void*>(fused_tile_kernel<0, 2>);
        const void* fn1 = reinterpret_cast<const void*>(fused_tile_kernel<1, 2>);
        CK(hipFuncSetAttribute(fn0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)winb));
        CK(hipFuncSetAttribute(fn1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + winb)));
        const void* fn2 = reinterpret_cast<const void*>(fused_tile_kernel<1, 1>);
        CK(hipFuncSetAttribute(fn2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds + park)));
    }
    for (int staged = 0; staged < 3; ++staged) {
        CK(hipMemset(items_fused, 0xFF, (size_t)S * n * 8));
        if (staged == 2) fused_win(); else fused(staged != 0);
        CK(hipGetLastError());
        CK(hipMemset(ndiff, 0, 8));
        hipLaunchKernelGGL(diff_kernel, dim3(4096), dim3(256), 0, 0, (const uint2*)items_ref, (const uint2*)items_fused, (size_t)S * n, ndiff);
        unsigned long long d = ~0ull;
        CK(hipMemcpy(&d, ndiff, 8, hipMemcpyDeviceToHost));
        printf("  fused (%s) vs three engines: %llu of %llu floats differ\n", staged == 2 ? "input window in LDS" : (staged ? "staged through LDS" : "lane-local"), d,
               (unsigned long long)(2 * S * n));
        h_ndiff += d;
    }

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int which = 0; which < 4; ++which) {
        auto run = [&]() { if (which == 3) fused_win(); else if (which) fused(which == 2); else three_engines(); };
        for (int w = 0; w < 5; ++w) run();
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) run();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double per = ms / iters;
        const double bytes = which ? (2.0 * S * raw_n * 8 + (double)S * n * 8) : ((double)S * raw_n * 8 + 4.0 * S * n * 8);
        printf("  %-44s %.3f ms per step, %.0f GB/s of the %.2f GB it must move\n",
               which == 3 ? "fused, input window in LDS:" : (which == 2 ? "fused, staged through LDS:" : (which ? "fused, lane-local:" : "resamp_kernel + agc_tile<0> + carry + agc_tile<1>:")), per,
               bytes / per / 1e6, bytes / 1e9);
    }
    return h_ndiff == 0 ? 0 : 1;
}
