// Lab harness (not product): cfg3's scan (m = 8, 16,384 items x 36,000 bins) with REAL operands on the bf16 / f16 matrix
// core, checked against fp64 on the host -- the accuracy side of DESIGN.md 10 item 5 on the hardware's own accumulation
// (tests/lab/f32_bulk_study.py emulates it on the CPU; scripts/ubench_bf16x3_scan.hip is the timing-only form).
//   d(item, bin) = <q(item), t(bin)> over the 64 real terms of a^H Q a (q: diagonal, Re and Im of the upper triangle of the
//   noise projector Q = I - S S^H; t: |a_i|^2, 2 Re conj(a_i) a_j, -2 Im conj(a_i) a_j of the steering vector), spectrum = 1 / d.
//   PARTS = 3: both operands split into three bf16 parts, cross products (l,h) (h,l) (m,m) (m,h) (h,m) (h,h), K = 384
//   PARTS = 2: two f16 parts of q 2^10 and t 2^12, cross products (l,l) (l,h) (h,l) (h,h), K = 256
//   f32 accumulation inside v_mfma_f32_32x32x16_{bf16,f16}, small cross products first.
// Scenes: every item's signal subspace is spanned by the steering vectors of two random directions plus a perturbation
// of random size (10 .. 60 dB down), so the spectrum has the two nulls of a MUSIC scene at random places and depths.
// The first 64 items' spectra are compared with fp64 values computed on the host, by class of d / ||a||^2.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/scan_parts_lab scripts/scan_parts_lab.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef short v8s __attribute__((ext_vector_type(8)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

constexpr uint32_t M = 8, MM = 64, ITEMS = 16384, RES = 36000, TILE = 32, NTILES = RES / TILE, RANGES = 16, WG_ITEMS = 128;

template <int PARTS>
__global__ __launch_bounds__(256) void scan_parts(const v4u* __restrict__ A, const v4u* __restrict__ B, float* __restrict__ spec, float out_scale)
{
    constexpr int NSTEP = (PARTS == 3 ? 6 : 4) * 4;
    extern __shared__ v4u lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t range = blockIdx.x % RANGES, iblk = blockIdx.x / RANGES;
    const uint32_t t0 = (uint32_t)(((uint64_t)NTILES * range) / RANGES), t1 = (uint32_t)(((uint64_t)NTILES * (range + 1)) / RANGES);
    const uint32_t item0 = iblk * WG_ITEMS + wave * 32;
    v4u a[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) a[s] = A[((size_t)(item0 / 32) * NSTEP + s) * 64 + lane];
    constexpr int UNITS = NSTEP * 64;
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
            const v4u b = lds[buf * UNITS + s * 64 + lane];
            if constexpr (PARTS == 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8s, a[s]), __builtin_bit_cast(v8s, b), acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a[s]), __builtin_bit_cast(v8h, b), acc, 0, 0, 0);
        }
        const uint32_t bin = t * TILE + (lane & 31);                  // C/D: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t row = item0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            __builtin_nontemporal_store(out_scale * __builtin_amdgcn_rcpf(acc[r]), spec + (size_t)row * RES + bin);
        }
        __syncthreads();
    }
}

static uint16_t bf16_of(float x) { uint32_t u; memcpy(&u, &x, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf16_to(uint16_t h) { uint32_t u = (uint32_t)h << 16; float x; memcpy(&x, &u, 4); return x; }
static uint16_t f16_of(float x) { _Float16 h = (_Float16)x; uint16_t u; memcpy(&u, &h, 2); return u; }
static float f16_to(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

// parts[p][e] of one 64-term vector; p = 0 is the leading part
template <int PARTS>
static void split(const double* x, double scale, uint16_t (*parts)[MM])
{
    for (uint32_t e = 0; e < MM; ++e) {
        float r = (float)(x[e] * scale);
        for (int p = 0; p < PARTS; ++p) {
            const uint16_t h = PARTS == 3 ? bf16_of(r) : f16_of(r);
            parts[p][e] = h;
            r = r - (PARTS == 3 ? bf16_to(h) : f16_to(h));             // exact in float
        }
    }
}

// operand images: [block of 32 vectors][step][lane] x 8 halfwords; step s = cross product s / 4, terms 16 (s % 4) + 8 (lane >> 5) + j
template <int PARTS>
static void pack(const std::vector<double>& img, uint32_t nvec, double scale, bool is_a, std::vector<uint16_t>& out)
{
    constexpr int NC = PARTS == 3 ? 6 : 4, NSTEP = NC * 4;
    // part index per cross product (0 = h, 1 = m or l, 2 = l), small products first
    static const int ca3[6] = {2, 0, 1, 1, 0, 0}, cb3[6] = {0, 2, 1, 0, 1, 0};
    static const int ca2[4] = {1, 1, 0, 0}, cb2[4] = {1, 0, 1, 0};
    out.assign((size_t)(nvec / 32) * NSTEP * 64 * 8, 0);
    std::vector<uint16_t> parts((size_t)PARTS * MM);
    for (uint32_t v = 0; v < nvec; ++v) {
        split<PARTS>(&img[(size_t)v * MM], scale, reinterpret_cast<uint16_t(*)[MM]>(parts.data()));
        const uint32_t blk = v / 32, col = v % 32;
        for (int s = 0; s < NSTEP; ++s) {
            const int c = s / 4, p = PARTS == 3 ? (is_a ? ca3[c] : cb3[c]) : (is_a ? ca2[c] : cb2[c]);
            for (int half = 0; half < 2; ++half)
                for (int j = 0; j < 8; ++j)
                    out[(((size_t)blk * NSTEP + s) * 64 + (col + 32 * half)) * 8 + j] = parts[(size_t)p * MM + 16 * (s % 4) + 8 * half + j];
        }
    }
}

static double urand(uint64_t& st) { st = st * 6364136223846793005ull + 1442695040888963407ull; return (double)(st >> 11) / 9007199254740992.0; }

// the packed images evaluated on the host the way the matrix core is meant to (no GPU needed: scan_parts_lab --host-check)
template <int PARTS>
static void host_check(const std::vector<double>& q, const std::vector<double>& timg)
{
    constexpr int NSTEP = (PARTS == 3 ? 6 : 4) * 4;
    const double sq = PARTS == 3 ? 1.0 : 1024.0, st = PARTS == 3 ? 1.0 : 4096.0;
    std::vector<double> q64(q.begin(), q.begin() + 64 * MM), t64(timg.begin() + (size_t)7040 * MM, timg.begin() + (size_t)7104 * MM);
    std::vector<uint16_t> hA, hB;
    pack<PARTS>(q64, 64, sq, true, hA);
    pack<PARTS>(t64, 64, st, false, hB);
    double worst = 0;
    for (uint32_t i = 0; i < 64; ++i)
        for (uint32_t b = 0; b < 64; ++b) {
            double acc = 0;
            for (int s = 0; s < NSTEP; ++s)
                for (int half = 0; half < 2; ++half)
                    for (int j = 0; j < 8; ++j) {
                        const uint16_t ua = hA[(((size_t)(i / 32) * NSTEP + s) * 64 + (i % 32 + 32 * half)) * 8 + j];
                        const uint16_t ub = hB[(((size_t)(b / 32) * NSTEP + s) * 64 + (b % 32 + 32 * half)) * 8 + j];
                        acc += (double)(PARTS == 3 ? bf16_to(ua) : f16_to(ua)) * (double)(PARTS == 3 ? bf16_to(ub) : f16_to(ub));
                    }
            double d = 0; for (uint32_t e = 0; e < MM; ++e) d += q64[(size_t)i * MM + e] * t64[(size_t)b * MM + e];
            worst = std::max(worst, std::fabs(acc / (sq * st) - d));
        }
    printf("# host check, %d parts: packed images reproduce <q, t> to %.2e absolute over 64 x 64 pairs\n", PARTS, worst);
}

template <int PARTS>
static void run(const std::vector<double>& q, const std::vector<double>& timg, float* d_spec)
{
    constexpr int NSTEP = (PARTS == 3 ? 6 : 4) * 4;
    const double sq = PARTS == 3 ? 1.0 : 1024.0, st = PARTS == 3 ? 1.0 : 4096.0;
    std::vector<uint16_t> hA, hB;
    pack<PARTS>(q, ITEMS, sq, true, hA);
    pack<PARTS>(timg, RES, st, false, hB);
    v4u *dA, *dB;
    CK(hipMalloc((void**)&dA, hA.size() * 2)); CK(hipMalloc((void**)&dB, hB.size() * 2));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    const dim3 grid((ITEMS / WG_ITEMS) * RANGES), block(256);
    const size_t lds = (size_t)2 * NSTEP * 64 * 16;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> tm;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipDeviceSynchronize()); CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((scan_parts<PARTS>), grid, block, lds, 0, dA, dB, d_spec, (float)(sq * st));
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep > 1) tm.push_back(ms);
    }
    CK(hipGetLastError());
    std::sort(tm.begin(), tm.end());
    const uint32_t NCHK = 64;
    std::vector<float> hs((size_t)NCHK * RES);
    CK(hipMemcpy(hs.data(), d_spec, hs.size() * 4, hipMemcpyDeviceToHost));
    const double thr[5] = {0.5, 0.125, 0.05, 0.02, 0.0};
    double worst[5] = {0, 0, 0, 0, 0}; uint64_t cnt[5] = {0, 0, 0, 0, 0}; double sum_rel = 0; uint64_t bad = 0;
    for (uint32_t i = 0; i < NCHK; ++i)
        for (uint32_t b = 0; b < RES; ++b) {
            long double d = 0;
            for (uint32_t e = 0; e < MM; ++e) d += (long double)q[(size_t)i * MM + e] * (long double)timg[(size_t)b * MM + e];
            const double ref = 1.0 / (double)d, got = hs[(size_t)i * RES + b];
            const double rel = std::fabs(got - ref) / std::fabs(ref), frac = (double)d / M;
            if (!(rel == rel)) { ++bad; continue; }
            sum_rel += rel;
            for (int k = 0; k < 5; ++k) if (frac >= thr[k]) { ++cnt[k]; worst[k] = std::max(worst[k], rel); }
        }
    printf("%s: %.3f ms per %u x %u values (min %.3f) | %u items checked against fp64, mean rel %.1e, NaN %llu\n",
           PARTS == 3 ? "three bf16 parts, K = 384" : "two f16 parts,   K = 256", tm[tm.size() / 2], ITEMS, RES, tm[0], NCHK,
           sum_rel / ((double)NCHK * RES), (unsigned long long)bad);
    for (int k = 0; k < 5; ++k)
        printf("    d/||a||^2 >= %-5.3f: %5.1f %% of the values, worst rel err %.2e\n", thr[k], 100.0 * cnt[k] / ((double)NCHK * RES), worst[k]);
    fflush(stdout);
    CK(hipFree(dA)); CK(hipFree(dB));
}

int main()
{
    const double PI = 3.14159265358979323846;
    // 8-element circle, adjacent spacing half a wavelength: radius = 0.25 / sin(pi / 8) wavelengths
    const double rad = 0.25 / std::sin(PI / M);
    auto steer = [&](double th, std::complex<double>* a) {
        for (uint32_t k = 0; k < M; ++k) {
            const double px = rad * std::cos(2 * PI * k / M), py = rad * std::sin(2 * PI * k / M);
            const double ph = -2 * PI * (px * std::cos(th) + py * std::sin(th));
            a[k] = std::complex<double>((float)std::cos(ph), (float)std::sin(ph));      // the table is complex64
        }
    };
    auto image_t = [&](const std::complex<double>* a, double* t) {
        uint32_t o = 0;
        for (uint32_t i = 0; i < M; ++i) t[o++] = std::norm(a[i]);
        for (uint32_t i = 0; i < M; ++i) for (uint32_t j = i + 1; j < M; ++j) t[o++] = 2.0 * (std::conj(a[i]) * a[j]).real();
        for (uint32_t i = 0; i < M; ++i) for (uint32_t j = i + 1; j < M; ++j) t[o++] = -2.0 * (std::conj(a[i]) * a[j]).imag();
    };
    std::vector<double> timg((size_t)RES * MM), q((size_t)ITEMS * MM);
    std::complex<double> a[M];
    for (uint32_t b = 0; b < RES; ++b) { steer(b * 2 * PI / RES, a); image_t(a, &timg[(size_t)b * MM]); }
    uint64_t st = 20260922;
    for (uint32_t it = 0; it < ITEMS; ++it) {
        std::complex<double> S[2][M];
        const double eps = std::pow(10.0, -(0.5 + 2.5 * urand(st)));                  // perturbation 10 .. 60 dB down
        for (int c = 0; c < 2; ++c) {
            steer(2 * PI * urand(st), S[c]);
            for (uint32_t k = 0; k < M; ++k) S[c][k] += eps * std::complex<double>(urand(st) - 0.5, urand(st) - 0.5);
        }
        for (int c = 0; c < 2; ++c) {                                                // Gram-Schmidt, twice
            for (int pass = 0; pass < 2; ++pass)
                for (int p = 0; p < c; ++p) {
                    std::complex<double> h = 0; for (uint32_t k = 0; k < M; ++k) h += std::conj(S[p][k]) * S[c][k];
                    for (uint32_t k = 0; k < M; ++k) S[c][k] -= h * S[p][k];
                }
            double n2 = 0; for (uint32_t k = 0; k < M; ++k) n2 += std::norm(S[c][k]);
            for (uint32_t k = 0; k < M; ++k) S[c][k] /= std::sqrt(n2);
        }
        // d = a^H Q a = sum_i Q_ii |a_i|^2 + sum_{i<j} (Re Q_ij * 2 Re conj(a_i) a_j + Im Q_ij * (-2 Im conj(a_i) a_j))
        double* qi = &q[(size_t)it * MM];
        std::complex<double> Q[M][M];
        for (uint32_t i = 0; i < M; ++i) for (uint32_t j = 0; j < M; ++j)
            Q[i][j] = (i == j ? 1.0 : 0.0) - (S[0][i] * std::conj(S[0][j]) + S[1][i] * std::conj(S[1][j]));
        uint32_t o = 0;
        for (uint32_t i = 0; i < M; ++i) qi[o++] = Q[i][i].real();
        for (uint32_t i = 0; i < M; ++i) for (uint32_t j = i + 1; j < M; ++j) qi[o++] = Q[i][j].real();
        for (uint32_t i = 0; i < M; ++i) for (uint32_t j = i + 1; j < M; ++j) qi[o++] = Q[i][j].imag();
    }
    // self-check of the real image on the host: <q, t> must equal a^H Q a (item 0, bin 777)
    {
        steer(777 * 2 * PI / RES, a);
        std::complex<double> S0[M]; (void)S0;
        double d = 0; for (uint32_t e = 0; e < MM; ++e) d += q[e] * timg[(size_t)777 * MM + e];
        printf("# host: <q, t>(item 0, bin 777) = %.12f (must lie in [0, 8])\n", d);
    }
    host_check<3>(q, timg); host_check<2>(q, timg);
    {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { printf("# no GPU: host check only\n"); return 0; }
    }
    float* spec; CK(hipMalloc((void**)&spec, (size_t)ITEMS * RES * 4)); CK(hipMemset(spec, 0, (size_t)ITEMS * RES * 4));
    printf("# cfg3's scan with split operands on the bf16 / f16 matrix core, real operands: %u items x %u bins, m = %u\n", ITEMS, RES, M);
    run<3>(q, timg, spec);
    run<2>(q, timg, spec);
    return 0;
}
