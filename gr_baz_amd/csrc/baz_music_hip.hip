// baz_music_hip.hip -- C-ABI (include/baz_music_hip.h) over the gfx950 MUSIC-DoA kernels.
//
// Host-side counterpart of baz_music_doa's state and work() body
// (/root/reference/lib/baz_music_doa.cc:35-53, 60-70, 72-161).  No torch / GNU Radio types
// cross this boundary; no CPU arithmetic fallback exists: without a usable gfx950 device
// baz_music_create() fails with BAZ_MUSIC_E_NODEVICE / BAZ_MUSIC_E_HIP.
#include "../../include/baz_music_hip.h"
#include "music_kernels.hip.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

using namespace bazmusic;

namespace {


struct StageProf {
    std::vector<hipEvent_t> ev;   // pairs (start, stop)
    size_t used = 0;
    double total_ms = 0.0;
    uint64_t launches = 0;
};

}  // namespace

struct baz_music_ctx {
    uint32_t m = 0, n = 0, nsamples = 0, res = 0, K = 0;
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // steering table as the real bilinear-form table F[bin][m*m] (fp64) in MFMA A-operand order:
    // FA[tile][lane][ks] (see build_F / build_FA), NaN padded
    double* dFA = nullptr;
    uint32_t fa_tiles = 0;   // number of real 16-bin tiles (one extra NaN tile is stored after them)
    // per-bin-range top-n candidates (scan_mfma_kernel -> topn_merge_kernel)
    double* dCandD = nullptr;
    uint32_t* dCandB = nullptr;
    size_t cand_cap = 0;     // entries
    // workspace
    double2* dR = nullptr;
    double* dQ = nullptr;
    uint32_t cap = 0;   // items
    // host-path staging (device side)
    float* s_in = nullptr;
    float* s_ang = nullptr;
    float* s_lvl = nullptr;
    float* s_spec = nullptr;
    uint32_t s_cap = 0;
    bool s_has_spec = false;
    std::mutex mtx;   // serialises set_table against process*, like d_mutex (.cc:67,101)
    bool profiling = false;
    StageProf prof[BAZ_MUSIC_NUM_STAGES];
    std::string stage_name[BAZ_MUSIC_NUM_STAGES];
    char hip_err[256] = {0};
};

namespace {

int hip_fail(baz_music_ctx* c, hipError_t e, const char* what)
{
    if (c) snprintf(c->hip_err, sizeof(c->hip_err), "%s: %s", what, hipGetErrorString(e));
    return BAZ_MUSIC_E_HIP;
}

#define HIP_TRY(ctx, call)                                         \
    do {                                                           \
        hipError_t e__ = (call);                                   \
        if (e__ != hipSuccess) return hip_fail((ctx), e__, #call); \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) changed = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard()
    {
        if (changed) (void)hipSetDevice(prev);
    }
};

// F[bin][i*m+j]: i==j |a_i|^2 ; i<j Re(conj(a_i) a_j) ; i>j Im(conj(a_j) a_i)  (pair (j,i), j<i).
// The fp32 table entries are widened exactly (.cc:110-112); the products are fp64.
void build_F(const float* table_ri, uint32_t m, uint32_t res, std::vector<double>& F)
{
    const size_t mm = (size_t)m * m;
    F.assign((size_t)res * mm, 0.0);
    for (uint32_t s = 0; s < res; ++s) {
        const float* a = table_ri + (size_t)s * m * 2;
        double* f = F.data() + (size_t)s * mm;
        for (uint32_t i = 0; i < m; ++i) {
            const double air = a[2 * i], aii = a[2 * i + 1];
            f[i * m + i] = air * air + aii * aii;
            for (uint32_t j = i + 1; j < m; ++j) {
                const double ajr = a[2 * j], aji = a[2 * j + 1];
                // conj(a_i) a_j = (air - i aii)(ajr + i aji)
                f[i * m + j] = air * ajr + aii * aji;
                f[j * m + i] = air * aji - aii * ajr;
            }
        }
    }
}

// MFMA A-operand image of the table: FA[tile][lane][s], lane = (g = lane>>4, rho = lane&15):
//   value = F[bin = 16*tile + 4*(rho&3) + (rho>>2)][e = 4*s + g]      (0 for e >= m*m)
// so that accumulator register r of lane (g, c) holds bin 16*tile + 4*g + r (see scan_mfma_kernel).
// Bins >= res are NaN: their d is NaN, which never enters a top-n list and is never stored.
void build_FA(const std::vector<double>& F, uint32_t m, uint32_t res, uint32_t tiles, std::vector<double>& FA)
{
    const uint32_t mm = m * m;
    const uint32_t ks = (mm + 3) / 4;
    const double nan = std::nan("");
    FA.assign((size_t)(tiles + 1) * 64 * ks, nan);
    for (uint32_t t = 0; t < tiles; ++t)
        for (uint32_t lane = 0; lane < 64; ++lane) {
            const uint32_t g = lane >> 4, rho = lane & 15;
            const uint32_t bin = 16 * t + 4 * (rho & 3) + (rho >> 2);
            for (uint32_t s = 0; s < ks; ++s) {
                const uint32_t e = 4 * s + g;
                double v = nan;
                if (bin < res) v = (e < mm) ? F[(size_t)bin * mm + e] : 0.0;
                FA[((size_t)t * 64 + lane) * ks + s] = v;
            }
        }
}

uint32_t round_up(uint32_t v, uint32_t a) { return (v + a - 1) / a * a; }

int ensure_workspace(baz_music_ctx* c, uint32_t batch)
{
    if (batch <= c->cap) return BAZ_MUSIC_OK;
    const uint32_t cap = round_up(batch, 64);
    const size_t mm = (size_t)c->m * c->m;
    if (c->dR) { (void)hipFree(c->dR); c->dR = nullptr; }
    if (c->dQ) { (void)hipFree(c->dQ); c->dQ = nullptr; }
    c->cap = 0;
    HIP_TRY(c, hipMalloc((void**)&c->dR, (size_t)cap * mm * sizeof(double2)));
    HIP_TRY(c, hipMalloc((void**)&c->dQ, (size_t)cap * mm * sizeof(double)));
    c->cap = cap;
    return BAZ_MUSIC_OK;
}

struct ProfScope {
    baz_music_ctx* c;
    int stage;
    hipEvent_t stop = nullptr;
    ProfScope(baz_music_ctx* ctx, int st) : c(ctx), stage(st)
    {
        if (!c->profiling) return;
        StageProf& p = c->prof[stage];
        if (p.used + 2 > p.ev.size()) {
            for (int k = 0; k < 2; ++k) {
                hipEvent_t e;
                if (hipEventCreate(&e) != hipSuccess) return;
                p.ev.push_back(e);
            }
        }
        (void)hipEventRecord(p.ev[p.used], c->stream);
        stop = p.ev[p.used + 1];
        p.used += 2;
    }
    ~ProfScope()
    {
        if (stop) (void)hipEventRecord(stop, c->stream);
    }
};

void prof_collect(baz_music_ctx* c)
{
    for (int s = 0; s < BAZ_MUSIC_NUM_STAGES; ++s) {
        StageProf& p = c->prof[s];
        for (size_t k = 0; k + 1 < p.used; k += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p.ev[k], p.ev[k + 1]) == hipSuccess) {
                p.total_ms += ms;
                p.launches += 1;
            }
        }
        p.used = 0;
    }
}

// ---- kernel dispatch -------------------------------------------------------------------

template <int M>
int launch_cov_t(baz_music_ctx* c, const float* d_in, uint32_t batch, double2* dR)
{
    constexpr int IPT = 16 / (2 * M);
    const uint32_t ntiles = (batch + IPT - 1) / IPT;
    uint32_t blocks = (ntiles + 3) / 4;
    blocks = std::min<uint32_t>(blocks, 256u * 32u);
    hipLaunchKernelGGL((cov_mfma_kernel<M>), dim3(blocks), dim3(256), 0, c->stream, d_in, dR, batch, c->K);
    HIP_TRY(c, hipGetLastError());
    return BAZ_MUSIC_OK;
}

int launch_cov(baz_music_ctx* c, const float* d_in, uint32_t batch, double2* dR)
{
    ProfScope ps(c, BAZ_MUSIC_STAGE_COV);
    switch (c->m) {
        case 2: return launch_cov_t<2>(c, d_in, batch, dR);
        case 3: return launch_cov_t<3>(c, d_in, batch, dR);
        case 4: return launch_cov_t<4>(c, d_in, batch, dR);
        case 5: return launch_cov_t<5>(c, d_in, batch, dR);
        case 6: return launch_cov_t<6>(c, d_in, batch, dR);
        case 7: return launch_cov_t<7>(c, d_in, batch, dR);
        case 8: return launch_cov_t<8>(c, d_in, batch, dR);
        default: return BAZ_MUSIC_E_UNSUPPORTED;
    }
}

template <int M>
int launch_evd_t(baz_music_ctx* c, const double2* dR, uint32_t batch, double* dQ, uint32_t qstride)
{
    const uint32_t blocks = (batch + 63) / 64;
    hipLaunchKernelGGL((evd_proj_kernel<M>), dim3(blocks), dim3(64), 0, c->stream, dR, dQ, batch, c->n, qstride);
    HIP_TRY(c, hipGetLastError());
    return BAZ_MUSIC_OK;
}

int launch_evd(baz_music_ctx* c, const double2* dR, uint32_t batch, double* dQ, uint32_t qstride)
{
    ProfScope ps(c, BAZ_MUSIC_STAGE_EVD);
    switch (c->m) {
        case 2: return launch_evd_t<2>(c, dR, batch, dQ, qstride);
        case 3: return launch_evd_t<3>(c, dR, batch, dQ, qstride);
        case 4: return launch_evd_t<4>(c, dR, batch, dQ, qstride);
        case 5: return launch_evd_t<5>(c, dR, batch, dQ, qstride);
        case 6: return launch_evd_t<6>(c, dR, batch, dQ, qstride);
        case 7: return launch_evd_t<7>(c, dR, batch, dQ, qstride);
        case 8: return launch_evd_t<8>(c, dR, batch, dQ, qstride);
        default: return BAZ_MUSIC_E_UNSUPPORTED;
    }
}

// How many bin ranges to split each item group into so that the launch fills the chip
// (>= ~8 waves per SIMD) even for small batches / long tables (config 3: 4096 items x 36000 bins).
uint32_t pick_nsplit(uint32_t groups, uint32_t ntiles)
{
    const uint32_t want_waves = 256u * 4u * 8u;
    uint32_t ns = (want_waves + groups - 1) / groups;
    ns = std::max<uint32_t>(1u, std::min<uint32_t>(ns, std::max<uint32_t>(1u, ntiles / 8u)));
    return std::min<uint32_t>(ns, 64u);
}

int ensure_candidates(baz_music_ctx* c, size_t entries)
{
    if (entries <= c->cand_cap) return BAZ_MUSIC_OK;
    if (c->dCandD) { (void)hipFree(c->dCandD); c->dCandD = nullptr; }
    if (c->dCandB) { (void)hipFree(c->dCandB); c->dCandB = nullptr; }
    c->cand_cap = 0;
    HIP_TRY(c, hipMalloc((void**)&c->dCandD, entries * sizeof(double)));
    HIP_TRY(c, hipMalloc((void**)&c->dCandB, entries * sizeof(uint32_t)));
    c->cand_cap = entries;
    return BAZ_MUSIC_OK;
}

template <int M, int NMAX, int IT>
int launch_scan_mfma(baz_music_ctx* c, const double* dQ, uint32_t qstride, uint32_t batch, float* d_ang,
                     float* d_lvl, float* d_spec)
{
    constexpr uint32_t ITEMS = 16 * IT;
    const uint32_t groups = (batch + ITEMS - 1) / ITEMS;
    const uint32_t ntiles = c->fa_tiles;
    const uint32_t nsplit = pick_nsplit(groups, ntiles);
    const uint32_t waves = groups * nsplit;
    const uint32_t blocks = (waves + 3) / 4;
    if (nsplit > 1) {
        int r = ensure_candidates(c, (size_t)batch * nsplit * NMAX);
        if (r) return r;
    }
    const bool spec = d_spec != nullptr;
    const bool vec4 = (c->res % 4u) == 0 && (reinterpret_cast<uintptr_t>(d_spec) % 16u) == 0;
#define BAZ_SCAN_LAUNCH(SPEC, VEC4)                                                                        \
    hipLaunchKernelGGL((scan_mfma_kernel<M, NMAX, IT, SPEC, VEC4>), dim3(blocks), dim3(256), 0, c->stream, \
                       dQ, c->dFA, d_spec, d_ang, d_lvl, c->dCandD, c->dCandB, batch, c->res, c->n,        \
                       qstride, ntiles, nsplit)
    if (spec && vec4) BAZ_SCAN_LAUNCH(true, true);
    else if (spec) BAZ_SCAN_LAUNCH(true, false);
    else BAZ_SCAN_LAUNCH(false, false);
#undef BAZ_SCAN_LAUNCH
    HIP_TRY(c, hipGetLastError());
    if (nsplit > 1) {
        hipLaunchKernelGGL((topn_merge_kernel<NMAX>), dim3((batch + 255) / 256), dim3(256), 0, c->stream,
                           c->dCandD, c->dCandB, d_ang, d_lvl, batch, c->res, c->n, nsplit);
        HIP_TRY(c, hipGetLastError());
    }
    return BAZ_MUSIC_OK;
}

template <int M, int NMAX>
int launch_scan_t(baz_music_ctx* c, const double* dQ, uint32_t qstride, uint32_t batch, float* d_ang,
                  float* d_lvl, float* d_spec)
{
    // item tiles per wave: bounded by the VGPR budget (B operand IT*KS pairs + IT accumulators + lists)
    constexpr int IT = (M <= 4) ? ((NMAX <= 2) ? 4 : ((NMAX <= 4) ? 2 : 1)) : ((M <= 6 && NMAX <= 4) ? 2 : 1);
    return launch_scan_mfma<M, NMAX, IT>(c, dQ, qstride, batch, d_ang, d_lvl, d_spec);
}

template <int M>
int launch_scan_m(baz_music_ctx* c, const double* dQ, uint32_t qstride, uint32_t batch, float* d_ang,
                  float* d_lvl, float* d_spec)
{
    if (c->n <= 2) return launch_scan_t<M, 2>(c, dQ, qstride, batch, d_ang, d_lvl, d_spec);
    if (c->n <= 4) return launch_scan_t<M, 4>(c, dQ, qstride, batch, d_ang, d_lvl, d_spec);
    return launch_scan_t<M, 8>(c, dQ, qstride, batch, d_ang, d_lvl, d_spec);
}

int launch_scan(baz_music_ctx* c, const double* dQ, uint32_t qstride, uint32_t batch, float* d_ang,
                float* d_lvl, float* d_spec)
{
    ProfScope ps(c, BAZ_MUSIC_STAGE_SCAN);
    switch (c->m) {
        case 2: return launch_scan_m<2>(c, dQ, qstride, batch, d_ang, d_lvl, d_spec);
        case 3: return launch_scan_m<3>(c, dQ, qstride, batch, d_ang, d_lvl, d_spec);
        case 4: return launch_scan_m<4>(c, dQ, qstride, batch, d_ang, d_lvl, d_spec);
        case 5: return launch_scan_m<5>(c, dQ, qstride, batch, d_ang, d_lvl, d_spec);
        case 6: return launch_scan_m<6>(c, dQ, qstride, batch, d_ang, d_lvl, d_spec);
        case 7: return launch_scan_m<7>(c, dQ, qstride, batch, d_ang, d_lvl, d_spec);
        case 8: return launch_scan_m<8>(c, dQ, qstride, batch, d_ang, d_lvl, d_spec);
        default: return BAZ_MUSIC_E_UNSUPPORTED;
    }
}

int upload_table(baz_music_ctx* c, const float* table_ri)
{
    std::vector<double> F;
    build_F(table_ri, c->m, c->res, F);
    std::vector<double> FA;
    build_FA(F, c->m, c->res, c->fa_tiles, FA);
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // no batch in flight reads the old table
    HIP_TRY(c, hipMemcpy(c->dFA, FA.data(), FA.size() * sizeof(double), hipMemcpyHostToDevice));
    return BAZ_MUSIC_OK;
}

int process_device_locked(baz_music_ctx* c, const void* d_in, uint32_t batch, void* d_ang, void* d_lvl,
                          void* d_spec)
{
    int r = ensure_workspace(c, batch);
    if (r) return r;
    const uint32_t qstride = baz_music_q_stride(batch);
    r = launch_cov(c, static_cast<const float*>(d_in), batch, c->dR);
    if (r) return r;
    r = launch_evd(c, c->dR, batch, c->dQ, qstride);
    if (r) return r;
    return launch_scan(c, c->dQ, qstride, batch, static_cast<float*>(d_ang), static_cast<float*>(d_lvl),
                       static_cast<float*>(d_spec));
}

}  // namespace

extern "C" {

int baz_music_create(baz_music_ctx** out, uint32_t m, uint32_t n, uint32_t nsamples, uint32_t resolution,
                     const float* table_ri, int device_id)
{
    if (!out) return BAZ_MUSIC_E_INVALID;
    *out = nullptr;
    // lib/baz_music_doa.cc:45-50 (asserts, compiled out in Release) made real; n == m underflows .cc:93
    if (m == 0 || n == 0 || n >= m || nsamples == 0 || (nsamples % m) != 0 || resolution == 0 || !table_ri)
        return BAZ_MUSIC_E_INVALID;
    if (m > BAZ_MUSIC_MAX_M || n > BAZ_MUSIC_MAX_N) return BAZ_MUSIC_E_UNSUPPORTED;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return BAZ_MUSIC_E_NODEVICE;
    int dev = device_id;
    if (dev < 0) {
        if (hipGetDevice(&dev) != hipSuccess) return BAZ_MUSIC_E_NODEVICE;
    }
    if (dev >= ndev) return BAZ_MUSIC_E_NODEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return BAZ_MUSIC_E_NODEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return BAZ_MUSIC_E_NODEVICE;   // kernels are gfx950-only

    baz_music_ctx* c = new (std::nothrow) baz_music_ctx;
    if (!c) return BAZ_MUSIC_E_NOMEM;
    c->m = m; c->n = n; c->nsamples = nsamples; c->res = resolution; c->K = nsamples / m;
    c->device = dev;
    DeviceGuard guard(dev);
    int r = BAZ_MUSIC_OK;
    do {
        if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { r = BAZ_MUSIC_E_HIP; break; }
        c->stream = c->own_stream;
        c->fa_tiles = (resolution + 15) / 16;
        if (hipMalloc((void**)&c->dFA, (size_t)(c->fa_tiles + 1) * 64 * ((m * m + 3) / 4) * sizeof(double)) != hipSuccess) { r = BAZ_MUSIC_E_NOMEM; break; }
        r = upload_table(c, table_ri);
    } while (0);
    if (r != BAZ_MUSIC_OK) {
        baz_music_destroy(c);
        return r;
    }
    char buf[128];
    snprintf(buf, sizeof(buf), "bazmusic::cov_mfma_kernel<%u>", m);
    c->stage_name[BAZ_MUSIC_STAGE_COV] = buf;
    snprintf(buf, sizeof(buf), "bazmusic::evd_proj_kernel<%u>", m);
    c->stage_name[BAZ_MUSIC_STAGE_EVD] = buf;
    snprintf(buf, sizeof(buf), "bazmusic::scan_mfma_kernel<%u,", m);
    c->stage_name[BAZ_MUSIC_STAGE_SCAN] = buf;
    *out = c;
    return BAZ_MUSIC_OK;
}

void baz_music_destroy(baz_music_ctx* c)
{
    if (!c) return;
    {
        DeviceGuard guard(c->device);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        for (auto& p : c->prof)
            for (auto e : p.ev) (void)hipEventDestroy(e);
        if (c->dFA) (void)hipFree(c->dFA);
        if (c->dCandD) (void)hipFree(c->dCandD);
        if (c->dCandB) (void)hipFree(c->dCandB);
        if (c->dR) (void)hipFree(c->dR);
        if (c->dQ) (void)hipFree(c->dQ);
        if (c->s_in) (void)hipFree(c->s_in);
        if (c->s_ang) (void)hipFree(c->s_ang);
        if (c->s_lvl) (void)hipFree(c->s_lvl);
        if (c->s_spec) (void)hipFree(c->s_spec);
        if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    }
    delete c;
}

int baz_music_set_table(baz_music_ctx* c, const float* table_ri)
{
    if (!c || !table_ri) return BAZ_MUSIC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);   // .cc:67
    DeviceGuard guard(c->device);
    return upload_table(c, table_ri);
}

int baz_music_set_stream(baz_music_ctx* c, void* hip_stream)
{
    if (!c) return BAZ_MUSIC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    return BAZ_MUSIC_OK;
}

int baz_music_sync(baz_music_ctx* c)
{
    if (!c) return BAZ_MUSIC_E_INVALID;
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return BAZ_MUSIC_OK;
}

int baz_music_reserve(baz_music_ctx* c, uint32_t max_batch)
{
    if (!c || max_batch == 0) return BAZ_MUSIC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return ensure_workspace(c, max_batch);
}

uint32_t baz_music_q_stride(uint32_t batch) { return round_up(batch ? batch : 1, 64); }

int baz_music_process_device(baz_music_ctx* c, const void* d_in, uint32_t batch, void* d_ang, void* d_lvl,
                             void* d_spec)
{
    if (!c || !d_in || !d_ang) return BAZ_MUSIC_E_INVALID;
    if (batch == 0) return BAZ_MUSIC_OK;
    std::lock_guard<std::mutex> lk(c->mtx);   // .cc:101
    DeviceGuard guard(c->device);
    return process_device_locked(c, d_in, batch, d_ang, d_lvl, d_spec);
}

int baz_music_process(baz_music_ctx* c, const float* in_ri, uint32_t batch, float* ang, float* lvl,
                      float* spectrum)
{
    if (!c || !in_ri || !ang) return BAZ_MUSIC_E_INVALID;
    if (batch == 0) return 0;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);

    // chunk so that device staging stays below ~512 MiB
    const size_t per_item = (size_t)c->nsamples * 8 + (size_t)c->res * 4 + (size_t)c->n * 8;
    uint32_t chunk = (uint32_t)std::max<size_t>(1, std::min<size_t>(batch, (512u << 20) / per_item));
    const bool want_spec = spectrum != nullptr;
    if (chunk > c->s_cap || (want_spec && !c->s_has_spec)) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->s_in) (void)hipFree(c->s_in);
        if (c->s_ang) (void)hipFree(c->s_ang);
        if (c->s_lvl) (void)hipFree(c->s_lvl);
        if (c->s_spec) (void)hipFree(c->s_spec);
        c->s_in = c->s_ang = c->s_lvl = c->s_spec = nullptr;
        c->s_cap = 0;
        const uint32_t cap = std::max(chunk, c->s_cap);
        HIP_TRY(c, hipMalloc((void**)&c->s_in, (size_t)cap * c->nsamples * 8));
        HIP_TRY(c, hipMalloc((void**)&c->s_ang, (size_t)cap * c->n * 4));
        HIP_TRY(c, hipMalloc((void**)&c->s_lvl, (size_t)cap * c->n * 4));
        if (want_spec || c->s_has_spec) {
            HIP_TRY(c, hipMalloc((void**)&c->s_spec, (size_t)cap * c->res * 4));
            c->s_has_spec = true;
        }
        c->s_cap = cap;
    }
    for (uint32_t done = 0; done < batch; done += chunk) {
        const uint32_t nb = std::min(chunk, batch - done);
        HIP_TRY(c, hipMemcpyAsync(c->s_in, in_ri + (size_t)done * c->nsamples * 2, (size_t)nb * c->nsamples * 8,
                                  hipMemcpyHostToDevice, c->stream));
        int r = process_device_locked(c, c->s_in, nb, c->s_ang, c->s_lvl, want_spec ? c->s_spec : nullptr);
        if (r) return r;
        HIP_TRY(c, hipMemcpyAsync(ang + (size_t)done * c->n, c->s_ang, (size_t)nb * c->n * 4, hipMemcpyDeviceToHost, c->stream));
        if (lvl)
            HIP_TRY(c, hipMemcpyAsync(lvl + (size_t)done * c->n, c->s_lvl, (size_t)nb * c->n * 4, hipMemcpyDeviceToHost, c->stream));
        if (want_spec)
            HIP_TRY(c, hipMemcpyAsync(spectrum + (size_t)done * c->res, c->s_spec, (size_t)nb * c->res * 4,
                                      hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return (int)batch;
}

int baz_music_profile(baz_music_ctx* c, int enable)
{
    if (!c) return BAZ_MUSIC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (enable) {
        for (auto& p : c->prof) { p.used = 0; p.total_ms = 0.0; p.launches = 0; }
        c->profiling = true;
    } else {
        prof_collect(c);
        c->profiling = false;
    }
    return BAZ_MUSIC_OK;
}

int baz_music_stage_ms(baz_music_ctx* c, int stage, double* total_ms, uint64_t* launches)
{
    if (!c || stage < 0 || stage >= BAZ_MUSIC_NUM_STAGES) return BAZ_MUSIC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    prof_collect(c);
    if (total_ms) *total_ms = c->prof[stage].total_ms;
    if (launches) *launches = c->prof[stage].launches;
    return BAZ_MUSIC_OK;
}

const char* baz_music_stage_name(baz_music_ctx* c, int stage)
{
    if (!c || stage < 0 || stage >= BAZ_MUSIC_NUM_STAGES) return "";
    return c->stage_name[stage].c_str();
}

int baz_music_debug_cov(baz_music_ctx* c, const void* d_in, uint32_t batch, void* d_R)
{
    if (!c || !d_in || !d_R || batch == 0) return BAZ_MUSIC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    return launch_cov(c, static_cast<const float*>(d_in), batch, static_cast<double2*>(d_R));
}

int baz_music_debug_evd(baz_music_ctx* c, const void* d_R, uint32_t batch, void* d_Q)
{
    if (!c || !d_R || !d_Q || batch == 0) return BAZ_MUSIC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    return launch_evd(c, static_cast<const double2*>(d_R), batch, static_cast<double*>(d_Q), baz_music_q_stride(batch));
}

uint64_t baz_music_bytes_per_item(const baz_music_ctx* c, int with_spectrum)
{
    if (!c) return 0;
    return 8ull * c->nsamples + 8ull * c->n + (with_spectrum ? 4ull * c->res : 0ull);
}

const char* baz_music_strerror(int code)
{
    switch (code) {
        case BAZ_MUSIC_OK: return "ok";
        case BAZ_MUSIC_E_INVALID: return "invalid argument";
        case BAZ_MUSIC_E_NOMEM: return "out of memory";
        case BAZ_MUSIC_E_HIP: return "HIP runtime error";
        case BAZ_MUSIC_E_UNSUPPORTED: return "configuration not supported by the gfx950 kernels";
        case BAZ_MUSIC_E_NODEVICE: return "no usable gfx950 device";
        default: return "unknown error";
    }
}

const char* baz_music_last_hip_error(const baz_music_ctx* c) { return c ? c->hip_err : ""; }

const char* baz_music_version(void) { return "gr_baz_amd/baz_music_hip 0.1 (gfx950)"; }

}  // extern "C"
