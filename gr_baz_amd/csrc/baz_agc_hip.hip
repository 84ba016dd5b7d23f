// baz_agc_hip.hip -- C-ABI (include/baz_agc_hip.h) over the gfx950 AGC kernels.
// Host-side counterpart of baz_agc_cc's state and work() (/root/reference/lib/baz_agc_cc.cc:50-102).
// No CPU arithmetic fallback: without a gfx950 device baz_agc_create() fails.
#include "../../include/baz_agc_hip.h"
#include "agc_kernels.hip.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>

using namespace bazagc;

struct baz_agc_ctx {
    uint32_t nstreams = 0;
    AgcParams P{};
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    uint64_t count = 0;            // _count (identical for every stream of the context)
    double* d_env = nullptr;       // _env per stream
    double* d_pw = nullptr;        // per-lane powers of a^IE for the fast path (AgcParams::pw)
    double2* d_pair = nullptr;     // per-chunk maps
    double* d_carry = nullptr;     // per-chunk carry-in
    size_t chunk_cap = 0;          // chunks per stream the workspace holds
    float *s_in = nullptr, *s_out = nullptr, *s_env = nullptr, *s_mul = nullptr;   // host-path staging
    size_t s_cap = 0;              // samples (all streams)
    std::mutex mtx;
};

namespace {

#define AGC_TRY(call)                                   \
    do {                                                \
        if ((call) != hipSuccess) return BAZ_AGC_E_HIP; \
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

int ensure_chunks(baz_agc_ctx* c, size_t nchunks)
{
    if (nchunks <= c->chunk_cap) return BAZ_AGC_OK;
    // launches of earlier calls may still use the tables freed below: drain them first (round 6: nothing relies on hipFree synchronising)
    if (c->chunk_cap) AGC_TRY(hipStreamSynchronize(c->stream));
    if (c->d_pair) (void)hipFree(c->d_pair);
    if (c->d_carry) (void)hipFree(c->d_carry);
    c->d_pair = nullptr; c->d_carry = nullptr; c->chunk_cap = 0;
    AGC_TRY(hipMalloc((void**)&c->d_pair, nchunks * c->nstreams * sizeof(double2)));
    AGC_TRY(hipMalloc((void**)&c->d_carry, nchunks * c->nstreams * sizeof(double)));
    c->chunk_cap = nchunks;
    return BAZ_AGC_OK;
}

int process_device_locked(baz_agc_ctx* c, const void* d_in, uint64_t n, uint64_t stride, void* d_out, void* d_env,
                          void* d_mul)
{
    // planar form: waves dealt over (stream, tile) pairs, 16 waves per workgroup
    const uint64_t ntiles64 = (n + AGC_IT - 1) / AGC_IT;
    if (ntiles64 > 0x7FFFFFFFull) return BAZ_AGC_E_INVALID;
    const uint32_t ntiles = (uint32_t)ntiles64;
    int r = ensure_chunks(c, ntiles);
    if (r) return r;
    const float2* in = static_cast<const float2*>(d_in);
    const uint64_t nwaves = (uint64_t)ntiles * c->nstreams;
    const uint64_t nblocks = (nwaves + 15) / 16;
    if (nblocks > 0x7FFFFFFFull) return BAZ_AGC_E_INVALID;
    const dim3 grid((uint32_t)nblocks), block(1024);
    hipLaunchKernelGGL((agc_tile_kernel<0, true>), grid, block, 0, c->stream, in, n, stride, c->P, c->d_pair,
                       (const double*)nullptr, ntiles, (float2*)nullptr, (double*)nullptr, c->nstreams, (float*)nullptr,
                       (float*)nullptr);
    hipLaunchKernelGGL(agc_carry_kernel, dim3(c->nstreams), dim3(AGC_CARRY_THREADS), 0, c->stream, in, stride, c->d_pair,
                       c->d_carry, ntiles, c->d_env, c->count == 0 ? 1 : 0);
    hipLaunchKernelGGL((agc_tile_kernel<2, true>), grid, block, 0, c->stream, in, n, stride, c->P, c->d_pair,
                       c->d_carry, ntiles, static_cast<float2*>(d_out), c->d_env, c->nstreams, static_cast<float*>(d_env),
                       static_cast<float*>(d_mul));
    AGC_TRY(hipGetLastError());
    c->count += n;
    return BAZ_AGC_OK;
}

// Interleaved-output form: out[t * nstreams + s] (MUSIC item layout), tiles of AGC_IT samples, nstreams <= 16.
int process_device_interleaved_locked(baz_agc_ctx* c, const void* d_in, uint64_t n, uint64_t stride, void* d_items)
{
    if (c->nstreams > (uint32_t)AGC_IMAX) return BAZ_AGC_E_INVALID;
    const uint64_t ntiles64 = (n + AGC_IT - 1) / AGC_IT;
    if (ntiles64 > 0x7FFFFFFFull) return BAZ_AGC_E_INVALID;
    const uint32_t ntiles = (uint32_t)ntiles64;
    int r = ensure_chunks(c, ntiles);
    if (r) return r;
    const float2* in = static_cast<const float2*>(d_in);
    const dim3 block(64 * c->nstreams);
    const size_t lds = (size_t)c->nstreams * (AGC_IT + 1) * sizeof(float2);
    hipLaunchKernelGGL((agc_tile_kernel<0>), dim3(ntiles), block, 0, c->stream, in, n, stride, c->P, c->d_pair,
                       (const double*)nullptr, ntiles, (float2*)nullptr, (double*)nullptr, c->nstreams);
    hipLaunchKernelGGL(agc_carry_kernel, dim3(c->nstreams), dim3(AGC_CARRY_THREADS), 0, c->stream, in, stride, c->d_pair, c->d_carry,
                       ntiles, c->d_env, c->count == 0 ? 1 : 0);
    hipLaunchKernelGGL((agc_tile_kernel<1>), dim3(ntiles), block, lds, c->stream, in, n, stride, c->P, c->d_pair,
                       c->d_carry, ntiles, static_cast<float2*>(d_items), c->d_env, c->nstreams);
    AGC_TRY(hipGetLastError());
    c->count += n;
    return BAZ_AGC_OK;
}

}  // namespace

extern "C" {

int baz_agc_create(baz_agc_ctx** out, uint32_t nstreams, float rate, float reference, float gain, float max_gain,
                   int device_id)
{
    (void)gain; (void)max_gain;   // only used by the reference's dead code (lib/baz_agc_cc.cc:103-149)
    if (!out) return BAZ_AGC_E_INVALID;
    *out = nullptr;
    if (nstreams == 0 || nstreams > 65535u) return BAZ_AGC_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return BAZ_AGC_E_NODEVICE;
    int dev = device_id;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return BAZ_AGC_E_NODEVICE;
    if (dev >= ndev) return BAZ_AGC_E_NODEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return BAZ_AGC_E_NODEVICE;
    baz_agc_ctx* c = new (std::nothrow) baz_agc_ctx;
    if (!c) return BAZ_AGC_E_NOMEM;
    c->nstreams = nstreams;
    c->P.a = 1.0 - (double)rate;          // lib/baz_agc_cc.cc:82  (1.0 - _rate), _rate is a float member
    c->P.b = (double)rate;
    c->P.reference = (double)reference;   // _reference is a double member initialised from the float argument
    c->device = dev;
    DeviceGuard guard(dev);
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void**)&c->d_env, nstreams * sizeof(double)) != hipSuccess ||
        // (hipMemset of device memory may return before the fill has run, and the null stream does not order it against this context's
        // non-blocking stream: the fill goes on the stream that will use the buffer -- round 6, found with the MUSIC library's guard allocator)
        hipMemsetAsync(c->d_env, 0, nstreams * sizeof(double), c->own_stream) != hipSuccess) {
        baz_agc_destroy(c);
        return BAZ_AGC_E_HIP;
    }
    c->stream = c->own_stream;
    // The fast path of full tiles (agc_kernels.hip.h): powers of a^IE, computed in long double and rounded once.  Only for
    // an ordinary smoothing rate and reference (0 < a < 1 and the quotient reference / envelope far from the range's ends);
    // BAZ_AGC_FAST=0 keeps the general path (lab / A-B tests).
    c->P.pw = nullptr;
    const char* fenv = getenv("BAZ_AGC_FAST");
    const double ar = std::fabs(c->P.reference);
    if (!(fenv && atoi(fenv) == 0) && c->P.a > 0.0 && c->P.a < 1.0 && c->P.b > 0.0 && ar >= 0x1p-100 && ar <= 0x1p100) {
        const long double a4 = powl((long double)c->P.a, (long double)AGC_IE);
        c->P.c1 = (double)a4;
        c->P.c2 = (double)powl(a4, 2.0L);
        c->P.c4 = (double)powl(a4, 4.0L);
        c->P.c8 = (double)powl(a4, 8.0L);
        c->P.a_tile = (double)powl(a4, 64.0L);
        double h[64 * 4];
        for (int l = 0; l < 64; ++l) {
            h[4 * l + 0] = (double)powl(a4, (long double)((l & 15) + 1));
            h[4 * l + 1] = (double)powl(a4, (long double)((l & 31) + 1));
            h[4 * l + 2] = (double)powl(a4, (long double)l);
            h[4 * l + 3] = 0.0;
        }
        if (hipMalloc((void**)&c->d_pw, sizeof(h)) != hipSuccess || hipMemcpy(c->d_pw, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) {
            baz_agc_destroy(c);
            return BAZ_AGC_E_HIP;
        }
        c->P.pw = c->d_pw;
    }
    *out = c;
    return BAZ_AGC_OK;
}

void baz_agc_destroy(baz_agc_ctx* c)
{
    if (!c) return;
    {
        DeviceGuard guard(c->device);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        if (c->d_env) (void)hipFree(c->d_env);
        if (c->d_pw) (void)hipFree(c->d_pw);
        if (c->d_pair) (void)hipFree(c->d_pair);
        if (c->d_carry) (void)hipFree(c->d_carry);
        if (c->s_in) (void)hipFree(c->s_in);
        if (c->s_out) (void)hipFree(c->s_out);
        if (c->s_env) (void)hipFree(c->s_env);
        if (c->s_mul) (void)hipFree(c->s_mul);
        if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    }
    delete c;
}

int baz_agc_process_device(baz_agc_ctx* c, const void* d_in, uint64_t n, uint64_t stride, void* d_out, void* d_env,
                           void* d_mul)
{
    if (!c || !d_in || !d_out || (c->nstreams > 1 && stride < n)) return BAZ_AGC_E_INVALID;
    if (n == 0) return BAZ_AGC_OK;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    return process_device_locked(c, d_in, n, stride, d_out, d_env, d_mul);
}

int baz_agc_process_device_interleaved(baz_agc_ctx* c, const void* d_in, uint64_t n, uint64_t stride, void* d_items)
{
    if (!c || !d_in || !d_items || (c->nstreams > 1 && stride < n)) return BAZ_AGC_E_INVALID;
    if (n == 0) return BAZ_AGC_OK;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    return process_device_interleaved_locked(c, d_in, n, stride, d_items);
}

int baz_agc_process(baz_agc_ctx* c, const float* in_ri, uint64_t n, uint64_t stride, float* out_ri, float* env, float* mul)
{
    if (!c || !in_ri || !out_ri || (c->nstreams > 1 && stride < n)) return BAZ_AGC_E_INVALID;
    if (n == 0) return 0;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    const size_t total = (c->nstreams > 1) ? (size_t)stride * c->nstreams : (size_t)n;
    if (total > c->s_cap) {
        AGC_TRY(hipStreamSynchronize(c->stream));
        if (c->s_in) (void)hipFree(c->s_in);
        if (c->s_out) (void)hipFree(c->s_out);
        if (c->s_env) (void)hipFree(c->s_env);
        if (c->s_mul) (void)hipFree(c->s_mul);
        c->s_in = c->s_out = c->s_env = c->s_mul = nullptr;
        c->s_cap = 0;
        AGC_TRY(hipMalloc((void**)&c->s_in, total * 8));
        AGC_TRY(hipMalloc((void**)&c->s_out, total * 8));
        AGC_TRY(hipMalloc((void**)&c->s_env, total * 4));
        AGC_TRY(hipMalloc((void**)&c->s_mul, total * 4));
        c->s_cap = total;
    }
    const uint64_t st = (c->nstreams > 1) ? stride : n;
    AGC_TRY(hipMemcpyAsync(c->s_in, in_ri, total * 8, hipMemcpyHostToDevice, c->stream));
    int r = process_device_locked(c, c->s_in, n, st, c->s_out, env ? c->s_env : nullptr, mul ? c->s_mul : nullptr);
    if (r) return r;
    AGC_TRY(hipMemcpyAsync(out_ri, c->s_out, total * 8, hipMemcpyDeviceToHost, c->stream));
    if (env) AGC_TRY(hipMemcpyAsync(env, c->s_env, total * 4, hipMemcpyDeviceToHost, c->stream));
    if (mul) AGC_TRY(hipMemcpyAsync(mul, c->s_mul, total * 4, hipMemcpyDeviceToHost, c->stream));
    AGC_TRY(hipStreamSynchronize(c->stream));
    return (int)n;
}

int baz_agc_reset(baz_agc_ctx* c)
{
    if (!c) return BAZ_AGC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    AGC_TRY(hipStreamSynchronize(c->stream));
    AGC_TRY(hipMemsetAsync(c->d_env, 0, c->nstreams * sizeof(double), c->stream));     // stream-ordered before the next call's launches
    c->count = 0;
    return BAZ_AGC_OK;
}

int baz_agc_set_stream(baz_agc_ctx* c, void* hip_stream)
{
    if (!c) return BAZ_AGC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    AGC_TRY(hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    return BAZ_AGC_OK;
}

int baz_agc_sync(baz_agc_ctx* c)
{
    if (!c) return BAZ_AGC_E_INVALID;
    DeviceGuard guard(c->device);
    AGC_TRY(hipStreamSynchronize(c->stream));
    return BAZ_AGC_OK;
}

uint64_t baz_agc_count(const baz_agc_ctx* c) { return c ? c->count : 0; }

int baz_agc_debug_selfcheck(baz_agc_ctx* c, const void* d_a, const void* d_b, uint64_t n, uint64_t mismatches[2])
{
    if (!c || !d_a || !d_b || !mismatches || n == 0 || n > 0x7FFFFFFFull * 256ull) return BAZ_AGC_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    unsigned long long* d_bad = nullptr;
    AGC_TRY(hipMalloc((void**)&d_bad, 2 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d_bad, 0, 2 * sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(agc_selfcheck_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream,
                           static_cast<const double*>(d_a), static_cast<const double*>(d_b), n, d_bad);
        e = hipGetLastError();
    }
    unsigned long long h[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(h, d_bad, sizeof(h), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_bad);
    if (e != hipSuccess) return BAZ_AGC_E_HIP;
    mismatches[0] = h[0];
    mismatches[1] = h[1];
    return BAZ_AGC_OK;
}

const char* baz_agc_strerror(int code)
{
    switch (code) {
        case BAZ_AGC_OK: return "ok";
        case BAZ_AGC_E_INVALID: return "invalid argument";
        case BAZ_AGC_E_NOMEM: return "out of memory";
        case BAZ_AGC_E_HIP: return "HIP runtime error";
        case BAZ_AGC_E_NODEVICE: return "no usable gfx950 device";
        default: return "unknown error";
    }
}

}  // extern "C"
