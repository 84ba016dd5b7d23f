// baz_resamp_hip.hip -- C-ABI (include/baz_resamp_hip.h) over the gfx950 fractional-resampler kernel.
// Host-side counterpart of fractional_resampler_cc_impl's state, setters, forecast() and general_work()
// (/root/reference/lib/baz_fractional_resampler_cc.cc:80-101, 141-254).  No CPU arithmetic fallback: without a
// gfx950 device baz_resamp_create() fails.
#include "../../include/baz_resamp_hip.h"
#include "resamp_kernels.hip.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <locale.h>
#include <mutex>
#include <new>

using namespace bazresamp;

typedef unsigned __int128 u128;
typedef __int128 i128;

struct baz_resamp_ctx {
    bool stopped_at_bad = false;   // the last two-input call ended on an unusable ratio sample
    uint32_t nstreams = 0;
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // phase state in 64.64 fixed point (the reference's long double members, .cc:41-49)
    u128 mu = 0, mu_inc = 0;
    bool update = false;        u128 mu_inc_update = 0;
    bool update_mu = false;     u128 mu_update = 0;
    bool update_mu_adj = false; i128 mu_adj = 0;
    bool exact = true;
    float taps[(RS_NSTEPS + 1) * RS_NTAPS];
    float* d_taps = nullptr;
    float *s_in = nullptr, *s_out = nullptr;    // host-path staging (device)
    size_t s_in_cap = 0, s_out_cap = 0;         // complex samples
    // two-input branch (per-sample ratio input): phase table of the walk, its result, ratio staging
    uint32_t *d_ii = nullptr, *d_imu = nullptr;
    size_t walk_cap = 0;                        // outputs
    WalkResult* d_walk = nullptr;
    float* s_rr = nullptr;
    size_t s_rr_cap = 0;                        // floats
    std::mutex mtx;
};

namespace {

#define RS_TRY(call)                                       \
    do {                                                   \
        if ((call) != hipSuccess) return BAZ_RESAMP_E_HIP; \
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

const long double TWO64 = 18446744073709551616.0L;

// long double -> 64.64 fixed point; *exact = false when bits below 2^-64 were dropped
u128 to_fixed(long double v, bool* exact)
{
    const long double ip = floorl(v);
    const long double fr = (v - ip) * TWO64;          // exact: scaling by a power of two
    const long double frf = floorl(fr);
    if (frf != fr && exact) *exact = false;
    return ((u128)(uint64_t)ip << 64) | (u128)(uint64_t)frf;
}
i128 to_fixed_signed(long double v, bool* exact) { return v < 0 ? -(i128)to_fixed(-v, exact) : (i128)to_fixed(v, exact); }
long double from_fixed(u128 f) { return (long double)(uint64_t)(f >> 64) + (long double)(uint64_t)f / TWO64; }

bool ratio_ok(long double r) { return r >= 1.0L / 2048.0L && r <= 2147483648.0L; }

long double sincl_(long double x)
{
    const long double pi = 3.14159265358979323846264338327950288L;
    return x == 0.0L ? 1.0L : sinl(pi * x) / (pi * x);
}

// gnuradio-filter's MMSE interpolator table from its published criterion: for mu = i/128 the 8 taps h minimise
// the band-limited (|f| <= B = 1/4) squared error between sum_j h_j e^{-i 2 pi f j} and the ideal delay
// e^{-i 2 pi f (4 - mu)} -- linear least squares, normal equations A h = b with A_jl = 2B sinc(2B (j - l)),
// b_j = 2B sinc(2B (j - 4 + mu)).  (GNU Radio solves the same problem with a numerical optimiser and prints
// 6 digits; see the rounding below.)  Rows 0 and 128 are pure delays.
void build_taps(float* taps)
{
    // the decimal round trip below must read "." as the decimal point whatever locale the host application has set
    locale_t cloc = newlocale(LC_ALL_MASK, "C", (locale_t)0);
    locale_t prev = cloc ? uselocale(cloc) : (locale_t)0;
    const long double B = 0.25L;
    for (int i = 0; i <= RS_NSTEPS; ++i) {
        long double a[RS_NTAPS][RS_NTAPS + 1];
        const long double delay = 4.0L - (long double)i / RS_NSTEPS;
        for (int j = 0; j < RS_NTAPS; ++j) {
            for (int l = 0; l < RS_NTAPS; ++l) a[j][l] = 2 * B * sincl_(2 * B * (long double)(j - l));
            a[j][RS_NTAPS] = 2 * B * sincl_(2 * B * ((long double)j - delay));
        }
        for (int c = 0; c < RS_NTAPS; ++c) {
            int piv = c;
            for (int r = c + 1; r < RS_NTAPS; ++r)
                if (fabsl(a[r][c]) > fabsl(a[piv][c])) piv = r;
            if (piv != c)
                for (int k = 0; k <= RS_NTAPS; ++k) { const long double t = a[c][k]; a[c][k] = a[piv][k]; a[piv][k] = t; }
            for (int r = 0; r < RS_NTAPS; ++r) {
                if (r == c) continue;
                const long double f = a[r][c] / a[c][c];
                for (int k = c; k <= RS_NTAPS; ++k) a[r][k] -= f * a[c][k];
            }
        }
        // gnuradio-filter's interpolator_taps.h holds these values as printed by its generator ("%12.5e": six significant
        // digits) and compiled as float: the same decimal rounding here, then the float nearest to that decimal
        for (int j = 0; j < RS_NTAPS; ++j) {
            char dec[40];
            snprintf(dec, sizeof dec, "%.5Le", a[j][RS_NTAPS] / a[j][j]);
            taps[i * RS_NTAPS + j] = (float)strtod(dec, nullptr);
        }
    }
    if (cloc) { uselocale(prev); freelocale(cloc); }
    for (int j = 0; j < RS_NTAPS; ++j) taps[j] = taps[RS_NSTEPS * RS_NTAPS + j] = 0.0f;
    taps[4] = 1.0f;
    taps[RS_NSTEPS * RS_NTAPS + 3] = 1.0f;
}

// One general_work() on device buffers: applies the pending updates in the reference's order, launches, advances
// the phase state.  Returns outputs produced per stream.
int64_t process_device_locked(baz_resamp_ctx* c, const void* d_in, uint64_t in_stride, uint64_t ninput, void* d_out,
                              uint64_t out_stride, uint32_t noutput, uint64_t* consumed)
{
    if (consumed) *consumed = 0;
    c->stopped_at_bad = false;       // the one-input path changes the state: a later two-input call is a first call again
    if (noutput == 0) return 0;
    // .cc:165-170: a pending mu applies to the first output; .cc:175-181: a pending ratio from the first step on;
    // .cc:184-189: the adjustment is added to the first step
    const u128 first = c->update_mu ? c->mu_update : c->mu;
    const u128 inc = c->update ? c->mu_inc_update : c->mu_inc;
    const i128 adj = c->update_mu_adj ? c->mu_adj : 0;
    // P_0 = first; P_o = step1 + (o - 1) * inc for o >= 1, step1 = first + inc + adj (the phase after the first step)
    const i128 step1_s = (i128)first + (i128)inc + adj;
    if (step1_s < 0) return BAZ_RESAMP_E_INVALID;               // the reference would index before in[0]
    const u128 step1 = (u128)step1_s;
    // largest o whose 8-sample window [floor(P_o), floor(P_o)+7] lies inside [0, ninput):  P_o < ninput - 7
    if (ninput < RS_NTAPS) return 0;
    const u128 limit = (u128)(ninput - (RS_NTAPS - 1)) << 64;   // exclusive bound on P_o
    if (first >= limit) return 0;
    uint64_t n = 1;
    if (noutput > 1 && step1 < limit) {
        const u128 k = (limit - 1 - step1) / inc;               // largest j = o - 1 with step1 + j*inc <= limit - 1
        n = (k >= (u128)(noutput - 2)) ? noutput : (uint64_t)k + 2;
    }
    const u128 pend = step1 + (u128)(n - 1) * inc;               // phase after the n-th step
    if ((uint64_t)(pend >> 64) > 0xFFFFFFFFull * 2048ull) return BAZ_RESAMP_E_INVALID;
    PhaseParams p;
    p.first_lo = (uint64_t)first; p.first_hi = (uint64_t)(first >> 64);
    p.base_lo = (uint64_t)step1;  p.base_hi = (uint64_t)(step1 >> 64);
    p.inc_lo = (uint64_t)inc;     p.inc_hi = (uint64_t)(inc >> 64);
    const dim3 grid((uint32_t)((n + RS_BLOCK * RS_PER_THREAD - 1) / (RS_BLOCK * RS_PER_THREAD)), c->nstreams);
    hipLaunchKernelGGL(resamp_kernel, grid, dim3(RS_BLOCK), 0, c->stream, static_cast<const float2*>(d_in), in_stride,
                       static_cast<float2*>(d_out), out_stride, (uint32_t)n, p, c->d_taps);
    RS_TRY(hipGetLastError());
    c->update_mu = c->update = c->update_mu_adj = false;
    c->mu_inc = inc;
    c->mu = pend & (((u128)1 << 64) - 1);                        // d_mu = s - floor(s)
    if (consumed) *consumed = (uint64_t)(pend >> 64);            // consume_each(ii)
    return (int64_t)n;
}

// One general_work() of the TWO-input branch on device buffers (.cc:205-217): walk + table kernels, then the host
// reads the walk's result (the counts are data dependent), i.e. this call synchronises the stream.  The pending
// set_mu / set_resamp_ratio / adjustment flags are left pending: the reference only looks at them in the one-input
// branch (.cc:165-189).
int64_t process2_device_locked(baz_resamp_ctx* c, const void* d_in, uint64_t in_stride, uint64_t ninput,
                               const void* d_ratio, void* d_out, uint64_t out_stride, uint32_t noutput, uint64_t* consumed)
{
    if (consumed) *consumed = 0;
    if (noutput == 0 || ninput < RS_NTAPS) return 0;
    if (ninput > 0xFFFFFFFFull) return BAZ_RESAMP_E_UNSUPPORTED;    // the phase table holds 32-bit input indices
    if ((c->mu >> 64) != 0) return BAZ_RESAMP_E_INVALID;            // d_mu is a fraction between calls
    if (noutput > c->walk_cap) {
        // launches of earlier calls may still use the tables freed below: drain them first (round 6: nothing relies on hipFree synchronising)
        if (c->walk_cap) RS_TRY(hipStreamSynchronize(c->stream));
        if (c->d_ii) (void)hipFree(c->d_ii);
        if (c->d_imu) (void)hipFree(c->d_imu);
        c->d_ii = c->d_imu = nullptr; c->walk_cap = 0;
        RS_TRY(hipMalloc((void**)&c->d_ii, (size_t)noutput * 4));
        RS_TRY(hipMalloc((void**)&c->d_imu, (size_t)noutput * 4));
        c->walk_cap = noutput;
    }
    if (!c->d_walk) RS_TRY(hipMalloc((void**)&c->d_walk, sizeof(WalkResult)));
    hipLaunchKernelGGL(resamp_walk_kernel, dim3(1), dim3(256), 0, c->stream, static_cast<const float*>(d_ratio), ninput,
                       noutput, (uint64_t)c->mu, c->d_ii, c->d_imu, c->d_walk);
    RS_TRY(hipGetLastError());
    const dim3 grid((noutput + RS_BLOCK - 1) / RS_BLOCK, c->nstreams);
    hipLaunchKernelGGL(resamp_table_kernel, grid, dim3(RS_BLOCK), 0, c->stream, static_cast<const float2*>(d_in), in_stride,
                       static_cast<float2*>(d_out), out_stride, c->d_ii, c->d_imu, c->d_walk, c->d_taps);
    RS_TRY(hipGetLastError());
    WalkResult w;
    RS_TRY(hipMemcpyAsync(&w, c->d_walk, sizeof(w), hipMemcpyDeviceToHost, c->stream));
    RS_TRY(hipStreamSynchronize(c->stream));
    // The window STARTS on an unusable ratio sample and the PREVIOUS call already stopped on it (nothing was consumed since):
    // the one output this call could emit is the one already delivered; report it instead of looping.  A first
    // call -- or a control stream that begins with such a sample -- still delivers its output.
    const bool at_bad_start = (w.status == 1 && w.ii == 0 && w.n <= 1);
    if (at_bad_start && c->stopped_at_bad) return BAZ_RESAMP_E_INVALID;
    c->stopped_at_bad = (w.status == 1);
    c->mu = (u128)w.frac;
    if (w.last_bits) {                                              // d_mu_inc = the last ratio sample read (.cc:207,215)
        float r;
        std::memcpy(&r, &w.last_bits, sizeof(r));
        c->mu_inc = to_fixed((long double)r, &c->exact);
    }
    if (consumed) *consumed = w.ii;
    return (int64_t)w.n;
}

int ensure_staging(baz_resamp_ctx* c, size_t nin, size_t nout)
{
    if ((nin > c->s_in_cap && c->s_in_cap) || (nout > c->s_out_cap && c->s_out_cap)) RS_TRY(hipStreamSynchronize(c->stream));   // (as above)
    if (nin > c->s_in_cap) {
        if (c->s_in) (void)hipFree(c->s_in);
        c->s_in = nullptr; c->s_in_cap = 0;
        RS_TRY(hipMalloc((void**)&c->s_in, nin * 8));
        c->s_in_cap = nin;
    }
    if (nout > c->s_out_cap) {
        if (c->s_out) (void)hipFree(c->s_out);
        c->s_out = nullptr; c->s_out_cap = 0;
        RS_TRY(hipMalloc((void**)&c->s_out, nout * 8));
        c->s_out_cap = nout;
    }
    return BAZ_RESAMP_OK;
}

}  // namespace

extern "C" {

int baz_resamp_create(baz_resamp_ctx** out, uint32_t nstreams, double phase_shift, double resamp_ratio,
                      uint64_t num, uint64_t denom, int device_id)
{
    if (!out) return BAZ_RESAMP_E_INVALID;
    *out = nullptr;
    if (nstreams == 0) return BAZ_RESAMP_E_INVALID;
    long double ratio = (long double)resamp_ratio;
    if (denom != 0) ratio = (long double)num / (long double)denom;                     // .cc:89-92
    if (!(ratio > 0)) return BAZ_RESAMP_E_INVALID;                                     // .cc:94-95
    if (!(phase_shift >= 0 && phase_shift <= 1)) return BAZ_RESAMP_E_INVALID;          // .cc:96-97
    if (!ratio_ok(ratio)) return BAZ_RESAMP_E_UNSUPPORTED;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return BAZ_RESAMP_E_NODEVICE;
    int dev = device_id;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return BAZ_RESAMP_E_NODEVICE;
    if (dev >= ndev) return BAZ_RESAMP_E_INVALID;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return BAZ_RESAMP_E_NODEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return BAZ_RESAMP_E_NODEVICE;
    baz_resamp_ctx* c = new (std::nothrow) baz_resamp_ctx;
    if (!c) return BAZ_RESAMP_E_NOMEM;
    c->nstreams = nstreams;
    c->device = dev;
    c->mu = to_fixed((long double)phase_shift, &c->exact);
    c->mu_inc = to_fixed(ratio, &c->exact);
    build_taps(c->taps);
    DeviceGuard guard(dev);
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void**)&c->d_taps, sizeof(c->taps)) != hipSuccess ||
        hipMemcpy(c->d_taps, c->taps, sizeof(c->taps), hipMemcpyHostToDevice) != hipSuccess) {
        baz_resamp_destroy(c);
        return BAZ_RESAMP_E_HIP;
    }
    c->stream = c->own_stream;
    fprintf(stderr, "[fractional_resampler_cc<hip:%d>] Ratio: %.25Lf\n", dev, ratio);   // banner, .cc:92
    *out = c;
    return BAZ_RESAMP_OK;
}

void baz_resamp_destroy(baz_resamp_ctx* c)
{
    if (!c) return;
    {
        DeviceGuard guard(c->device);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        if (c->d_taps) (void)hipFree(c->d_taps);
        if (c->s_in) (void)hipFree(c->s_in);
        if (c->s_out) (void)hipFree(c->s_out);
        if (c->d_ii) (void)hipFree(c->d_ii);
        if (c->d_imu) (void)hipFree(c->d_imu);
        if (c->d_walk) (void)hipFree(c->d_walk);
        if (c->s_rr) (void)hipFree(c->s_rr);
        if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    }
    delete c;
}

int64_t baz_resamp_forecast(const baz_resamp_ctx* c, uint32_t noutput)
{
    if (!c) return BAZ_RESAMP_E_INVALID;
    return (int64_t)ceill((long double)noutput * from_fixed(c->mu_inc) + RS_NTAPS);    // .cc:146-148
}

int64_t baz_resamp_process_device(baz_resamp_ctx* c, const void* d_in, uint64_t in_stride, uint64_t ninput, void* d_out,
                                  uint64_t out_stride, uint32_t noutput, uint64_t* consumed)
{
    if (!c || !d_in || !d_out || in_stride < ninput || out_stride < noutput) return BAZ_RESAMP_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    return process_device_locked(c, d_in, in_stride, ninput, d_out, out_stride, noutput, consumed);
}

int64_t baz_resamp_process(baz_resamp_ctx* c, const float* in_ri, uint64_t in_stride, uint64_t ninput, float* out_ri,
                           uint64_t out_stride, uint32_t noutput, uint64_t* consumed)
{
    if (!c || !in_ri || !out_ri || in_stride < ninput || out_stride < noutput) return BAZ_RESAMP_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    // only the samples general_work can touch travel: ceil(noutput * ratio) + 8 + 2 (pending adjustment slack)
    const long double inc = from_fixed(c->update ? c->mu_inc_update : c->mu_inc);
    uint64_t need = (uint64_t)ceill((long double)noutput * inc + RS_NTAPS + 2 +
                                    (c->update_mu_adj && c->mu_adj > 0 ? from_fixed((u128)c->mu_adj) : 0.0L));
    const uint64_t nin = ninput < need ? ninput : need;
    int r = ensure_staging(c, (size_t)nin * c->nstreams, (size_t)noutput * c->nstreams);
    if (r) return r;
    for (uint32_t s = 0; s < c->nstreams; ++s)
        RS_TRY(hipMemcpyAsync(c->s_in + (size_t)s * nin * 2, in_ri + (size_t)s * in_stride * 2, (size_t)nin * 8,
                              hipMemcpyHostToDevice, c->stream));
    const int64_t n = process_device_locked(c, c->s_in, nin, nin, c->s_out, noutput, noutput, consumed);
    if (n < 0) return n;
    for (uint32_t s = 0; s < c->nstreams && n > 0; ++s)
        RS_TRY(hipMemcpyAsync(out_ri + (size_t)s * out_stride * 2, c->s_out + (size_t)s * noutput * 2, (size_t)n * 8,
                              hipMemcpyDeviceToHost, c->stream));
    RS_TRY(hipStreamSynchronize(c->stream));
    return n;
}

int64_t baz_resamp_process2_device(baz_resamp_ctx* c, const void* d_in, uint64_t in_stride, uint64_t ninput,
                                   const void* d_ratio, void* d_out, uint64_t out_stride, uint32_t noutput,
                                   uint64_t* consumed)
{
    if (!c || !d_in || !d_ratio || !d_out || in_stride < ninput || out_stride < noutput) return BAZ_RESAMP_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    return process2_device_locked(c, d_in, in_stride, ninput, d_ratio, d_out, out_stride, noutput, consumed);
}

int64_t baz_resamp_process2(baz_resamp_ctx* c, const float* in_ri, uint64_t in_stride, uint64_t ninput,
                            const float* ratio, float* out_ri, uint64_t out_stride, uint32_t noutput, uint64_t* consumed)
{
    if (!c || !in_ri || !ratio || !out_ri || in_stride < ninput || out_stride < noutput) return BAZ_RESAMP_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    int r = ensure_staging(c, (size_t)ninput * c->nstreams, (size_t)noutput * c->nstreams);
    if (r) return r;
    if (ninput > c->s_rr_cap) {
        if (c->s_rr) (void)hipFree(c->s_rr);
        c->s_rr = nullptr; c->s_rr_cap = 0;
        RS_TRY(hipMalloc((void**)&c->s_rr, (size_t)ninput * 4));
        c->s_rr_cap = ninput;
    }
    for (uint32_t s = 0; s < c->nstreams; ++s)
        RS_TRY(hipMemcpyAsync(c->s_in + (size_t)s * ninput * 2, in_ri + (size_t)s * in_stride * 2, (size_t)ninput * 8,
                              hipMemcpyHostToDevice, c->stream));
    RS_TRY(hipMemcpyAsync(c->s_rr, ratio, (size_t)ninput * 4, hipMemcpyHostToDevice, c->stream));
    const int64_t n = process2_device_locked(c, c->s_in, ninput, ninput, c->s_rr, c->s_out, noutput, noutput, consumed);
    if (n < 0) return n;
    for (uint32_t s = 0; s < c->nstreams && n > 0; ++s)
        RS_TRY(hipMemcpyAsync(out_ri + (size_t)s * out_stride * 2, c->s_out + (size_t)s * noutput * 2, (size_t)n * 8,
                              hipMemcpyDeviceToHost, c->stream));
    RS_TRY(hipStreamSynchronize(c->stream));
    return n;
}

int baz_resamp_set_mu(baz_resamp_ctx* c, double mu)
{
    if (!c || !(mu >= 0)) return BAZ_RESAMP_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    c->mu_update = to_fixed((long double)mu, &c->exact);
    c->update_mu = true;
    c->stopped_at_bad = false;       // a state change: the next two-input call delivers its output
    return BAZ_RESAMP_OK;
}

static int set_ratio_ld(baz_resamp_ctx* c, long double r)
{
    if (!(r > 0)) return BAZ_RESAMP_E_INVALID;
    if (!ratio_ok(r)) return BAZ_RESAMP_E_UNSUPPORTED;
    std::lock_guard<std::mutex> lk(c->mtx);
    c->mu_inc_update = to_fixed(r, &c->exact);
    c->update = true;
    c->stopped_at_bad = false;
    return BAZ_RESAMP_OK;
}

int baz_resamp_set_ratio(baz_resamp_ctx* c, double r) { return c ? set_ratio_ld(c, (long double)r) : BAZ_RESAMP_E_INVALID; }

int baz_resamp_set_ratio_rational(baz_resamp_ctx* c, uint64_t num, uint64_t denom)
{
    if (!c) return BAZ_RESAMP_E_INVALID;
    if (denom == 0) return BAZ_RESAMP_OK;                                                // ignored, .cc:249
    return set_ratio_ld(c, (long double)num / (long double)denom);
}

int baz_resamp_set_ratio_ppb(baz_resamp_ctx* c, long whole, double frac)
{
    if (!c) return BAZ_RESAMP_E_INVALID;
    return set_ratio_ld(c, ((long double)whole + (long double)frac) / (long double)1e9);  // .cc:122-123
}

int baz_resamp_adjust(baz_resamp_ctx* c, double d)
{
    if (!c || !std::isfinite(d)) return BAZ_RESAMP_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    c->mu_adj = to_fixed_signed((long double)d * from_fixed(c->mu_inc), &c->exact);      // .cc:132
    c->update_mu_adj = true;
    c->stopped_at_bad = false;
    return BAZ_RESAMP_OK;
}

double baz_resamp_mu(const baz_resamp_ctx* c) { return c ? (double)from_fixed(c->mu) : 0.0; }
double baz_resamp_ratio(const baz_resamp_ctx* c) { return c ? (double)from_fixed(c->mu_inc) : 0.0; }
int baz_resamp_phase_exact(const baz_resamp_ctx* c) { return c && c->exact ? 1 : 0; }
const float* baz_resamp_taps(const baz_resamp_ctx* c) { return c ? c->taps : nullptr; }

void baz_resamp_default_taps(float* out)
{
    if (out) build_taps(out);
}

int baz_resamp_set_taps(baz_resamp_ctx* c, const float* taps)
{
    if (!c || !taps) return BAZ_RESAMP_E_INVALID;
    for (int i = 0; i < (RS_NSTEPS + 1) * RS_NTAPS; ++i)
        if (!std::isfinite(taps[i])) return BAZ_RESAMP_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    RS_TRY(hipStreamSynchronize(c->stream));        // no launch in flight reads the old table
    std::memcpy(c->taps, taps, sizeof(c->taps));
    c->stopped_at_bad = false;
    RS_TRY(hipMemcpy(c->d_taps, c->taps, sizeof(c->taps), hipMemcpyHostToDevice));
    return BAZ_RESAMP_OK;
}

int baz_resamp_set_stream(baz_resamp_ctx* c, void* hip_stream)
{
    if (!c) return BAZ_RESAMP_E_INVALID;
    std::lock_guard<std::mutex> lk(c->mtx);
    DeviceGuard guard(c->device);
    RS_TRY(hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    return BAZ_RESAMP_OK;
}

int baz_resamp_sync(baz_resamp_ctx* c)
{
    if (!c) return BAZ_RESAMP_E_INVALID;
    DeviceGuard guard(c->device);
    RS_TRY(hipStreamSynchronize(c->stream));
    return BAZ_RESAMP_OK;
}

const char* baz_resamp_strerror(int code)
{
    switch (code) {
        case BAZ_RESAMP_OK: return "ok";
        case BAZ_RESAMP_E_INVALID: return "invalid argument";
        case BAZ_RESAMP_E_NOMEM: return "out of memory";
        case BAZ_RESAMP_E_HIP: return "HIP runtime error";
        case BAZ_RESAMP_E_UNSUPPORTED: return "unsupported configuration";
        case BAZ_RESAMP_E_NODEVICE: return "no gfx950 device";
        default: return "unknown error";
    }
}

}  // extern "C"
