// resamp_kernels.hip.h -- gfx950 kernel for gr-baz's fractional resampler (config 5 front-end).
//
// Replaces the one-input loop of fractional_resampler_cc_impl::general_work
// (/root/reference/lib/baz_fractional_resampler_cc.cc:162-203):
//     out[oo++] = d_resamp->interpolate(&in[ii], d_mu);            (.cc:172; gnuradio-filter MMSE interpolator)
//     s = d_mu + d_mu_inc (+ d_mu_adj once); f = floor(s); ii += (int)f; d_mu = s - f;        (.cc:183-193)
// The phase recurrence is a sum, so output o is evaluated independently: P_o = base + o * inc in 64.64-bit fixed
// point (ii_o = integer part, mu_o = fraction), exactly the reference's x87 sequence whenever that one is exact
// (include/baz_resamp_hip.h).  One thread per (output, stream); a block's input window is contiguous
// (256 outputs span <= 256*ratio + 8 samples) and is read through L1/L2 -- 8 overlapping float2 loads per output,
// HBM sees every input once.  The 129 x 8 tap table (4 KiB) sits in LDS.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bazresamp {

constexpr int RS_NTAPS = 8;
constexpr int RS_NSTEPS = 128;
constexpr int RS_BLOCK = 256;
constexpr int RS_PER_THREAD = 4;

struct PhaseParams {
    uint64_t first_lo, first_hi;   // P_0 = mu of the first output (64.64): hi = integer part, lo = fraction
    uint64_t base_lo, base_hi;     // P_o = base + (o - 1) * inc for o >= 1 (base = P_0 + inc + adjustment)
    uint64_t inc_lo, inc_hi;
};

__device__ __forceinline__ void phase_of(const PhaseParams& p, uint32_t o, uint64_t& ipart, uint64_t& frac)
{
    if (o == 0) { ipart = p.first_hi; frac = p.first_lo; return; }
    const uint64_t j = (uint64_t)o - 1u;
    const uint64_t plo = j * p.inc_lo;                        // low 64 bits of j * inc_lo
    const uint64_t phi = __umul64hi(j, p.inc_lo);             // its carry into the integer part
    const uint64_t lo = p.base_lo + plo;
    const uint64_t carry = lo < plo ? 1u : 0u;
    frac = lo;
    ipart = p.base_hi + j * p.inc_hi + phi + carry;
}

__global__ __launch_bounds__(RS_BLOCK) void resamp_kernel(const float2* __restrict__ in, uint64_t in_stride,
                                                           float2* __restrict__ out, uint64_t out_stride,
                                                           uint32_t noutput, PhaseParams p,
                                                           const float* __restrict__ taps)
{
    __shared__ float st[(RS_NSTEPS + 1) * RS_NTAPS];
    for (int i = threadIdx.x; i < (RS_NSTEPS + 1) * RS_NTAPS; i += RS_BLOCK) st[i] = taps[i];
    __syncthreads();
    const uint32_t stream = blockIdx.y;
    const float2* __restrict__ xs = in + (size_t)stream * in_stride;
    float2* __restrict__ ys = out + (size_t)stream * out_stride;
    // RS_PER_THREAD outputs per thread, RS_BLOCK apart (coalesced stores), so the 4 KiB tap table is staged once per
    // 1,024 outputs instead of once per 256 (it was as much traffic as the samples themselves)
#pragma unroll
    for (int it = 0; it < RS_PER_THREAD; ++it) {
        const uint32_t o = (blockIdx.x * RS_PER_THREAD + it) * RS_BLOCK + threadIdx.x;
        if (o >= noutput) break;
        uint64_t ii, frac;
        phase_of(p, o, ii, frac);
        // (float)d_mu: round-to-nearest-even of the 64-bit fraction, like the x87 -> float conversion at .cc:172
        const float mu = __ull2float_rn(frac) * 5.42101086242752217e-20f;   // * 2^-64 (exact)
        const int imu = __float2int_rn(mu * (float)RS_NSTEPS);              // rint(mu * NSTEPS), ties to even
        const float* t = st + imu * RS_NTAPS;
        const float2* x = xs + ii;
        float re = 0.0f, im = 0.0f;
        {
#pragma clang fp contract(off)   // float multiply, then float add, like the reference's FIR kernel (no fma)
#pragma unroll
            for (int k = 0; k < RS_NTAPS; ++k) {
                const float2 v = x[k];
                const float w = t[RS_NTAPS - 1 - k];                        // the FIR kernel stores its taps reversed
                re = re + v.x * w;
                im = im + v.y * w;
            }
        }
        ys[o] = make_float2(re, im);
    }
}

// -------------------------------------------------------------------------------------------------------------
// Two-input branch of general_work (.cc:205-217): the ratio is read PER SAMPLE from a second input,
//     out[oo++] = interpolate(&in[ii], d_mu);  d_mu_inc = rr[ii];  s = d_mu + d_mu_inc;  ii += floor(s);  d_mu = s - floor(s)
// so ii_{o+1} depends on the ratio sample found at ii_o: a data-dependent serial chain, no closed form.  It is offered
// so that flowgraphs which wire the port connect and produce the reference's stream, not for throughput: ONE lane walks
// the phase chain over an LDS window of the ratio input (the other lanes of the workgroup only fetch that window) and
// records (ii_o, imu_o) per output; the interpolation itself then runs in parallel from that table
// (resamp_table_kernel).  A float ratio >= 2^-41 (24-bit significand: its last bit is >= 2^-64) and a phase with <= 64
// fractional bits add exactly in 64.64 fixed point; that equals the reference's x87 sequence whenever the x87 sum
// s = mu + inc is itself exact in its 64-bit significand -- always for ratios below 1, and for larger ones as long as
// mu + inc needs no more than 64 significant bits (a ratio >= 1 pushes mu's last bits out: the two then differ by
// < 2^-63 in mu, which moves imu only when mu * 128 sits that close to a rounding boundary).  Smaller ratios
// (0 < r < 2^-41) are truncated to 2^-64 (the x87 sum rounds them instead).  The walk stops early -- fewer outputs, like a
// short input -- at a ratio sample that is not a finite number in (0, 2^31]: zero, negative, NaN, inf (the reference
// would stand still, walk backwards or index with garbage there); a call that STARTS on such a sample is an error
// (baz_resamp_process2: BAZ_RESAMP_E_INVALID), not an endless one-output / zero-consumed loop.
// -------------------------------------------------------------------------------------------------------------
struct WalkResult {
    uint64_t n;          // outputs produced
    uint64_t ii;         // input index after the last step = consume_each()
    uint64_t frac;       // d_mu after the last step (0.64 fixed point)
    uint32_t last_bits;  // bit pattern of the last ratio sample applied (d_mu_inc), 0 if none
    uint32_t status;     // 0 ran to noutput / end of input, 1 stopped at an unusable ratio sample
};

constexpr int RS_WALK_WIN = 8192;

__global__ __launch_bounds__(256) void resamp_walk_kernel(const float* __restrict__ rr, uint64_t ninput, uint32_t noutput,
                                                          uint64_t frac0, uint32_t* __restrict__ ii_out,
                                                          uint32_t* __restrict__ imu_out, WalkResult* __restrict__ res)
{
    __shared__ float win[RS_WALK_WIN];
    __shared__ uint64_t s_ii, s_frac;
    __shared__ uint32_t s_o, s_stop, s_last, s_status;
    if (threadIdx.x == 0) { s_ii = 0; s_frac = frac0; s_o = 0; s_stop = 0; s_last = 0; s_status = 0; }
    __syncthreads();
    while (true) {
        const uint64_t w0 = s_ii;
        if (s_stop || s_o >= noutput || w0 + (RS_NTAPS - 1) >= ninput) break;
        const uint64_t avail = ninput - w0;
        const uint32_t cnt = avail < (uint64_t)RS_WALK_WIN ? (uint32_t)avail : (uint32_t)RS_WALK_WIN;
        for (uint32_t i = threadIdx.x; i < cnt; i += 256) win[i] = rr[w0 + i];
        __syncthreads();
        if (threadIdx.x == 0) {
            uint64_t ii = w0, frac = s_frac;
            uint32_t o = s_o, last = s_last;
            while (o < noutput && ii < w0 + cnt && ii + (RS_NTAPS - 1) < ninput) {
                const float mu = __ull2float_rn(frac) * 5.42101086242752217e-20f;     // (float)d_mu, .cc:206
                ii_out[o] = (uint32_t)ii;
                imu_out[o] = (uint32_t)__float2int_rn(mu * (float)RS_NSTEPS);
                ++o;
                const float r = win[ii - w0];                                          // d_mu_inc = rr[ii], .cc:207
                if (!(r > 0.0f && r <= 2147483648.0f)) { s_stop = 1; s_status = 1; break; }
                last = __float_as_uint(r);
                const double rd = (double)r;
                const uint64_t ip = (uint64_t)rd;
                const uint64_t fr = (uint64_t)((rd - (double)ip) * 18446744073709551616.0);   // exact from 2^-41 up (<= 24 significant bits), else truncated
                const uint64_t nf = frac + fr;
                ii += ip + (nf < frac ? 1u : 0u);                                      // s = mu + inc; ii += floor(s), .cc:209-213
                frac = nf;                                                             // d_mu = s - floor(s)
            }
            s_ii = ii; s_frac = frac; s_o = o; s_last = last;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        res->n = s_o; res->ii = s_ii; res->frac = s_frac; res->last_bits = s_last; res->status = s_status;
    }
}

__global__ __launch_bounds__(RS_BLOCK) void resamp_table_kernel(const float2* __restrict__ in, uint64_t in_stride,
                                                                 float2* __restrict__ out, uint64_t out_stride,
                                                                 const uint32_t* __restrict__ ii_in,
                                                                 const uint32_t* __restrict__ imu_in,
                                                                 const WalkResult* __restrict__ res,
                                                                 const float* __restrict__ taps)
{
    __shared__ float st[(RS_NSTEPS + 1) * RS_NTAPS];
    for (int i = threadIdx.x; i < (RS_NSTEPS + 1) * RS_NTAPS; i += RS_BLOCK) st[i] = taps[i];
    __syncthreads();
    const uint32_t o = blockIdx.x * RS_BLOCK + threadIdx.x;
    if ((uint64_t)o >= res->n) return;
    const float2* __restrict__ x = in + (size_t)blockIdx.y * in_stride + ii_in[o];
    const float* t = st + imu_in[o] * RS_NTAPS;
    float re = 0.0f, im = 0.0f;
    {
#pragma clang fp contract(off)   // float multiply, then float add, like the reference's FIR kernel (no fma)
#pragma unroll
        for (int k = 0; k < RS_NTAPS; ++k) {
            const float2 v = x[k];
            const float w = t[RS_NTAPS - 1 - k];
            re = re + v.x * w;
            im = im + v.y * w;
        }
    }
    out[(size_t)blockIdx.y * out_stride + o] = make_float2(re, im);
}

}  // namespace bazresamp
