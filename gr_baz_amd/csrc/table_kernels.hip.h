// table_kernels.hip.h -- the steering table's device images, built ON the device (round 5).
//
// set_array_response (/root/reference/lib/baz_music_doa.cc:60-70) takes the lock (:67) and copies one vector (:69): a
// 2.3 MB copy at config 3.  Until round 4 the replacement built every image of the table -- F / FB / TB / the coarse f16
// pieces / the int8 digit planes / ||a||^2 -- on ONE host thread and uploaded them with synchronous copies while it held the
// context mutex: 0.3 - 0.5 s of stalled work() per retune at config 3 (missing 2).  Here every image is a pure
// function of the raw fp32 table and of three scalars (max|F|, max||a||^2, "finite"), evaluated per output element by the
// kernels below on a side stream into a SHADOW set of buffers; the context mutex is held for the pointer swap only
// (baz_music_hip.hip: build_tables_device / baz_music_set_table).
//
// Bit-exactness against the former host builders (kept in baz_music_hip.hip as baz_music_debug_host_table_image, the
// checker of tests/test_retune.py): every value is either
//   * a sum / difference of two EXACT products of widened fp32 values (48-bit significands fit fp64), i.e. one rounding,
//     the same with or without FMA contraction;
//   * a power-of-two scaling, a round-to-nearest-even conversion (fp64 -> fp32, fp64 -> int64) or the integer digit cuts;
//   * the f16 pieces through the SAME integer routine on both sides (f16_bits / f16_value below are __host__ __device__);
//   * ||a||^2 accumulated antenna by antenna in the host loop's order (an add of an exact-product sum: no fusable multiply).
// gfx950 only.
#pragma once

#include "scan_i8_kernels.hip.h"      // (scan_coarse_kernels.hip.h, music_kernels.hip.h: layouts and their constexpr helpers)
#include "scan_coarse_kernels.hip.h"
#ifdef BAZ_MUSIC_LAB
#include "scan_i8p_kernels.hip.h"
#endif

namespace baztab {

using namespace bazmusic;

// float -> IEEE binary16 bits, round to nearest even (host and device: one routine, so the images agree by construction)
__host__ __device__ inline uint16_t f16_bits(float f)
{
    const uint32_t x = __builtin_bit_cast(uint32_t, f);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const int32_t ex = (int32_t)((x >> 23) & 0xFF) - 127 + 15;
    uint32_t man = x & 0x7FFFFFu;
    if (((x >> 23) & 0xFF) == 0xFF) return (uint16_t)(sign | 0x7C00u | (man ? 0x200u : 0u));
    if (ex >= 31) return (uint16_t)(sign | 0x7C00u);
    if (ex <= 0) {                      // subnormal or zero
        if (ex < -10) return (uint16_t)sign;
        man |= 0x800000u;
        const int shift = 14 - ex;      // 24-bit significand -> 10-bit field of a subnormal
        uint32_t h = man >> shift;
        const uint32_t rem = man & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1u))) ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((uint32_t)ex << 10) | (man >> 13);
    const uint32_t rem = man & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;   // may carry into the exponent: still right
    return (uint16_t)(sign | h);
}

// binary16 bits -> float, exact (finite inputs)
__host__ __device__ inline float f16_value(uint16_t h)
{
    const uint32_t ex = (h >> 10) & 0x1Fu, man = h & 0x3FFu;
    float v;
    if (ex) v = __builtin_bit_cast(float, ((ex + 112u) << 23) | (man << 13));
    else v = (float)man * 0x1p-24f;
    return (h & 0x8000u) ? -v : v;
}

// F[bin][e], e = r m + c (build_F): r == c  |a_r|^2;  r < c  Re(conj(a_r) a_c);  r > c  Im(conj(a_c) a_r) -- the real
// bilinear-form table of a^H Q a.  `a` = the bin's m complex fp32 entries; widened exactly, products exact, one rounding.
__host__ __device__ inline double tab_F(const float* __restrict__ a, uint32_t m, uint32_t e)
{
#pragma clang fp contract(off)       // (both products are exact -- 24-bit factors --, so a contraction could not change the one rounding; stated anyway)
    const uint32_t r = e / m, c = e - r * m;
    if (r == c) {
        const double re = a[2 * r], im = a[2 * r + 1];
        return re * re + im * im;
    }
    if (r < c) {                 // i = r, j = c: air ajr + aii aji
        const double air = a[2 * r], aii = a[2 * r + 1], ajr = a[2 * c], aji = a[2 * c + 1];
        return air * ajr + aii * aji;
    }
    // i = c, j = r: air aji - aii ajr
    const double air = a[2 * c], aii = a[2 * c + 1], ajr = a[2 * r], aji = a[2 * r + 1];
    return air * aji - aii * ajr;
}

// ||a||^2 of one bin, antenna by antenna (the host loops' order)
__host__ __device__ inline double tab_a2(const float* __restrict__ a, uint32_t m)
{
#pragma clang fp contract(off)       // v + (re^2 + im^2) as written on both sides: never fma(re, re, fma(im, im, v))
    double v = 0.0;
    for (uint32_t i = 0; i < m; ++i) {
        const double re = a[2 * i], im = a[2 * i + 1];
        const double t = re * re + im * im;
        v = v + t;
    }
    return v;
}

// the three scalars every image's parameters follow from
// 8-byte words from one buffer to another, either of which may be page-locked HOST memory addressed over the link: the retune path moves its
// 2.3 MB table and its 24-byte statistic with this kernel instead of hipMemcpyAsync -- once per process the runtime spent 6.6 ms INSIDE that call
// (a copy-engine queue set up on first contention with another thread's copies), which a stream that must not stall cannot afford.
__global__ __launch_bounds__(256) void copy_words_kernel(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

struct TableStats {
    unsigned long long fmax_bits;    // bits of max |F| over the finite entries (>= 0: the bits order like the values)
    unsigned long long amax2_bits;   // bits of max ||a||^2 over the bins with ||a||^2 < 1e300
    unsigned int nonfinite;          // some F entry is not finite
    unsigned int pad;
};

// One thread per bin.  need_f = 0 (the run-time-m path keeps no bilinear-form table): only ||a||^2.
__global__ __launch_bounds__(256) void table_stats_kernel(const float* __restrict__ tab, uint32_t m, uint32_t res, int need_f,
                                                          TableStats* __restrict__ st)
{
    const uint32_t bin = blockIdx.x * 256u + threadIdx.x;
    double fmax = 0.0, amax2 = 0.0;
    unsigned int bad = 0;
    if (bin < res) {
        const float* a = tab + (size_t)bin * m * 2;
        if (need_f) {
            const uint32_t mm = m * m;
            for (uint32_t e = 0; e < mm; ++e) {
                const double v = fabs(tab_F(a, m, e));
                if (!(v <= 1.79769313486231570815e308)) bad = 1;      // inf or NaN
                else fmax = (v > fmax) ? v : fmax;
            }
        }
        const double a2 = tab_a2(a, m);
        if (a2 < 1e300) amax2 = a2;                                    // (false for NaN)
    }
#pragma unroll
    for (int msk = 1; msk < 64; msk <<= 1) {
        const double of = __shfl_xor(fmax, msk, 64), oa = __shfl_xor(amax2, msk, 64);
        fmax = (of > fmax) ? of : fmax;
        amax2 = (oa > amax2) ? oa : amax2;
        bad |= (unsigned int)__shfl_xor((int)bad, msk, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (fmax > 0.0) atomicMax(&st->fmax_bits, (unsigned long long)__builtin_bit_cast(uint64_t, fmax));
        if (amax2 > 0.0) atomicMax(&st->amax2_bits, (unsigned long long)__builtin_bit_cast(uint64_t, amax2));
        if (bad) atomicOr(&st->nonfinite, 1u);
    }
}

// FB (build_FB): FB[(sti, c2 = 2 s + (t >> 1), lane)].{x, y} = F[bin = 64 (sti - 1) + 4 c + t][e = 4 s + g], t = 2 (c2 & 1) + {0, 1};
// outside the table a huge diagonal.  One thread per double2.
__global__ __launch_bounds__(256) void build_fb_kernel(const float* __restrict__ tab, uint32_t m, uint32_t res, uint32_t steps,
                                                       double2* __restrict__ FB)
{
    const uint32_t mm = m * m, ks = (mm + 3) / 4;
    const size_t idx = (size_t)blockIdx.x * 256u + threadIdx.x;
    const size_t total = (size_t)(steps + 2) * 2 * ks * 64;
    if (idx >= total) return;
    const uint32_t lane = (uint32_t)(idx & 63u), g = lane >> 4, c = lane & 15u;
    const size_t rest = idx >> 6;
    const uint32_t c2 = (uint32_t)(rest % (2 * ks)), sti = (uint32_t)(rest / (2 * ks));
    const uint32_t s = c2 >> 1, e = 4 * s + g;
    double v[2] = {0.0, 0.0};
    if (e < mm) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t t = 2 * (c2 & 1u) + (uint32_t)k;
            const long long bin = 64ll * ((long long)sti - 1) + 4 * c + t;
            if (bin >= 0 && bin < (long long)res) v[k] = tab_F(tab + (size_t)bin * m * 2, m, e);
            else v[k] = ((e / m) == (e % m)) ? 1e300 : 0.0;
        }
    }
    FB[idx] = make_double2(v[0], v[1]);
}

// TB (build_TB): the raw table widened, K dimension = the 2m real coordinates; zero outside the table
__global__ __launch_bounds__(256) void build_tb_kernel(const float* __restrict__ tab, uint32_t m, uint32_t res, uint32_t steps,
                                                       double2* __restrict__ TB)
{
    const uint32_t ks2 = (2 * m + 3) / 4;
    const size_t idx = (size_t)blockIdx.x * 256u + threadIdx.x;
    const size_t total = (size_t)(steps + 2) * 2 * ks2 * 64;
    if (idx >= total) return;
    const uint32_t lane = (uint32_t)(idx & 63u), g = lane >> 4, c = lane & 15u;
    const size_t rest = idx >> 6;
    const uint32_t c2 = (uint32_t)(rest % (2 * ks2)), sti = (uint32_t)(rest / (2 * ks2));
    const uint32_t s = c2 >> 1, e = 4 * s + g;
    double v[2] = {0.0, 0.0};
    if (e < 2 * m) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t t = 2 * (c2 & 1u) + (uint32_t)k;
            const long long bin = 64ll * ((long long)sti - 1) + 4 * c + t;
            if (bin >= 0 && bin < (long long)res) v[k] = (double)tab[((size_t)bin * m + (e >> 1)) * 2 + (e & 1u)];
        }
    }
    TB[idx] = make_double2(v[0], v[1]);
}

// ||a||^2 per bin, padded by one 64-bin step on either side with a huge value (the scan's short form), or plain [res]
__global__ __launch_bounds__(256) void build_a2_kernel(const float* __restrict__ tab, uint32_t m, uint32_t res, uint32_t count,
                                                       uint32_t front_pad, double* __restrict__ out)
{
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= count) return;
    const long long b = (long long)idx - (long long)front_pad;
    out[idx] = (b >= 0 && b < (long long)res) ? tab_a2(tab + (size_t)b * m * 2, m) : 1e300;
}

// the run-time-m path's transposed table: TA[i][bin] = table[bin][i]
__global__ __launch_bounds__(256) void build_ta_kernel(const float2* __restrict__ tab, uint32_t m, uint32_t res, float2* __restrict__ TA)
{
    const size_t idx = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (idx >= (size_t)m * res) return;
    const uint32_t i = (uint32_t)(idx / res), b = (uint32_t)(idx - (size_t)i * res);
    TA[idx] = tab[(size_t)b * m + i];
}

// Coarse image, C part (build_coarse_image): per 16-bin tile t <= tiles the f16 pieces Fh, Fl of Fs = F FS.  One thread per
// (t, c, e < m^2); everything else of the image is zero (the caller clears it first).
__global__ __launch_bounds__(256) void build_coarse_c_kernel(const float* __restrict__ tab, uint32_t m, uint32_t res, uint32_t tiles,
                                                             double FS, uint8_t* __restrict__ img)
{
    const uint32_t mm = m * m;
    const size_t idx = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (idx >= (size_t)(tiles + 1) * 16 * mm) return;
    const uint32_t e = (uint32_t)(idx % mm);
    const uint32_t c = (uint32_t)((idx / mm) & 15u), t = (uint32_t)(idx / ((size_t)mm * 16));
    const bool wide = m > 4;
    const size_t cbytes = (size_t)cs_c_units((int)m) * 16;
    uint8_t* T = img + (size_t)t * cbytes;
    uint16_t* fh = reinterpret_cast<uint16_t*>(T + (wide ? (size_t)(e >> 5) * 2048 : 0));
    uint16_t* fl = reinterpret_cast<uint16_t*>(T + (wide ? (size_t)(e >> 5) * 2048 + 1024 : 512));
    const uint32_t bin = 16 * t + c;
    uint16_t hi = 0, lo = 0;
    if (bin >= res) hi = ((e / m) == (e % m)) ? f16_bits(32768.0f) : (uint16_t)0;
    else {
        const float fs = (float)(tab_F(tab + (size_t)bin * m * 2, m, e) * FS);
        hi = f16_bits(fs);
        lo = f16_bits(fs - f16_value(hi));
    }
    const uint32_t gb = wide ? ((e >> 3) & 3u) : (e >> 3), j = e & 7u;
    fh[(gb * 16 + c) * 8 + j] = hi;
    fl[(gb * 16 + c) * 8 + j] = lo;
}

// Coarse image, X part: the fp64 B operand of a fired tile, X[(sidx >> 1) 128 + lane 2 + (sidx & 1)] = F[bin][e = 4 sidx + g]
__global__ __launch_bounds__(256) void build_coarse_x_kernel(const float* __restrict__ tab, uint32_t m, uint32_t res, uint32_t tiles,
                                                             double* __restrict__ X0)
{
    const uint32_t mm = m * m, ng = (uint32_t)cs_groups((int)m);
    const size_t per_tile = (size_t)256 * ng;                       // doubles
    const size_t idx = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (idx >= (size_t)tiles * per_tile) return;
    const uint32_t t = (uint32_t)(idx / per_tile), w = (uint32_t)(idx - (size_t)t * per_tile);
    const uint32_t k = w & 1u, lane = (w >> 1) & 63u, hp = w >> 7;  // w = hp 128 + lane 2 + k, sidx = 2 hp + k
    const uint32_t sidx = 2 * hp + k, g = lane >> 4, c = lane & 15u, e = 4 * sidx + g;
    const uint32_t bin = 16 * t + c;
    double v = 0.0;
    if (e < mm) v = (bin < res) ? tab_F(tab + (size_t)bin * m * 2, m, e) : (((e / m) == (e % m)) ? 1e300 : 0.0);
    X0[idx] = v;
}

// int8 digit image (build_i8_image): one thread per (bin, kb, g) = 16 terms x 7 digits, one 16-B store per digit plane.
// Bins beyond the table inside the last step and terms e >= m^2 stay zero (the caller clears the image first).
__global__ __launch_bounds__(256) void build_i8_kernel(const float* __restrict__ tab, uint32_t m, uint32_t res, uint32_t steps, double sf,
                                                       uint8_t* __restrict__ img, uint8_t* __restrict__ img2)
{
    constexpr int NS = I8_NS, ND = I8_ND;
    const uint32_t mm = m * m, nkb = (uint32_t)i8_nkb((int)m);
    const size_t idx = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (idx >= (size_t)res * nkb * 4) return;
    const uint32_t g = (uint32_t)(idx & 3u), kb = (uint32_t)((idx >> 2) % nkb), bin = (uint32_t)((idx >> 2) / nkb);
    const uint32_t st = bin >> 6, w = bin & 63u, c = w >> 2, t = w & 3u;
    const float* a = tab + (size_t)bin * m * 2;
    uint32_t dig[ND][4];
#pragma unroll
    for (int s = 0; s < ND; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) dig[s][q] = 0u;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint32_t e = 64 * kb + 16 * g + (uint32_t)j;
        if (e < mm) {
            long long v = __double2ll_rn(tab_F(a, m, e) * sf);          // |v| <= 2^54 (1 + 2^-10); the product is exact
#pragma unroll
            for (int s = ND - 1; s >= 1; --s) {
                const long long h = (v + 128) >> 8;                      // floor((v + 128) / 256)
                dig[s][j >> 2] |= (uint32_t)((v - h * 256) & 255) << (8 * (j & 3));
                v = h;
            }
            dig[0][j >> 2] |= (uint32_t)(v & 255) << (8 * (j & 3));
        }
    }
    const size_t tile = (size_t)(st * 4 + t) * nkb + kb, in_lane = (size_t)(g * 16 + c) * 16;
#pragma unroll
    for (int s = 0; s < ND; ++s) {
        uint8_t* dst = (s >= NS) ? img2 + (tile * (ND - NS) + (size_t)(s - NS)) * 1024 + in_lane
                                 : img + (tile * NS + (size_t)s) * 1024 + in_lane;
        *reinterpret_cast<uint4*>(dst) = make_uint4(dig[s][0], dig[s][1], dig[s][2], dig[s][3]);
    }
}

#ifdef BAZ_MUSIC_LAB
// Level-packed int8 operands for 2 .. 4 antennas (scan_i8p_kernels.hip.h; lab builds only): one thread per bin = 16 terms x 7 digits; digit s of the
// terms goes to slot s of B (s <= 3) or slot s - 4 of B' (the image's second half), 16 B each.  Padded steps, bins outside the table,
// terms e >= m^2 and B' slot 3 stay zero (the caller clears the image first).
__global__ __launch_bounds__(256) void build_i8p_kernel(const float* __restrict__ tab, uint32_t m, uint32_t res, uint32_t steps, double sf,
                                                        uint4* __restrict__ img)
{
    constexpr int ND = I8_ND;
    const uint32_t mm = m * m;
    const uint32_t bin = blockIdx.x * 256u + threadIdx.x;
    if (bin >= res) return;
    const uint32_t st = bin >> 6, w = bin & 63u, c = w >> 2, t = w & 3u;
    const float* a = tab + (size_t)bin * m * 2;
    uint32_t dig[ND][4];
#pragma unroll
    for (int s = 0; s < ND; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) dig[s][q] = 0u;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if ((uint32_t)j < mm) {
            long long v = __double2ll_rn(tab_F(a, m, (uint32_t)j) * sf);
#pragma unroll
            for (int s = ND - 1; s >= 1; --s) {
                const long long h = (v + 128) >> 8;                      // floor((v + 128) / 256)
                dig[s][j >> 2] |= (uint32_t)((v - h * 256) & 255) << (8 * (j & 3));
                v = h;
            }
            dig[0][j >> 2] |= (uint32_t)(v & 255) << (8 * (j & 3));
        }
    }
    uint4* B1 = img + ((size_t)(st + 1) * 4 + t) * 64 + c;
    uint4* B2 = B1 + i8p_operand_units(steps);
#pragma unroll
    for (int s = 0; s < ND; ++s) {
        uint4* dst = (s < 4) ? B1 + s * 16 : B2 + (s - 4) * 16;
        *dst = make_uint4(dig[s][0], dig[s][1], dig[s][2], dig[s][3]);
    }
}
#endif   // BAZ_MUSIC_LAB

}  // namespace baztab
