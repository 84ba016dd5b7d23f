// agc_kernels.hip.h -- gfx950 kernels for gr-baz's AGC block (config 5 front-end).
//
// Replaces the live part of baz_agc_cc::work (/root/reference/lib/baz_agc_cc.cc:64-102):
//     mag  = sqrt(re^2 + im^2)                                   (double, .cc:74-77)
//     env  = count == 0 ? mag : env*(1.0 - rate) + mag*rate      (.cc:79-82)
//     gain = reference / env                                     (.cc:89)
//     out  = (float)(re*gain), (float)(im*gain)                  (.cc:97-100);  env[i], mul[i] optional ports
// The envelope is a first-order linear recurrence e_i = a e_{i-1} + b m_i (a = 1 - rate, b = rate, fp64): its
// composition is associative ((A2,S2) o (A1,S1) = (A1 A2, A2 S1 + S2)), so it is evaluated as a three-pass
// tiled scan instead of the reference's one-sample-at-a-time loop:
//   agc_tile_kernel<0>  per 256-sample tile (one wave): the pair (A_c, S_c) that maps the carry-in to the carry-out
//   agc_carry_kernel    per stream: carry-in of every tile (incl. the count == 0 rule: e_{-1} := |x_0|)
//   agc_tile_kernel<2>  per tile: recompute the local recurrence from its carry-in, emit out / env / mul (planar), or
//   agc_tile_kernel<1>  the same with the S streams of a tile written interleaved as MUSIC items (config 5)
// Each lane runs 4 consecutive samples in registers (32 contiguous bytes per lane, 2 KiB per wave: coalesced without
// an LDS transpose).  Per sample the arithmetic is the reference's, operation for operation (agc_env_step & co.); only
// a tile's ENTRY state carries the scan's fp64 re-association (~1e-15 relative), see the note at agc_mag().
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bazagc {

struct AgcParams {
    double a;          // 1.0 - (double)rate
    double b;          // (double)rate
    double reference;
    // the fast path of full tiles (see agc_tile_kernel); pw == nullptr switches it off
    double c1, c2, c4, c8;     // (a^IE)^1, ^2, ^4, ^8
    double a_tile;             // a^(64 IE): the A of a full tile's map
    const double* pw;          // [64][4]: (a^IE)^((l & 15) + 1), (a^IE)^((l & 31) + 1), (a^IE)^l, unused
};

// The reference's per-sample expressions (.cc:74-100), operation for operation: every product and sum is rounded
// on its own (the host compiler does not fuse them), so the device code must not contract them into FMAs either --
// hipcc would (fp-contract=fast): these helpers switch contraction off.  With them a tile that starts from the
// sequential loop's state reproduces its float32 outputs BIT FOR BIT; what remains is the tile's entry state, which
// comes out of the re-associated scan a few ulp_f64 (~1e-15) away from the sequential value and flips the float32
// rounding of about one output sample in 1e8 by one ulp (tests/lab/agc_exact.py; an exact parallel evaluation does
// not exist: the rounded recurrence is not associative, and two states one ulp apart only coalesce after ~1/rate samples).
// (ROCm's __dadd_rn / __dmul_rn are plain operators and get contracted like any other: the pragma is what holds.)
__device__ __forceinline__ double agc_mag(const float2 x)
{
#pragma clang fp contract(off)
    const double d0 = x.x, d1 = x.y;
    const double p0 = d0 * d0, p1 = d1 * d1;
    return __dsqrt_rn(p0 + p1);                                                // .cc:74-77
}
__device__ __forceinline__ double agc_env_step(const double e, const double mag, const double a, const double b)
{
#pragma clang fp contract(off)
    const double p0 = e * a, p1 = mag * b;
    return p0 + p1;                                                            // .cc:82
}
__device__ __forceinline__ float2 agc_apply(const float2 x, const double gain)
{
#pragma clang fp contract(off)
    const double re = (double)x.x * gain, im = (double)x.y * gain;             // .cc:97-100
    return make_float2((float)re, (float)im);
}

// (A2,S2) o (A1,S1): apply 1 first, then 2
__device__ __forceinline__ void compose(double& A, double& S, const double A1, const double S1)
{
    // (A,S) := (A,S) o (A1,S1)
    S = fma(A, S1, S);
    A = A * A1;
}

// One workgroup (AGC_CARRY_THREADS lanes) per stream: carry-in of every chunk.  first != 0 means this call starts the
// stream (count == 0, .cc:79-80): the state "before sample 0" is |x_0| itself, which makes e_0 = |x_0| (a + b) = |x_0|.
// Thread t composes the maps of its contiguous slice of chunks, a wave scan + a scan over the wave totals give the
// state entering each slice, and the thread walks its slice again writing the per-chunk carry-ins.  (A single-lane
// loop took 21 % of the AGC time at 4,096 chunks per stream; one wave per stream still 3 % of the config-5 step at
// 16,384 tiles: the walk is a chain of dependent L2 loads.)
constexpr int AGC_CARRY_THREADS = 1024;

__global__ __launch_bounds__(AGC_CARRY_THREADS) void agc_carry_kernel(const float2* __restrict__ in, uint64_t stride,
                                                                       const double2* __restrict__ chunk_pair,
                                                                       double* __restrict__ carry_in, uint32_t nchunks,
                                                                       const double* __restrict__ env_state, int first)
{
    __shared__ double wA[AGC_CARRY_THREADS / 64], wS[AGC_CARRY_THREADS / 64];
    const uint32_t stream = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double e0;
    if (first) {
        e0 = agc_mag(in[(size_t)stream * stride]);
    } else {
        e0 = env_state[stream];
    }
    const double2* __restrict__ cp = chunk_pair + (size_t)stream * nchunks;
    double* __restrict__ ci = carry_in + (size_t)stream * nchunks;
    const uint32_t per = (nchunks + AGC_CARRY_THREADS - 1) / AGC_CARRY_THREADS;
    const uint32_t c0 = ((uint64_t)tid * per < nchunks) ? (uint32_t)tid * per : nchunks;
    const uint32_t c1 = (c0 + per < nchunks) ? (c0 + per) : nchunks;
    double A = 1.0, S = 0.0;
    for (uint32_t c = c0; c < c1; ++c) {
        const double2 p = cp[c];
        S = fma(p.x, S, p.y);
        A *= p.x;
    }
    double Ai = A, Si = S;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double Ap = __shfl_up(Ai, d, 64), Sp = __shfl_up(Si, d, 64);
        if (lane >= d) compose(Ai, Si, Ap, Sp);
    }
    if (lane == 63) { wA[wave] = Ai; wS[wave] = Si; }
    __syncthreads();
    double Ae = __shfl_up(Ai, 1, 64), Se = __shfl_up(Si, 1, 64);
    if (lane == 0) { Ae = 1.0; Se = 0.0; }
    double Aw = 1.0, Sw = 0.0;              // map of all earlier waves
    for (int w = 0; w < wave; ++w) { const double a2 = wA[w], s2 = wS[w]; Sw = fma(a2, Sw, s2); Aw *= a2; }
    compose(Ae, Se, Aw, Sw);                // thread-exclusive map of the whole block prefix
    double e = fma(Ae, e0, Se);             // state entering this thread's slice
    for (uint32_t c = c0; c < c1; ++c) {
        ci[c] = e;
        const double2 p = cp[c];
        e = fma(p.x, e, p.y);
    }
}

// -------------------------------------------------------------------------------------------------------------
// Interleaved-output form (config 5: the AGC sits right in front of MUSIC-DoA): the S per-antenna streams leave
// the AGC already as MUSIC items, out[t*S + s] = agc_s(x_s[t]) -- x(r,c) = in[c*m + r] of lib/baz_music_doa.cc:82-84
// -- so GNU Radio's interleave / streams_to_vector hop between the two blocks costs no extra pass over HBM.
// Workgroup = S waves (one per stream), tile = AGC_IT samples per stream: lane l owns AGC_IE consecutive samples
// (32 B contiguous per lane, 2 KiB contiguous per wave: no LDS needed on the way in).  The tile is transposed in
// LDS and written as one contiguous block of AGC_IT*S samples (32 KiB at S = 16), 16 B per thread.
//   agc_tile_kernel<0>  per (tile, stream): the map (A, S) of the tile  -> chunk_pair[stream][tile]
//   agc_carry_kernel    (shared with the planar form)                   -> carry_in[stream][tile]
//   agc_tile_kernel<1>  apply + interleave
// -------------------------------------------------------------------------------------------------------------
constexpr int AGC_IE = 4;                  // consecutive samples per lane
constexpr int AGC_IT = 64 * AGC_IE;        // 256 samples per stream per workgroup
constexpr int AGC_IMAX = 16;               // streams per context in this form

// ---- the fast path of a FULL tile whose values are in the ordinary range (round 3) ---------------------------------------
// What the tile kernels spend is instruction issue, not bytes: fp64 sqrt and division come with scaling and
// special-case handling around their Newton cores, the wave scan moved (A, S) pairs through ds_bpermute with a select per
// step, and every sample carried its "j < cnt" predicate.  For a tile of 256 valid samples whose |x|^2 and carry-in lie in
// [2^-400, 2^400] (every non-zero finite float input does) none of that is needed:
//   sqrt, division : the same rsq / rcp + fma refinement the compiler's expansion runs between its scaling steps -- the
//                    scaling is the identity in this range, so the results are the same bits (checked against __dsqrt_rn /
//                    __ddiv_rn on the GPU: tests/test_agc.py::test_fast_sqrt_and_division_are_the_rounded_ones);
//   scan           : A of a full lane is the constant a^4, so only S is scanned:  S_i = sum_{j<=i} (a^4)^(i-j) S_j  by
//                    DPP row shifts (1, 2, 4, 8 with the constants a^4, a^8, a^16, a^32) and the two row broadcasts
//                    (per-lane constants (a^4)^((lane & 15) + 1), (a^4)^((lane & 31) + 1)): 3 instructions per step.
// The constants come from the host (long double powers, rounded once): AgcParams::pw = [64][4] doubles
// {(a^4)^((l&15)+1), (a^4)^((l&31)+1), (a^4)^l, -}.  Tiles that do not qualify (stream tail, zeros, inf / NaN, denormal
// products) take the general path below, as before.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double agc_dpp(const double v)     // lanes without a source (or outside ROW_MASK) read 0.0
{
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, ROW_MASK, 0xF, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, ROW_MASK, 0xF, false);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
// 2^-400 <= v <= 2^400 (false for 0, denormals, inf, NaN, negative): one subtract and one compare on the high word
__device__ __forceinline__ bool agc_ordinary(const double v)
{
    const uint32_t hi = (uint32_t)(__builtin_bit_cast(uint64_t, v) >> 32);
    return (hi - 0x26F00000u) < (0x58F00000u - 0x26F00000u);
}
__device__ __forceinline__ double agc_mag2(const float2 x)
{
#pragma clang fp contract(off)
    const double d0 = x.x, d1 = x.y;
    const double p0 = d0 * d0, p1 = d1 * d1;
    return p0 + p1;                                                            // .cc:74-76
}
__device__ __forceinline__ double agc_sqrt_ordinary(const double x)            // == __dsqrt_rn(x) for agc_ordinary(x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    d = fma(-g, g, x);
    return fma(d, h, g);
}
__device__ __forceinline__ double agc_div_ordinary(const double n, const double d)   // == __ddiv_rn(n, d), both ordinary
{
    double y = __builtin_amdgcn_rcp(d);
    double t = fma(-d, y, 1.0);
    y = fma(t, y, y);
    t = fma(-d, y, 1.0);
    y = fma(t, y, y);
    const double q = n * y;
    const double r = fma(-d, q, n);
    return fma(r, y, q);
}

// MODE 0: tile maps; MODE 1: apply + interleave (workgroup = the S streams of one tile); MODE 2: apply, planar output
// with the optional env / gain ports (waves are dealt over (stream, tile) pairs, consecutive waves = consecutive tiles
// of one stream).  PLANAR_GRID selects the MODE-2 wave -> (stream, tile) mapping for MODE 0 as well.
template <int MODE, bool PLANAR_GRID = (MODE == 2)>
__global__ __launch_bounds__(1024) void agc_tile_kernel(const float2* __restrict__ in, uint64_t n, uint64_t stride,
                                                         AgcParams P, double2* __restrict__ chunk_pair,
                                                         const double* __restrict__ carry_in, uint32_t ntiles,
                                                         float2* __restrict__ out, double* __restrict__ env_state,
                                                         uint32_t nstreams, float* __restrict__ env_out = nullptr,
                                                         float* __restrict__ mul_out = nullptr)
{
    extern __shared__ float2 tile[];       // MODE 1: [nstreams][AGC_IT + 1]
    const int lane = threadIdx.x & 63;
    uint32_t stream, t;
    bool live = true;
    if constexpr (PLANAR_GRID) {
        const uint64_t gw = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (gw >= (uint64_t)ntiles * nstreams) return;
        stream = (uint32_t)(gw / ntiles);
        t = (uint32_t)(gw - (uint64_t)stream * ntiles);
    } else {
        stream = threadIdx.x >> 6;
        t = blockIdx.x;
    }
    (void)live;
    const uint64_t base = (uint64_t)t * AGC_IT;
    const uint32_t valid = (uint32_t)((n - base < (uint64_t)AGC_IT) ? (n - base) : AGC_IT);
    const int i0 = lane * AGC_IE;
    const int cnt = ((int)valid - i0) < 0 ? 0 : (((int)valid - i0) > AGC_IE ? AGC_IE : ((int)valid - i0));
    const float2* __restrict__ xin = in + (size_t)stream * stride + base + i0;
    float2 x[AGC_IE];
    double mag[AGC_IE];
    if (cnt == AGC_IE && (reinterpret_cast<uintptr_t>(xin) & 15u) == 0) {
#pragma unroll
        for (int j = 0; j < AGC_IE; j += 2) {
            const float4 v = *reinterpret_cast<const float4*>(xin + j);
            x[j] = make_float2(v.x, v.y);
            x[j + 1] = make_float2(v.z, v.w);
        }
    } else {
#pragma unroll
        for (int j = 0; j < AGC_IE; ++j) x[j] = (j < cnt) ? xin[j] : make_float2(0.f, 0.f);
    }
    // the fast path's per-lane constants and the tile's carry-in are requested before the input is needed
    double pw15 = 0.0, pw31 = 0.0, pwl = 0.0, carry = 0.0;
    const bool try_fast = P.pw != nullptr && valid == (uint32_t)AGC_IT;                // wave-uniform
    if (try_fast) {
        const double2 c01 = *reinterpret_cast<const double2*>(P.pw + 4 * lane);
        pw15 = c01.x; pw31 = c01.y;
        pwl = P.pw[4 * lane + 2];
    }
    if constexpr (MODE != 0) carry = carry_in[(size_t)stream * ntiles + t];
    bool fast = false;
    if (try_fast) {
        bool ok = (MODE == 0) || agc_ordinary(carry);
#pragma unroll
        for (int j = 0; j < AGC_IE; ++j) {
            mag[j] = agc_mag2(x[j]);                             // (|x|^2 for now)
            ok = ok && agc_ordinary(mag[j]);
        }
        fast = __all(ok);
    }
    float2 y[AGC_IE];
    [[maybe_unused]] float ev[AGC_IE], gv[AGC_IE];
    double e = 0.0;
    if (fast) {
        double S = 0.0;
#pragma unroll
        for (int j = 0; j < AGC_IE; ++j) {
            mag[j] = agc_sqrt_ordinary(mag[j]);                  // .cc:77
            S = fma(P.a, S, P.b * mag[j]);
        }
        S = fma(P.c1, agc_dpp<0x111, 0xF>(S), S);                // row_shr:1
        S = fma(P.c2, agc_dpp<0x112, 0xF>(S), S);                // row_shr:2
        S = fma(P.c4, agc_dpp<0x114, 0xF>(S), S);                // row_shr:4
        S = fma(P.c8, agc_dpp<0x118, 0xF>(S), S);                // row_shr:8
        S = fma(pw15, agc_dpp<0x142, 0xA>(S), S);                // row_bcast:15 into rows 1 and 3
        S = fma(pw31, agc_dpp<0x143, 0xC>(S), S);                // row_bcast:31 into rows 2 and 3
        if constexpr (MODE == 0) {
            if (lane == 63) chunk_pair[(size_t)stream * ntiles + t] = make_double2(P.a_tile, S);
            return;
        } else {
            e = fma(pwl, carry, agc_dpp<0x138, 0xF>(S));         // wave_shr:1: the inclusive value of the lane before
#pragma unroll
            for (int j = 0; j < AGC_IE; ++j) {
                e = agc_env_step(e, mag[j], P.a, P.b);            // .cc:82
                const double gain = agc_div_ordinary(P.reference, e);   // .cc:89
                y[j] = agc_apply(x[j], gain);                     // .cc:97-100
                if constexpr (MODE == 2) { ev[j] = (float)e; gv[j] = (float)gain; }   // .cc:85, 92
            }
        }
    } else {
        double A = 1.0, S = 0.0;
#pragma unroll
        for (int j = 0; j < AGC_IE; ++j) {
            mag[j] = agc_mag(x[j]);                              // .cc:74-77
            if (j < cnt) { S = fma(P.a, S, P.b * mag[j]); A *= P.a; }
        }
        double Ai = A, Si = S;                                   // inclusive wave scan of the lane maps
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double Ap = __shfl_up(Ai, d, 64), Sp = __shfl_up(Si, d, 64);
            if (lane >= d) compose(Ai, Si, Ap, Sp);
        }
        if constexpr (MODE == 0) {
            if (lane == 63) chunk_pair[(size_t)stream * ntiles + t] = make_double2(Ai, Si);
            return;
        } else {
            double Ae = __shfl_up(Ai, 1, 64), Se = __shfl_up(Si, 1, 64);
            if (lane == 0) { Ae = 1.0; Se = 0.0; }
            e = fma(Ae, carry, Se);                               // state entering this lane's run
#pragma unroll
            for (int j = 0; j < AGC_IE; ++j) {
                y[j] = make_float2(0.f, 0.f);
                if constexpr (MODE == 2) { ev[j] = 0.f; gv[j] = 0.f; }
                if (j < cnt) {
                    e = agc_env_step(e, mag[j], P.a, P.b);            // .cc:82
                    const double gain = __ddiv_rn(P.reference, e);    // .cc:89
                    y[j] = agc_apply(x[j], gain);                     // .cc:97-100
                    if constexpr (MODE == 2) { ev[j] = (float)e; gv[j] = (float)gain; }   // .cc:85, 92
                }
            }
        }
    }
    if constexpr (MODE != 0) {
        if (env_state && base + i0 + cnt == n && cnt > 0) env_state[stream] = e;
    }
    if constexpr (MODE == 2) {     // planar: the lane's AGC_IE outputs are 32 contiguous bytes of its stream
        const size_t o = (size_t)stream * stride + base + i0;
        float2* __restrict__ yo = out + o;
        if (cnt == AGC_IE && (reinterpret_cast<uintptr_t>(yo) & 15u) == 0) {
            *reinterpret_cast<float4*>(yo) = make_float4(y[0].x, y[0].y, y[1].x, y[1].y);
            *reinterpret_cast<float4*>(yo + 2) = make_float4(y[2].x, y[2].y, y[3].x, y[3].y);
        } else {
#pragma unroll
            for (int j = 0; j < AGC_IE; ++j) if (j < cnt) yo[j] = y[j];
        }
        if (env_out) {
            float* __restrict__ eo = env_out + o;
            if (cnt == AGC_IE && (reinterpret_cast<uintptr_t>(eo) & 15u) == 0) *reinterpret_cast<float4*>(eo) = make_float4(ev[0], ev[1], ev[2], ev[3]);
            else {
#pragma unroll
                for (int j = 0; j < AGC_IE; ++j) if (j < cnt) eo[j] = ev[j];
            }
        }
        if (mul_out) {
            float* __restrict__ mo = mul_out + o;
            if (cnt == AGC_IE && (reinterpret_cast<uintptr_t>(mo) & 15u) == 0) *reinterpret_cast<float4*>(mo) = make_float4(gv[0], gv[1], gv[2], gv[3]);
            else {
#pragma unroll
                for (int j = 0; j < AGC_IE; ++j) if (j < cnt) mo[j] = gv[j];
            }
        }
    }
    if constexpr (MODE == 1) {
        float2* __restrict__ row = tile + (size_t)stream * (AGC_IT + 1);
#pragma unroll
        for (int j = 0; j < AGC_IE; ++j)
            if (j < cnt) row[i0 + j] = y[j];
        __syncthreads();
        // contiguous write-out: element p of the tile block is (time p / S, stream p % S)
        float2* __restrict__ ob = out + (size_t)base * nstreams;
        const uint32_t total = valid * nstreams;
        if ((nstreams & 1u) == 0 && (reinterpret_cast<uintptr_t>(ob) & 15u) == 0) {
            for (uint32_t p = threadIdx.x * 2; p < total; p += blockDim.x * 2) {     // two streams of one time step
                const uint32_t tt = p / nstreams, ss = p - tt * nstreams;
                const float2 a = tile[(size_t)ss * (AGC_IT + 1) + tt], b = tile[(size_t)(ss + 1) * (AGC_IT + 1) + tt];
                *reinterpret_cast<float4*>(ob + p) = make_float4(a.x, a.y, b.x, b.y);
            }
        } else {
            for (uint32_t p = threadIdx.x; p < total; p += blockDim.x) {
                const uint32_t tt = p / nstreams, ss = p - tt * nstreams;
                ob[p] = tile[(size_t)ss * (AGC_IT + 1) + tt];
            }
        }
    }
}

// lab / test: the fast path's square root and division against the rounded library ones, bit for bit
__global__ void agc_selfcheck_kernel(const double* __restrict__ v, const double* __restrict__ w, uint64_t n,
                                     unsigned long long* __restrict__ bad)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = v[i], b = w[i];
    unsigned int k = 0;
    if (agc_ordinary(a) && __builtin_bit_cast(uint64_t, agc_sqrt_ordinary(a)) != __builtin_bit_cast(uint64_t, __dsqrt_rn(a))) k |= 1u;
    if (agc_ordinary(a) && agc_ordinary(b) &&
        __builtin_bit_cast(uint64_t, agc_div_ordinary(a, b)) != __builtin_bit_cast(uint64_t, __ddiv_rn(a, b))) k |= 2u;
    if (k & 1u) atomicAdd(bad, 1ull);
    if (k & 2u) atomicAdd(bad + 1, 1ull);
}

}  // namespace bazagc
