// music_wide_kernels.hip.h -- the MUSIC path for WIDE arrays, 17 <= m <= 64 antennas, m a run-time value.
//
// The reference has no antenna limit (/root/reference/lib/baz_music_doa.cc:45-50 only checks the table's shape); the
// kernels of music_kernels.hip.h are specialised per m <= 16 (registers hold m^2 projector coefficients per lane).  No
// BASELINE configuration uses more than 16 antennas, so this path is built for correctness and sane speed, not for a
// roofline: one workgroup per item, everything resident in LDS, the literal form 1 / ||G^H a||^2 of .cc:104-121
// evaluated directly (no projector, hence no refinement pass either).
//     cov_wide_kernel    .cc:74-85    R = x x^H / K                       (exact fp32 products, fp64 sums)
//     evd_wide_kernel    .cc:88-93    Hermitian Jacobi in tournament rounds -> the m-n noise eigenvectors G
//     sub_wide_kernel    (few emitters) the signal subspace by orthogonal iteration; the Jacobi only for what it hands back
//     scan_wide_kernel   .cc:104-121  strength = 1 / norm(G^H a)^2 per bin: ||a||^2 - ||S^H a||^2 where that is accurate,
//                                     the reference's literal form near the nulls
//     topn_wide_kernel   .cc:95,129-160  n strongest bins on the fp64 strengths, earlier bin first on ties
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "music_kernels.hip.h"     // dpp_f64, cmul / cmulc

namespace bazwide {

constexpr int WB = 256;            // workgroup size of every kernel here
constexpr int COV_TC = 32;         // time columns staged per pass of the covariance
constexpr int EVD_MAX_SWEEPS = 40;

// ---------------------------------------------------------------------------------------------------------------
// 1. Covariance.  in: [batch][K][m] complex64 (the port's item, x(r, c) = in[c*m + r], .cc:76-84).  R: [batch][m][m].
//    Thread e owns entries e, e + 256, ...; R_ji is bitwise conj(R_ij): the same exact products, summed in the same order.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WB) void cov_wide_kernel(const float2* __restrict__ in, double2* __restrict__ R, uint32_t m,
                                                      uint32_t K)
{
    extern __shared__ float2 sx[];             // [COV_TC][m]
    const uint32_t item = blockIdx.x, tid = threadIdx.x, mm = m * m;
    const float2* __restrict__ x = in + (size_t)item * K * m;
    constexpr int EPT = 16;                    // m <= 64: m^2 <= 16 * 256
    double ar[EPT], ai[EPT];
    uint32_t ei[EPT], ej[EPT];
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        ar[u] = 0.0; ai[u] = 0.0;
        const uint32_t e = tid + WB * u, ec = e < mm ? e : 0u;
        ei[u] = ec / m; ej[u] = ec - ei[u] * m;
    }
    for (uint32_t t0 = 0; t0 < K; t0 += COV_TC) {
        const uint32_t nt = (K - t0 < (uint32_t)COV_TC) ? K - t0 : (uint32_t)COV_TC;
        for (uint32_t e = tid; e < nt * m; e += WB) sx[e] = x[(size_t)t0 * m + e];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < EPT; ++u) {
            if (tid + WB * u < mm) {
                double sr = ar[u], si = ai[u];
                for (uint32_t t = 0; t < nt; ++t) {
                    const float2 a = sx[t * m + ei[u]], b = sx[t * m + ej[u]];
                    // a conj(b): fp32 x fp32 products are exact in fp64 (.cc:77 widens first)
                    sr += (double)a.x * (double)b.x + (double)a.y * (double)b.y;
                    si += (double)a.y * (double)b.x - (double)a.x * (double)b.y;
                }
                ar[u] = sr; ai[u] = si;
            }
        }
        __syncthreads();
    }
    const double dK = (double)K;
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const uint32_t e = tid + WB * u;
        if (e < mm) R[(size_t)item * mm + e] = make_double2(ar[u] / dK, ai[u] / dK);    // .cc:85
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 1b. The covariance on the fp64 matrix core for 17 <= m <= 32 (round 3).  One wave per item.  The item is split into four
//     16-row real operands by ANTENNA BLOCK and PART -- Re0 / Im0 = real / imaginary parts of antennas 0..15, Re1 / Im1 of
//     antennas 16..31 -- so that lane (i, kk) fetches one complex sample (8 B; a wave instruction = 4 runs of 128 B) per block
//     and k-step, and the four real Grams that make up one entry of R sit in the SAME lane and register of four accumulators:
//         R[a][b] K = (Re_a Re_b^T + Im_a Im_b^T) + i (Im_a Re_b^T - Re_a Im_b^T)
//     block (1, 0): 4 products; blocks (0, 0) and (1, 1): 3 each (Re Re^T, Im Im^T, Im Re^T; Re Im^T is the transpose of the
//     last, fetched through LDS once per item) -> 10 v_mfma_f64_16x16x4 per k-step of 4 time columns.  Only entries a >= b
//     are formed; R[b][a] is written as the bitwise conjugate.  Rows of antennas >= m re-read antenna m - 1: row r of an
//     operand only reaches row / column r of a product, which is not stored.  fp32 x fp32 products are exact in fp64, the
//     sums are fp64 (.cc:77-85); their order differs from cov_wide_kernel's (last-bit differences in R).
//     The load ring (CH k-steps, re-armed slot by slot) runs across the chunks AND items of a wave.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void cov_wide_mfma_kernel(const float2* __restrict__ in, double2* __restrict__ R,
                                                               uint32_t batch, uint32_t m, uint32_t K)
{
    using bazmusic::v4f64;
    constexpr int CH = 8;
    __shared__ double tr[4][2][16 * 17];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, kk = lane >> 4;
    const uint32_t a0 = (uint32_t)i, a1 = (16u + (uint32_t)i < m) ? 16u + (uint32_t)i : m - 1;
    const uint32_t mm = m * m;
    const uint32_t steps = (K + 3) >> 2;            // k-steps of 4 time columns
    const uint32_t chunks = (K >> 2) / CH;          // whole chunks of CH full k-steps: the ring; the rest one by one
    const uint32_t stride = gridDim.x * 4;
    const double dK = (double)K;
    uint32_t item = blockIdx.x * 4 + wave;
    if (item >= batch) return;
    const size_t item_elems = (size_t)K * m;
    const size_t step_elems = (size_t)4 * m;
    const float2* __restrict__ src = in + (size_t)item * item_elems + (size_t)kk * m;
    float2 p0[CH], p1[CH];
    if (chunks) {
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            p0[u] = src[(size_t)u * step_elems + a0];
            p1[u] = src[(size_t)u * step_elems + a1];
        }
    }
    for (; item < batch; item += stride) {
        const uint32_t nitem = (item + stride < batch) ? item + stride : item;
        const float2* __restrict__ nsrc = in + (size_t)nitem * item_elems + (size_t)kk * m;
        v4f64 c00 = {0, 0, 0, 0}, c11 = c00, c10 = c00, d00 = c00, d11 = c00, d10 = c00, err = c00, eii = c00, eir = c00, eri = c00;
        auto kstep = [&](double r0, double i0, double r1, double i1) {
            c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(r0, r0, c00, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(i0, i0, c11, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(i0, r0, c10, 0, 0, 0);
            d00 = __builtin_amdgcn_mfma_f64_16x16x4f64(r1, r1, d00, 0, 0, 0);
            d11 = __builtin_amdgcn_mfma_f64_16x16x4f64(i1, i1, d11, 0, 0, 0);
            d10 = __builtin_amdgcn_mfma_f64_16x16x4f64(i1, r1, d10, 0, 0, 0);
            err = __builtin_amdgcn_mfma_f64_16x16x4f64(r1, r0, err, 0, 0, 0);
            eii = __builtin_amdgcn_mfma_f64_16x16x4f64(i1, i0, eii, 0, 0, 0);
            eir = __builtin_amdgcn_mfma_f64_16x16x4f64(i1, r0, eir, 0, 0, 0);
            eri = __builtin_amdgcn_mfma_f64_16x16x4f64(r1, i0, eri, 0, 0, 0);
        };
        for (uint32_t cg = 0; cg < chunks; ++cg) {
            const float2* __restrict__ rearm = (cg + 1 < chunks) ? src + (size_t)(cg + 1) * CH * step_elems : nsrc;   // wave-uniform
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const double r0 = (double)p0[u].x, i0 = (double)p0[u].y, r1 = (double)p1[u].x, i1 = (double)p1[u].y;   // exact (.cc:77)
                p0[u] = rearm[(size_t)u * step_elems + a0];
                p1[u] = rearm[(size_t)u * step_elems + a1];
                kstep(r0, i0, r1, i1);
            }
        }
        for (uint32_t t = chunks * CH; t < steps; ++t) {                  // K not a multiple of 32: the last k-steps, columns >= K are zero
            const bool ok = 4 * t + (uint32_t)kk < K;
            const float2 z = make_float2(0.0f, 0.0f);
            const float2 q0 = ok ? src[(size_t)t * step_elems + a0] : z;
            const float2 q1 = ok ? src[(size_t)t * step_elems + a1] : z;
            kstep((double)q0.x, (double)q0.y, (double)q1.x, (double)q1.y);
        }
        // Re Im^T of the two diagonal blocks = (Im Re^T)^T: through LDS (D layout: row = kk + 4 reg, col = i)
        double* t0 = tr[wave][0];
        double* t1 = tr[wave][1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            t0[(kk + 4 * r) * 17 + i] = c10[r];
            t1[(kk + 4 * r) * 17 + i] = d10[r];
        }
        bazmusic::wave_lds_fence();
        double2* __restrict__ Ri = R + (size_t)item * mm;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t row = (uint32_t)(kk + 4 * r), col = (uint32_t)i;
            if (row >= col) {                                             // diagonal blocks: the lower triangle, mirrored
                {
                    const double re = (c00[r] + c11[r]) / dK, im = (c10[r] - t0[col * 17 + row]) / dK;     // .cc:85
                    Ri[(size_t)row * m + col] = make_double2(re, im);
                    if (row != col) Ri[(size_t)col * m + row] = make_double2(re, -im);
                }
                const uint32_t a = 16u + row, b = 16u + col;
                if (a < m) {
                    const double re = (d00[r] + d11[r]) / dK, im = (d10[r] - t1[col * 17 + row]) / dK;
                    Ri[(size_t)a * m + b] = make_double2(re, im);
                    if (row != col) Ri[(size_t)b * m + a] = make_double2(re, -im);
                }
            }
            const uint32_t a = 16u + row;                                 // block (1, 0): a = 16 + row > b = col
            if (a < m) {
                const double re = (err[r] + eii[r]) / dK, im = (eir[r] - eri[r]) / dK;
                Ri[(size_t)a * m + col] = make_double2(re, im);
                Ri[(size_t)col * m + a] = make_double2(re, -im);
            }
        }
        bazmusic::wave_lds_fence();
        src = nsrc;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 1c. The covariance on the fp64 matrix core for 33 <= m <= 64 (round 3).  The antennas form NB = ceil(m / 16) blocks of 16;
//     a wave task = (item, block pair (A, B), A >= B): the 16 x 16 block R[16 A .. +16][16 B .. +16] from four real Grams,
//         R[a][b] K = (Re_A Re_B^T + Im_A Im_B^T) + i (Im_A Re_B^T - Re_A Im_B^T)
//     -- four v_mfma_f64_16x16x4 per k-step of 4 time columns from two 8-B loads per lane (antenna 16 A + i and 16 B + i of
//     column 4 t + kk), the four terms of an entry in the same lane and register of four accumulators.  A diagonal pair
//     (A = B) spends the fourth product on what is the transpose of the third; its lower triangle is stored and mirrored as
//     the bitwise conjugate, like cov_wide_mfma_kernel, and an off-diagonal pair writes its block and the conjugate transpose.
//     An item is read NB + 1 times (10 pairs touch 20 blocks of 4 at m = 64), out of L2 after the first.  Rows of antennas
//     >= m re-read antenna m - 1 and are not stored.  fp32 x fp32 products are exact in fp64, the sums are fp64 (.cc:77-85).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cov_wide_pairs_kernel(const float2* __restrict__ in, double2* __restrict__ R,
                                                             uint32_t batch, uint32_t m, uint32_t K, uint32_t npairs)
{
    using bazmusic::v4f64;
    constexpr int CH = 8;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15, kk = lane >> 4;
    const uint32_t mm = m * m;
    const uint32_t steps = (K + 3) >> 2;
    const uint32_t full = K >> 2;                      // k-steps whose four columns all exist
    const double dK = (double)K;
    const uint32_t ntasks = batch * npairs;
    for (uint32_t task = blockIdx.x * 4 + wave; task < ntasks; task += gridDim.x * 4) {
        const uint32_t item = task / npairs, pr = task - item * npairs;
        uint32_t A = 0;                                 // pair index -> (A, B), A >= B: pr = A (A + 1) / 2 + B   (wave-uniform)
        while ((A + 1) * (A + 2) / 2 <= pr) ++A;
        const uint32_t B = pr - A * (A + 1) / 2;
        const uint32_t aA = (16u * A + (uint32_t)i < m) ? 16u * A + (uint32_t)i : m - 1;
        const uint32_t aB = (16u * B + (uint32_t)i < m) ? 16u * B + (uint32_t)i : m - 1;
        const float2* __restrict__ src = in + (size_t)item * K * m + (size_t)kk * m;
        const size_t step_elems = (size_t)4 * m;
        v4f64 err = {0, 0, 0, 0}, eii = err, eir = err, eri = err;
        auto kstep = [&](const float2 pa, const float2 pb) {
            const double ra = (double)pa.x, ia = (double)pa.y, rb = (double)pb.x, ib = (double)pb.y;   // exact (.cc:77)
            err = __builtin_amdgcn_mfma_f64_16x16x4f64(ra, rb, err, 0, 0, 0);
            eii = __builtin_amdgcn_mfma_f64_16x16x4f64(ia, ib, eii, 0, 0, 0);
            eir = __builtin_amdgcn_mfma_f64_16x16x4f64(ia, rb, eir, 0, 0, 0);
            eri = __builtin_amdgcn_mfma_f64_16x16x4f64(ra, ib, eri, 0, 0, 0);
        };
        uint32_t t = 0;
        for (; t + CH <= full; t += CH) {
            float2 pa[CH], pb[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                pa[u] = src[(size_t)(t + u) * step_elems + aA];
                pb[u] = src[(size_t)(t + u) * step_elems + aB];
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) kstep(pa[u], pb[u]);
        }
        for (; t < steps; ++t) {                        // the last k-steps one by one; columns >= K are zero
            const bool ok = 4 * t + (uint32_t)kk < K;
            const float2 z = make_float2(0.0f, 0.0f);
            kstep(ok ? src[(size_t)t * step_elems + aA] : z, ok ? src[(size_t)t * step_elems + aB] : z);
        }
        double2* __restrict__ Ri = R + (size_t)item * mm;
#pragma unroll
        for (int r = 0; r < 4; ++r) {                   // D layout: row = kk + 4 r (block A), col = i (block B)
            const uint32_t a = 16u * A + (uint32_t)(kk + 4 * r), b = 16u * B + (uint32_t)i;
            if (a < m && b < m && (A != B || a >= b)) {
                const double re = (err[r] + eii[r]) / dK, im = (eir[r] - eri[r]) / dK;               // .cc:85
                Ri[(size_t)a * m + b] = make_double2(re, im);
                if (a != b) Ri[(size_t)b * m + a] = make_double2(re, -im);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 2. EVD.  One workgroup per item; A and V (complex128, row stride m + 1) in LDS.  A sweep = ME - 1 rounds of the circle
//    method (ME = m rounded up to even): the pairs of a round are disjoint, so all their rotations are computed at once
//    (one thread per pair), applied to the columns of A and V (A J, V J), then to the rows of A (J^H (A J)).
//    The rotation, the power-of-two scaling, the convergence test and the ranking are those of evd_proj_lds_kernel.
//    G: [batch][m - n][m] complex128, row k = the eigenvector of the k-th smallest eigenvalue (.cc:93 cols(0, m-n-1)).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double v, double* red)
{
    const uint32_t tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (uint32_t s = WB / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(WB) void evd_wide_kernel(const double2* __restrict__ R, double2* __restrict__ G,
                                                      double2* __restrict__ Ssig, uint32_t m, uint32_t n,
                                                      const uint8_t* __restrict__ only)
{
    extern __shared__ double2 sm2[];
    const uint32_t item = blockIdx.x, tid = threadIdx.x;
    if (only && !only[item]) return;                     // behind sub_wide_kernel: just the items it handed back
    const uint32_t ld = m + 1, me = m + (m & 1u), np = me / 2;
    double2* A = sm2;                                    // [m][ld]
    double2* V = A + (size_t)m * ld;                     // [m][ld]
    double* par = reinterpret_cast<double*>(V + (size_t)m * ld);   // [np][6]
    double* red = par + (size_t)np * 6;                  // [WB]
    int* prs = reinterpret_cast<int*>(red + WB);         // [np][2]
    int* sel = prs + 2 * np;                             // [m]

    double psum = 0.0, dmax = 0.0;
    for (uint32_t e = tid; e < m * m; e += WB) {
        const uint32_t i = e / m, j = e - i * m;
        double2 v = R[(size_t)item * m * m + e];
        psum += v.x + v.y;
        if (i == j) { v.y = 0.0; dmax = fmax(dmax, fabs(v.x)); }
        A[i * ld + j] = v;
        V[i * ld + j] = make_double2(i == j ? 1.0 : 0.0, 0.0);
    }
    const double poison = block_sum(psum * 0.0, red);    // NaN iff R holds a NaN / Inf (see evd_proj_kernel)
    {   // exact power-of-two normalisation: largest diagonal entry into [0.5, 1)
        red[tid] = dmax;
        __syncthreads();
        for (uint32_t s = WB / 2; s > 0; s >>= 1) {
            if (tid < s) red[tid] = fmax(red[tid], red[tid + s]);
            __syncthreads();
        }
        dmax = red[0];
        __syncthreads();
    }
    int ex = 0;
    (void)frexp(dmax, &ex);
    const double scl = (dmax > 0.0 && dmax < __builtin_huge_val()) ? ldexp(1.0, -ex) : 1.0;
    for (uint32_t e = tid; e < m * m; e += WB) {
        const uint32_t i = e / m, j = e - i * m;
        double2 v = A[i * ld + j];
        v.x *= scl; v.y *= scl;
        A[i * ld + j] = v;
    }
    __syncthreads();

    for (int sweep = 0; sweep < EVD_MAX_SWEEPS; ++sweep) {
        double off = 0.0, dia = 0.0;
        for (uint32_t e = tid; e < m * m; e += WB) {
            const uint32_t i = e / m, j = e - i * m;
            const double2 v = A[i * ld + j];
            if (i == j) dia += v.x * v.x; else off += v.x * v.x + v.y * v.y;
        }
        off = block_sum(off, red);
        dia = block_sum(dia, red);
        if (!(off > 2e-33 * dia)) break;                 // (off counts every off-diagonal twice); also leaves on NaN

        for (uint32_t r = 0; r + 1 < me; ++r) {
            if (tid < np) {
                // circle method: me - 1 players on a ring, player me - 1 fixed
                const uint32_t k = tid, ring = me - 1;
                uint32_t a = (k == 0) ? ring : (r + k) % ring;
                uint32_t b = (k == 0) ? r : (r + ring - k) % ring;
                int pp = (int)(a < b ? a : b), qq = (int)(a < b ? b : a);
                double c = 1.0, sn = 0.0, ur = 1.0, ui = 0.0;
                if ((uint32_t)qq < m) {                  // (a pair with the phantom index of an odd m idles)
                    const double2 apq = A[pp * ld + qq];
                    const double app = A[pp * ld + pp].x, aqq = A[qq * ld + qq].x;
                    const double g2 = apq.x * apq.x + apq.y * apq.y;
                    const bool rot = g2 > 1e-40;
                    const double gg = sqrt(g2);
                    const double ig = rot ? 1.0 / gg : 0.0;
                    ur = rot ? apq.x * ig : 1.0;
                    ui = rot ? apq.y * ig : 0.0;
                    const double tau = (aqq - app) * 0.5 * ig;
                    double t = copysign(1.0, tau) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    t = rot ? t : 0.0;
                    c = 1.0 / sqrt(1.0 + t * t);
                    sn = t * c;
                } else {
                    pp = -1;
                }
                double* p6 = par + 6 * k;
                p6[0] = c; p6[1] = sn; p6[2] = sn * ur; p6[3] = sn * ui; p6[4] = c * ur; p6[5] = c * ui;
                prs[2 * k] = pp; prs[2 * k + 1] = qq;
            }
            __syncthreads();
            // columns p, q of every row of A and V:  A J, V J
            for (uint32_t w = tid; w < np * m; w += WB) {
                const uint32_t k = w / m, row = w - k * m;
                const int pp = prs[2 * k], qq = prs[2 * k + 1];
                if (pp < 0) continue;
                const double* p6 = par + 6 * k;
                const double c = p6[0], s = p6[1], sur = p6[2], sui = p6[3], cur = p6[4], cui = p6[5];
                {
                    const double2 x = A[row * ld + pp], y = A[row * ld + qq];
                    A[row * ld + pp] = make_double2(c * x.x - (sur * y.x + sui * y.y), c * x.y - (sur * y.y - sui * y.x));
                    A[row * ld + qq] = make_double2(s * x.x + (cur * y.x + cui * y.y), s * x.y + (cur * y.y - cui * y.x));
                }
                {
                    const double2 x = V[row * ld + pp], y = V[row * ld + qq];
                    V[row * ld + pp] = make_double2(c * x.x - (sur * y.x + sui * y.y), c * x.y - (sur * y.y - sui * y.x));
                    V[row * ld + qq] = make_double2(s * x.x + (cur * y.x + cui * y.y), s * x.y + (cur * y.y - cui * y.x));
                }
            }
            __syncthreads();
            // rows p, q of every column of A:  J^H (A J)
            for (uint32_t w = tid; w < np * m; w += WB) {
                const uint32_t k = w / m, col = w - k * m;
                const int pp = prs[2 * k], qq = prs[2 * k + 1];
                if (pp < 0) continue;
                const double* p6 = par + 6 * k;
                const double c = p6[0], s = p6[1], sur = p6[2], sui = p6[3], cur = p6[4], cui = p6[5];
                const double2 x = A[pp * ld + col], y = A[qq * ld + col];
                double2 vp = make_double2(c * x.x - (sur * y.x - sui * y.y), c * x.y - (sur * y.y + sui * y.x));
                double2 vq = make_double2(s * x.x + (cur * y.x - cui * y.y), s * x.y + (cur * y.y + cui * y.x));
                if ((int)col == qq) vp = make_double2(0.0, 0.0);                          // a_pq := 0
                if ((int)col == pp) vq = make_double2(0.0, 0.0);                          // a_qp := 0
                if ((int)col == pp) vp.y = 0.0;                                           // real diagonal
                if ((int)col == qq) vq.y = 0.0;
                A[pp * ld + col] = vp;
                A[qq * ld + col] = vq;
            }
            __syncthreads();
        }
    }

    // ascending rank of the eigenvalues, ties -> lower column first (as evd_proj_lds_kernel); noise = rank < m - n
    if (tid < m) sel[tid] = 0;
    __syncthreads();
    if (tid < m) {
        const double wj = A[tid * ld + tid].x;
        int rank = 0;
        for (uint32_t l = 0; l < m; ++l) {
            const double wl = A[l * ld + l].x;
            rank += (wl < wj || (wl == wj && l < tid)) ? 1 : 0;
        }
        sel[rank] = (int)tid;                            // (NaN eigenvalues: every rank is 0; G is poisoned anyway)
    }
    __syncthreads();
    const uint32_t nn = m - n;
    for (uint32_t e = tid; e < nn * m; e += WB) {
        const uint32_t k = e / m, i = e - k * m;
        const double2 v = V[i * ld + sel[k]];
        G[((size_t)item * nn + k) * m + i] = make_double2(v.x + poison, v.y + poison);
    }
    if (Ssig)      // the signal eigenvectors (rank >= m - n), for the scan's short form
        for (uint32_t e = tid; e < n * m; e += WB) {
            const uint32_t k = e / m, i = e - k * m;
            const double2 v = V[i * ld + sel[nn + k]];
            Ssig[((size_t)item * n + k) * m + i] = make_double2(v.x + poison, v.y + poison);
        }
}

// ---------------------------------------------------------------------------------------------------------------
// 2b. The noise basis WITHOUT the eigen-decomposition for few emitters (P = n <= 4): the signal subspace by orthogonal
//     iteration, exactly the scheme of bazmusic::evd_sub_kernel (music_kernels.hip.h, section 2c) with m at run time:
//     one wave per item, lane j = row j (m <= 64), R in LDS, all reductions XOR butterflies over the 64 lanes (DPP inside
//     a row of 16, then lane ^ 16 and lane ^ 32: every lane ends with the same bits).  Items that do not converge are
//     flagged in `redo` and take evd_wide_kernel (launched with only = redo).  At m = 64 a step is ~64 LDS reads and ~600
//     instructions per lane against ~40 x 63 workgroup-synchronised rounds of the Jacobi.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_allsum(double v)
{
    using bazmusic::dpp_f64;
    v += dpp_f64<0x140>(v);                // row_mirror       lane ^ 15
    v += dpp_f64<0x141>(v);                // row_half_mirror  lane ^ 7
    v += dpp_f64<0x1B>(v);                 // quad_perm [3,2,1,0]
    v += dpp_f64<0xB1>(v);                 // quad_perm [1,0,3,2]
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

__device__ __forceinline__ double wave_allmax(double v)
{
    using bazmusic::dpp_f64;
    v = fmax(v, dpp_f64<0x140>(v));
    v = fmax(v, dpp_f64<0x141>(v));
    v = fmax(v, dpp_f64<0x1B>(v));
    v = fmax(v, dpp_f64<0xB1>(v));
    v = fmax(v, __shfl_xor(v, 16));
    v = fmax(v, __shfl_xor(v, 32));
    return v;
}

template <int P, bool TRI = false>
__global__ __launch_bounds__(64) void sub_wide_kernel(const double2* __restrict__ R, double2* __restrict__ G,
                                                      double2* __restrict__ Ssig, uint8_t* __restrict__ redo, uint32_t m)
{
    using bazmusic::cmul;
    using bazmusic::cmulc;
    constexpr int MAX_IT = 64;
    constexpr double BAIL2 = 0.36, TOL2 = 1.6e-29 * P;     // as evd_sub_kernel for m > 8
    extern __shared__ double2 sw[];
    const uint32_t item = blockIdx.x;
    const int j = threadIdx.x;
    // TRI (from 50 antennas on, where the full matrix leaves room for fewer than four of these one-wave workgroups per CU): R is
    // Hermitian bit for bit (the covariance kernels mirror it), so only its lower triangle is kept, T(a, b) = R[a][b] for a >= b
    // at a (a + 1) / 2 + b, and lane j reads its row as T(j, k) for k <= j and conj T(k, j) beyond -- 33 instead of 66 KiB at 64
    // antennas: four workgroups per CU instead of two (EVD stage 0.196 -> 0.145 ms at n = 2, 1.36 -> 0.72 ms at n = 8).  Below
    // that the full rows (stride m + 1, no index arithmetic, no bank conflicts) are faster: 17 .. 48 antennas lose 10-20 % in
    // the triangular form.  Same values in the same order either way: the same bits.
    const uint32_t ld = m + 1;
    double2* sR = sw;                                      // TRI: [m (m + 1) / 2], else [m][ld] (row j is read by lane j only)
    double2* sY = sR + (TRI ? ((size_t)m * (m + 1)) / 2 : (size_t)m * ld);   // [P][64]
    const bool row = (uint32_t)j < m;
    const uint32_t tj = TRI ? (uint32_t)j * ((uint32_t)j + 1u) / 2u : (uint32_t)j * ld;     // where row j starts
    auto elem = [&](uint32_t k) -> double2 {               // R[j][k] of the scaled matrix
        if constexpr (TRI) {
            const bool low = k <= (uint32_t)j;
            double2 v = sR[low ? tj + k : k * (k + 1u) / 2u + (uint32_t)j];
            if (!low) v.y = -v.y;
            return v;
        } else {
            return sR[tj + k];
        }
    };
    const uint32_t kend = TRI ? (uint32_t)j + 1u : m;      // entries of row j this lane owns

    double psum = 0.0, dj = 0.0;
    if (row) {
        for (uint32_t k = 0; k < kend; ++k) {
            double2 v = R[((size_t)item * m + j) * m + k];
            psum += v.x + v.y;
            if (k == (uint32_t)j) { v.y = 0.0; dj = fabs(v.x); }
            sR[tj + k] = v;
        }
    }
    const double poison = wave_allsum(psum * 0.0);
    const double dmax = wave_allmax(dj);
    int ex = 0;
    (void)frexp(dmax, &ex);
    const double scl = (dmax > 0.0 && dmax < __builtin_huge_val()) ? ldexp(1.0, -ex) : 1.0;
    if (row)
        for (uint32_t k = 0; k < kend; ++k) { double2 v = sR[tj + k]; v.x *= scl; v.y *= scl; sR[tj + k] = v; }
    if constexpr (TRI) bazmusic::wave_lds_fence();         // (rows read each other's entries from here on)

    auto orth = [&](double2 (&z)[P], double2 (&yn)[P]) -> bool {
        bool ok = true;
#pragma unroll
        for (int c = 0; c < P; ++c) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int c2 = 0; c2 < c; ++c2) {
                    const double2 t = cmulc(z[c], yn[c2]);
                    const double hr = wave_allsum(t.x), hi = wave_allsum(t.y);
                    z[c].x -= hr * yn[c2].x - hi * yn[c2].y;
                    z[c].y -= hr * yn[c2].y + hi * yn[c2].x;
                }
            const double n2 = wave_allsum(z[c].x * z[c].x + z[c].y * z[c].y);
            const bool good = n2 > 0.0 && n2 < __builtin_huge_val();
            ok = ok && good;
            const double inv = good ? 1.0 / sqrt(n2) : 0.0;
            yn[c] = make_double2(z[c].x * inv, z[c].y * inv);
        }
        return ok;
    };

    double2 y[P], z[P];
#pragma unroll
    for (int c = 0; c < P; ++c) z[c] = row ? elem((uint32_t)c) : make_double2(0.0, 0.0);
    bool ok = orth(z, y) && !(poison != poison);
    bool conv = false;
    double d2prev = __builtin_huge_val();
    for (int it = 0; it < MAX_IT && ok && !conv; ++it) {   // (one item per wave: the conditions are wave-uniform)
        bazmusic::wave_lds_fence();
#pragma unroll
        for (int c = 0; c < P; ++c) sY[c * 64 + j] = y[c];
        bazmusic::wave_lds_fence();
#pragma unroll
        for (int c = 0; c < P; ++c) z[c] = make_double2(0.0, 0.0);
        if (row) {
            for (uint32_t k = 0; k < m; ++k) {
                const double2 r = elem(k);
#pragma unroll
                for (int c = 0; c < P; ++c) {
                    const double2 yk = sY[c * 64 + k];
                    z[c].x += r.x * yk.x - r.y * yk.y;
                    z[c].y += r.x * yk.y + r.y * yk.x;
                }
            }
        }
        double2 yn[P];
        const bool ok2 = orth(z, yn);
        double dloc = 0.0;
#pragma unroll
        for (int b = 0; b < P; ++b) {
            double2 d = yn[b];
#pragma unroll
            for (int a = 0; a < P; ++a) {
                const double2 t = cmulc(yn[b], y[a]);
                const double cr = wave_allsum(t.x), ci = wave_allsum(t.y);
                d.x -= y[a].x * cr - y[a].y * ci;
                d.y -= y[a].x * ci + y[a].y * cr;
            }
            dloc += d.x * d.x + d.y * d.y;
        }
        const double d2 = wave_allsum(dloc);
        ok = ok2 && (d2 == d2);
#pragma unroll
        for (int c = 0; c < P; ++c) y[c] = yn[c];
        if (ok && d2 <= TOL2) conv = true;
        else if (it >= 2 && d2 > 100.0 * TOL2 && d2 > BAIL2 * d2prev) ok = false;
        d2prev = d2;
    }
    if (j == 0) redo[item] = conv ? 0 : 1;
    if (!conv) return;
    if (Ssig && row) {
#pragma unroll
        for (int c = 0; c < P; ++c) Ssig[((size_t)item * P + c) * m + j] = y[c];
    }

    // Householder completion of S (see evd_sub_kernel): G rows = columns P .. m-1 of H_0 .. H_{P-1}
    double2 v[P], w[P];
    double tau[P];
#pragma unroll
    for (int c = 0; c < P; ++c) w[c] = y[c];
#pragma unroll
    for (int c = 0; c < P; ++c) {
        const bool in = row && j >= c;
        const double2 x = in ? w[c] : make_double2(0.0, 0.0);
        const double nx2 = wave_allsum(x.x * x.x + x.y * x.y);
        const double xcr = wave_allsum(j == c ? x.x : 0.0), xci = wave_allsum(j == c ? x.y : 0.0);
        const double nx = sqrt(nx2), ax = sqrt(xcr * xcr + xci * xci);
        const double pr = ax > 0.0 ? xcr / ax : 1.0, pi = ax > 0.0 ? xci / ax : 0.0;
        v[c] = x;
        if (j == c) { v[c].x += pr * nx; v[c].y += pi * nx; }
        const double nv2 = wave_allsum(v[c].x * v[c].x + v[c].y * v[c].y);
        tau[c] = nv2 > 0.0 ? 2.0 / nv2 : 0.0;
#pragma unroll
        for (int b = c + 1; b < P; ++b) {
            const double2 t = cmulc(w[b], v[c]);
            const double sr = wave_allsum(t.x) * tau[c], si = wave_allsum(t.y) * tau[c];
            w[b].x -= v[c].x * sr - v[c].y * si;
            w[b].y -= v[c].x * si + v[c].y * sr;
        }
    }
    double2 beta[P][P];
#pragma unroll
    for (int c = 0; c < P; ++c)
#pragma unroll
        for (int c2 = c + 1; c2 < P; ++c2) {
            const double2 t = cmulc(v[c2], v[c]);
            beta[c][c2] = make_double2(wave_allsum(t.x), wave_allsum(t.y));
        }
    bazmusic::wave_lds_fence();
#pragma unroll
    for (int c = 0; c < P; ++c) sY[c * 64 + j] = v[c];
    bazmusic::wave_lds_fence();
    if (row) {
        const uint32_t nn = m - P;
        for (uint32_t k = P; k < m; ++k) {
            double2 coef[P];
            double2 gk = make_double2((uint32_t)j == k ? 1.0 : 0.0, 0.0);
#pragma unroll
            for (int c = P - 1; c >= 0; --c) {
                const double2 vk = sY[c * 64 + k];
                double2 s = make_double2(vk.x, -vk.y);
#pragma unroll
                for (int c2 = c + 1; c2 < P; ++c2) {
                    const double2 t = cmul(coef[c2], beta[c][c2]);
                    s.x -= t.x; s.y -= t.y;
                }
                coef[c] = make_double2(tau[c] * s.x, tau[c] * s.y);
                const double2 t = cmul(coef[c], v[c]);
                gk.x -= t.x; gk.y -= t.y;
            }
            G[((size_t)item * nn + (k - P)) * m + j] = gk;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 3. Scan.  One workgroup per (item, range of bins).  TA: the steering table transposed, [m][res] complex64 (bin-minor:
//    a thread per bin reads coalesced).  For every bin: c_k = sum_i conj(G_k[i]) a_i (.cc:110-114 "G.t() * a"),
//    ss = sum_k |c_k|^2, strength = 1 / pow(sqrt(ss), 2) (.cc:115-119), all fp64 like the reference; the fp64 strengths
//    go to S (for the top-n), their fp32 casts to the spectrum port (.cc:120-121) when it is wired.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WB) void scan_wide_kernel(const double2* __restrict__ G, const double2* __restrict__ Ssig,
                                                       const float2* __restrict__ TA, const double* __restrict__ A2,
                                                       double below, double* __restrict__ S, float* __restrict__ spec,
                                                       uint32_t m, uint32_t n, uint32_t res, uint32_t bins_per_block)
{
    extern __shared__ double2 sg[];            // [(m - n)][m] noise basis, then [n][m] signal basis (when Ssig)
    const uint32_t item = blockIdx.x, tid = threadIdx.x, nn = m - n;
    double2* ssig = sg + (size_t)nn * m;
    for (uint32_t e = tid; e < nn * m; e += WB) sg[e] = G[(size_t)item * nn * m + e];
    if (Ssig)
        for (uint32_t e = tid; e < n * m; e += WB) ssig[e] = Ssig[(size_t)item * n * m + e];
    __syncthreads();
    const uint32_t b0 = blockIdx.y * bins_per_block;
    const uint32_t b1 = (b0 + bins_per_block < res) ? b0 + bins_per_block : res;
    constexpr int KC = 8;
    for (uint32_t b = b0 + tid; b < b1; b += WB) {
        // sum_k |v_k^H a|^2 over the `cnt` vectors at `base`
        auto proj = [&](const double2* base, uint32_t cnt) -> double {
            double ss = 0.0;
            for (uint32_t k0 = 0; k0 < cnt; k0 += KC) {
                double cr[KC], ci[KC];
#pragma unroll
                for (int u = 0; u < KC; ++u) { cr[u] = 0.0; ci[u] = 0.0; }
                for (uint32_t i = 0; i < m; ++i) {
                    const float2 af = TA[(size_t)i * res + b];
                    const double ar = (double)af.x, ai = (double)af.y;     // .cc:110-112 widens the fp32 table
#pragma unroll
                    for (int u = 0; u < KC; ++u) {
                        const uint32_t k = (k0 + u < cnt) ? k0 + u : cnt - 1;   // (clamped rows are not added below)
                        const double2 g = base[k * m + i];
                        cr[u] += g.x * ar + g.y * ai;          // conj(g) a
                        ci[u] += g.x * ai - g.y * ar;
                    }
                }
#pragma unroll
                for (int u = 0; u < KC; ++u)
                    if (k0 + u < cnt) ss += cr[u] * cr[u] + ci[u] * ci[u];
            }
            return ss;
        };
        // Few emitters: ||G^H a||^2 = ||a||^2 - ||S^H a||^2 needs n instead of m - n inner products.  The difference loses
        // ~m eps ||a||^2 absolutely, so it is kept only above `below` (= m 1e-8 max ||a||^2, the rule of the specialised
        // kernels' refinement: relative error <~ 1e-7 there); at or below it -- near a null -- the literal form runs.
        double ss = 0.0;
        bool literal = true;
        if (Ssig) {
            const double dp = A2[b] - proj(ssig, n);
            if (dp > below) { ss = dp; literal = false; }
        }
        if (literal) ss = proj(sg, nn);
        const double nrm = sqrt(ss);
        const double strength = 1.0 / (nrm * nrm);
        S[(size_t)item * res + b] = strength;
        if (spec) spec[(size_t)item * res + b] = (float)strength;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 3b. The scan on the fp64 matrix core for 17 <= m <= 64, n <= 8 (round 3; scan_wide_kernel above stays for everything
//     else).  Short form  d = ||a||^2 - sum_c |s_c^H a|^2  (music_kernels.hip.h, SIG): per (item, bin) 2n real inner products
//     of length 2m -- Re and Im of s_c^H a against the table's real coordinates (re a_0, im a_0, re a_1, ...) -- as a
//     GEMM [4 items x 4 outputs] x [2m] . [2m x 64 bins] on v_mfma_f64_16x16x4: tile row = item + 4 output, so that the
//     four outputs of one (item, bin) land in the four accumulator registers of ONE lane (fp64 C/D: row = (lane >> 4) +
//     4 reg) and d is formed in place.  B = the raw-table image TB (bazmusic's build_TB: columns permuted so that a lane
//     holds 4 consecutive bins of a 64-bin step -> one 16-B spectrum store, 256 B contiguous per item row); the 4 waves
//     of a workgroup take 4 x 4 items and share every slice of TB through a double-buffered LDS stage (phases of <= 8
//     k-steps, KS = ceil(2m / 4) <= 32).  Where the difference is at or below `below` (near a null: it loses ~m eps ||a||^2
//     absolutely) the reference's literal form sum_k |g_k^H a|^2 (.cc:110-119) runs for that value on the vector unit, from
//     G as sub_wide / evd_wide wrote it.  Top-n: bazmusic's packed keys; candidates per bin range -> topn_merge_kernel.
//     With one emitter two of the four outputs are zero rows (half the matrix work is idle; n = 1 is rare at these widths).
// ---------------------------------------------------------------------------------------------------------------
//     PMAX = staged phases per 64-bin step the instantiation can hold: 2 covers m <= 32 (KS <= 16, 16 coefficient registers), 4
//     covers 33 <= m <= 64 (KS <= 32: 32 coefficient registers; the same code, two more phases per step).
//     NOUT = real outputs per item: 4 for n <= 2 (a tile = 4 items x 4 outputs, the four of an (item, bin) in the four accumulator
//     registers of one lane), 8 for n = 3, 4 (a tile = 2 items x 8 outputs, tile row = item + 2 output: an item's eight outputs
//     sit in lane groups g and g ^ 2, whose partial sums of squares meet through one cross-lane add; lane groups 2, 3 then only
//     mirror 0, 1), 16 for n = 5 .. 8 (one item per tile, two cross-lane adds).  The list length follows (2 / 4 / 8 keys).
template <bool SPEC, bool VEC4, int PMAX = 2, int NOUT = 4>
__global__ __launch_bounds__(256) void scan_wide_mfma_kernel(const double2* __restrict__ Ssig, const double2* __restrict__ G,
                                                             const double2* __restrict__ TB, const double* __restrict__ A2p,
                                                             const float2* __restrict__ TA, float* __restrict__ spec,
                                                             double* __restrict__ cand, uint32_t batch, uint32_t m, uint32_t n,
                                                             uint32_t res, uint32_t nsplit, uint32_t keep_mask, double below,
                                                             unsigned long long* __restrict__ count)
{
    using namespace bazmusic;
    constexpr int SCH = 8;                         // k-steps per staged phase
    __shared__ v2f64 stage[2][2 * SCH * 64];       // 2 x 16 KiB
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    const uint32_t KS = (2 * m + 3) >> 2;          // 9 .. 16 (PMAX = 2), 17 .. 32 (PMAX = 4)
    const uint32_t pps = (KS + SCH - 1) / SCH;     // phases per step (2 .. PMAX)
    const uint32_t split = blockIdx.x % nsplit;
    constexpr int IPT = 16 / NOUT;                 // items per tile: 4, 2 or 1
    constexpr int NK = NOUT / 2;                   // list length: 2, 4 or 8
    const uint32_t item0 = ((blockIdx.x / nsplit) * 4 + wave) * IPT;        // this wave's items
    const uint32_t nsteps = (res + 63u) >> 6;
    const uint32_t st_begin = (uint32_t)(((uint64_t)nsteps * split) / nsplit);
    const uint32_t st_end = (uint32_t)(((uint64_t)nsteps * (split + 1)) / nsplit);

    // A operand: tile row c = (item c % IPT, output c / IPT); output o = 2 cI + part: Re (part 0) / Im (part 1) of s_cI^H a,
    // as coefficients of the real coordinate e = 4 s + g = (antenna e >> 1, re / im)
    double sa[SCH * PMAX];
    {
        const uint32_t it_r = item0 + (uint32_t)(c & (IPT - 1));
        const uint32_t itr = (it_r < batch) ? it_r : (batch - 1);
        const int o = c / IPT, cI = o >> 1, part = o & 1;
#pragma unroll
        for (int s = 0; s < SCH * PMAX; ++s) {
            const uint32_t e = 4u * (uint32_t)s + (uint32_t)g, j = e >> 1;
            double v = 0.0;
            if ((uint32_t)s < KS && j < m && (uint32_t)cI < n) {
                const double2 sv = Ssig[((size_t)itr * n + cI) * m + j];
                v = part == 0 ? ((e & 1u) ? sv.y : sv.x) : ((e & 1u) ? sv.x : -sv.y);
            }
            sa[s] = v;
        }
    }
    const uint32_t it_g = item0 + (uint32_t)(g & (IPT - 1));                // the item whose d this lane holds
    const bool row_ok = it_g < batch && g < IPT;                            // (NOUT = 8: lane groups 2, 3 mirror 0, 1 and write nothing)
    const uint32_t itg = (it_g < batch) ? it_g : (batch - 1);
    double key[NK];
#pragma unroll
    for (int i = 0; i < NK; ++i) key[i] = key_empty();
    const uint32_t nobin = ~keep_mask;
    uint32_t refined = 0;

    // staging: chunk (2 sl + h) of phase (st, p) = TB chunk ((st * KS + p * SCH + sl) * 2 + h); 4 waves x up to 4 chunks
    const v2f64* __restrict__ tb = reinterpret_cast<const v2f64*>(TB) + lane;
    v2f64 sr0 = {0, 0}, sr1 = {0, 0}, sr2 = {0, 0}, sr3 = {0, 0};
    auto phase_chunks = [&](uint32_t p) -> uint32_t { const uint32_t left = KS - p * SCH; return 2u * (left < (uint32_t)SCH ? left : (uint32_t)SCH); };
    auto stage_load = [&](uint32_t st, uint32_t p) {
        const uint32_t nch = phase_chunks(p);
        const size_t ch0 = ((size_t)st * KS + (size_t)p * SCH) * 2;
        const uint32_t w = (uint32_t)wave;
        sr0 = tb[(ch0 + (w < nch ? w : nch - 1)) * 64];
        sr1 = tb[(ch0 + (w + 4 < nch ? w + 4 : nch - 1)) * 64];
        sr2 = tb[(ch0 + (w + 8 < nch ? w + 8 : nch - 1)) * 64];
        sr3 = tb[(ch0 + (w + 12 < nch ? w + 12 : nch - 1)) * 64];
    };
    auto stage_store = [&](int b, uint32_t p) {
        const uint32_t nch = phase_chunks(p);
        const uint32_t w = (uint32_t)wave;
        if (w < nch) stage[b][w * 64 + lane] = sr0;
        if (w + 4 < nch) stage[b][(w + 4) * 64 + lane] = sr1;
        if (w + 8 < nch) stage[b][(w + 8) * 64 + lane] = sr2;
        if (w + 12 < nch) stage[b][(w + 12) * 64 + lane] = sr3;
    };
    if (st_begin < st_end) {
        stage_load(st_begin, 0);
        stage_store(0, 0);
    }
    __syncthreads();
    int buf = 0;
    for (uint32_t st = st_begin; st < st_end; ++st) {
        v4f64 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = (v4f64){0, 0, 0, 0};
#pragma unroll
        for (int pc = 0; pc < PMAX; ++pc) {                                // (compile-time phase index: sa[] stays in registers)
            const uint32_t p = (uint32_t)pc;
            if (p >= pps) break;                                           // wave-uniform
            const bool last_p = p + 1 == pps;
            const bool more = !last_p || (st + 1 < st_end);
            if (more) stage_load(last_p ? st + 1 : st, last_p ? 0u : p + 1);
            const uint32_t nks = phase_chunks(p) >> 1;                     // k-steps of this phase (wave-uniform)
            if (pc == 0) {
#pragma unroll
                for (int sl = 0; sl < SCH; ++sl) {
                    const v2f64 f01 = stage[buf][(2 * sl) * 64 + lane], f23 = stage[buf][(2 * sl + 1) * 64 + lane];
                    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[sl], f01.x, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[sl], f01.y, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[sl], f23.x, acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[sl], f23.y, acc[3], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int sl = 0; sl < SCH; ++sl) {
                    if ((uint32_t)sl < nks) {
                        const v2f64 f01 = stage[buf][(2 * sl) * 64 + lane], f23 = stage[buf][(2 * sl + 1) * 64 + lane];
                        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[pc * SCH + sl], f01.x, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[pc * SCH + sl], f01.y, acc[1], 0, 0, 0);
                        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[pc * SCH + sl], f23.x, acc[2], 0, 0, 0);
                        acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[pc * SCH + sl], f23.y, acc[3], 0, 0, 0);
                    }
                }
            }
            if (more) stage_store(buf ^ 1, last_p ? 0u : p + 1);
            __syncthreads();
            buf ^= 1;
        }
        // d of (item g, bins 64 st + 4 c + t): ||a||^2 - sum of the four squared outputs
        const uint32_t bin = st * 64u + 4u * (uint32_t)c;
        const v4f64 a2v = *reinterpret_cast<const v4f64*>(A2p + (size_t)st * 64 + 4 * c);
        double d[4];
        bool low = false;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            double ssq = (acc[t][0] * acc[t][0] + acc[t][1] * acc[t][1]) + (acc[t][2] * acc[t][2] + acc[t][3] * acc[t][3]);
            if constexpr (NOUT == 16) ssq += __shfl_xor(ssq, 16, 64);      // (one item per tile: its outputs span all four lane groups)
            if constexpr (NOUT >= 8) ssq += __shfl_xor(ssq, 32, 64);       // the other outputs of the item (lane group g ^ 2)
            d[t] = a2v[t] - ssq;
            low |= !(d[t] > below) && (bin + t < res);        // (also a negative or NaN difference, like scan_wide_kernel)
        }
        if (__any(low)) {                      // near a null: the reference's literal form for those values (.cc:110-119)
            const uint32_t nn = m - n;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (!(!(d[t] > below) && (bin + t < res))) continue;
                double ss = 0.0;
                for (uint32_t k = 0; k < nn; ++k) {
                    double cr = 0.0, ci = 0.0;
                    const double2* __restrict__ gk = G + ((size_t)itg * nn + k) * m;
                    for (uint32_t i = 0; i < m; ++i) {
                        const float2 af = TA[(size_t)i * res + bin + t];
                        const double ar = (double)af.x, ai = (double)af.y;
                        const double2 gv = gk[i];
                        cr += gv.x * ar + gv.y * ai;                        // conj(g) a
                        ci += gv.x * ai - gv.y * ar;
                    }
                    ss += cr * cr + ci * ci;
                }
                d[t] = ss;
                refined += row_ok ? 1u : 0u;
            }
        }
        v4f32 sv;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            sv[t] = strength_f32(fabs(d[t]));
            key_insert_new<NK>(key, make_key(d[t], (bin + t < res) ? bin + t : nobin, keep_mask));
        }
        if constexpr (SPEC) {
            if (row_ok) {
                float* __restrict__ dst = spec + (size_t)it_g * res + bin;
                if (VEC4 && bin + 3 < res) *reinterpret_cast<v4f32*>(dst) = sv;
                else {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (bin + t < res) dst[t] = sv[t];
                }
            }
        }
    }
    if (count) {
#pragma unroll
        for (int msk = 1; msk < 64; msk <<= 1) refined += __shfl_xor(refined, msk, 64);
        if (lane == 0 && refined) atomicAdd(count, (unsigned long long)refined);
    }
    key_merge_xor<NK>(key, 1);
    key_merge_xor<NK>(key, 2);
    key_merge_xor<NK>(key, 4);
    key_merge_xor<NK>(key, 8);
    if (c == 0 && row_ok) {
#pragma unroll
        for (int i = 0; i < NK; ++i) cand[((size_t)it_g * nsplit + split) * NK + i] = key[i];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 4. Top-n (.cc:95, 129-141: a list of n (angle, strength) pairs initialised to (0, 0); bins in ascending order, a bin is
//    inserted before the first entry with STRICTLY smaller strength).  Equivalent: the n largest strengths > 0, descending,
//    the earlier bin first among equals; NaN never enters.  One workgroup per item, n selection passes over the row.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WB) void topn_wide_kernel(const double* __restrict__ S, float* __restrict__ ang,
                                                       float* __restrict__ lvl, uint32_t res, uint32_t n)
{
    __shared__ double rs[WB];
    __shared__ uint32_t rb[WB];
    const uint32_t item = blockIdx.x, tid = threadIdx.x;
    const double* __restrict__ row = S + (size_t)item * res;
    double prev_s = __builtin_huge_val();
    uint32_t prev_b = 0xFFFFFFFFu;             // "before bin 0": (inf, -1) in the order (strength desc, bin asc)
    bool first = true;
    for (uint32_t k = 0; k < n; ++k) {
        double best = 0.0;                     // must beat the initial 0.0 strictly
        uint32_t bb = 0xFFFFFFFFu;
        for (uint32_t b = tid; b < res; b += WB) {
            const double s = row[b];
            const bool after = first || s < prev_s || (s == prev_s && b > prev_b);
            if (after && (s > best || (s == best && bb != 0xFFFFFFFFu && b < bb))) { best = s; bb = b; }
        }
        rs[tid] = best; rb[tid] = bb;
        __syncthreads();
        for (uint32_t st = WB / 2; st > 0; st >>= 1) {
            if (tid < st) {
                const double s2 = rs[tid + st]; const uint32_t b2 = rb[tid + st];
                const double s1 = rs[tid]; const uint32_t b1 = rb[tid];
                const bool take = (b2 != 0xFFFFFFFFu) && (b1 == 0xFFFFFFFFu || s2 > s1 || (s2 == s1 && b2 < b1));
                if (take) { rs[tid] = s2; rb[tid] = b2; }
            }
            __syncthreads();
        }
        const double ws = rs[0];
        const uint32_t wb = rb[0];
        __syncthreads();
        const bool used = wb != 0xFFFFFFFFu;
        if (tid == 0) {
            ang[(size_t)item * n + k] = used ? (float)((double)wb * 360.0 / (double)res) : 0.0f;   // .cc:134,152
            if (lvl) lvl[(size_t)item * n + k] = used ? (float)ws : 0.0f;
        }
        if (!used) {                           // nothing left: the remaining entries stay (0, 0)
            for (uint32_t k2 = k + 1 + tid; k2 < n; k2 += WB) {
                ang[(size_t)item * n + k2] = 0.0f;
                if (lvl) lvl[(size_t)item * n + k2] = 0.0f;
            }
            break;
        }
        prev_s = ws; prev_b = wb; first = false;
    }
}

}  // namespace bazwide
