// music_wide_kernels.hip.h -- the MUSIC path for WIDE arrays, 17 <= m <= 64 antennas, m a run-time value.
//
// The reference has no antenna limit (/root/reference/lib/baz_music_doa.cc:45-50 only checks the table's shape); the
// kernels of music_kernels.hip.h are specialised per m <= 16 (registers hold m^2 projector coefficients per lane).  No
// BASELINE configuration uses more than 16 antennas, so this path is built for correctness and sane speed, not for a
// roofline: one workgroup per item, everything resident in LDS, the literal form 1 / ||G^H a||^2 of .cc:104-121
// evaluated directly (no projector, hence no refinement pass either).
//     cov_wide_kernel    .cc:74-85    R = x x^H / K                       (exact fp32 products, fp64 sums)
//     evd_wide_kernel    .cc:88-93    Hermitian Jacobi in tournament rounds -> the m-n noise eigenvectors G
//     scan_wide_kernel   .cc:104-121  strength = 1 / norm(G^H a)^2 per bin, in the reference's operation order
//     topn_wide_kernel   .cc:95,129-160  n strongest bins on the fp64 strengths, earlier bin first on ties
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

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

__global__ __launch_bounds__(WB) void evd_wide_kernel(const double2* __restrict__ R, double2* __restrict__ G, uint32_t m,
                                                      uint32_t n)
{
    extern __shared__ double2 sm2[];
    const uint32_t item = blockIdx.x, tid = threadIdx.x;
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
}

// ---------------------------------------------------------------------------------------------------------------
// 3. Scan.  One workgroup per (item, range of bins).  TA: the steering table transposed, [m][res] complex64 (bin-minor:
//    a thread per bin reads coalesced).  For every bin: c_k = sum_i conj(G_k[i]) a_i (.cc:110-114 "G.t() * a"),
//    ss = sum_k |c_k|^2, strength = 1 / pow(sqrt(ss), 2) (.cc:115-119), all fp64 like the reference; the fp64 strengths
//    go to S (for the top-n), their fp32 casts to the spectrum port (.cc:120-121) when it is wired.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WB) void scan_wide_kernel(const double2* __restrict__ G, const float2* __restrict__ TA,
                                                       double* __restrict__ S, float* __restrict__ spec, uint32_t m,
                                                       uint32_t n, uint32_t res, uint32_t bins_per_block)
{
    extern __shared__ double2 sg[];            // [(m - n)][m]
    const uint32_t item = blockIdx.x, tid = threadIdx.x, nn = m - n;
    for (uint32_t e = tid; e < nn * m; e += WB) sg[e] = G[(size_t)item * nn * m + e];
    __syncthreads();
    const uint32_t b0 = blockIdx.y * bins_per_block;
    const uint32_t b1 = (b0 + bins_per_block < res) ? b0 + bins_per_block : res;
    constexpr int KC = 8;
    for (uint32_t b = b0 + tid; b < b1; b += WB) {
        double ss = 0.0;
        for (uint32_t k0 = 0; k0 < nn; k0 += KC) {
            double cr[KC], ci[KC];
#pragma unroll
            for (int u = 0; u < KC; ++u) { cr[u] = 0.0; ci[u] = 0.0; }
            for (uint32_t i = 0; i < m; ++i) {
                const float2 af = TA[(size_t)i * res + b];
                const double ar = (double)af.x, ai = (double)af.y;     // .cc:110-112 widens the fp32 table
#pragma unroll
                for (int u = 0; u < KC; ++u) {
                    const uint32_t k = (k0 + u < nn) ? k0 + u : nn - 1;   // (clamped rows are not added below)
                    const double2 g = sg[k * m + i];
                    cr[u] += g.x * ar + g.y * ai;          // conj(g) a
                    ci[u] += g.x * ai - g.y * ar;
                }
            }
#pragma unroll
            for (int u = 0; u < KC; ++u)
                if (k0 + u < nn) ss += cr[u] * cr[u] + ci[u] * ci[u];
        }
        const double nrm = sqrt(ss);
        const double strength = 1.0 / (nrm * nrm);
        S[(size_t)item * res + b] = strength;
        if (spec) spec[(size_t)item * res + b] = (float)strength;
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
