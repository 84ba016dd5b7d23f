// music_kernels.hip.h -- hand-written gfx950 (CDNA4, wave64) kernels for the MUSIC-DoA hot path.
//
// Replaces the arithmetic of baz_music_doa::work (/root/reference/lib/baz_music_doa.cc:72-161):
//   cov_mfma_kernel   .cc:74-85   widen c64->c128, x = reshape(m,K), R = x x^H / K
//   evd_proj_kernel   .cc:88-93   Hermitian EVD (ascending), noise basis G = first m-n eigenvectors;
//                                 emitted as the real coefficients of the projector Q = G G^H
//   scan_mfma_kernel  .cc:101-155 per-bin strength 1/||G^H a||^2 (as 1/(a^H Q a), an fp64 MFMA GEMM),
//                                 optional spectrum port, strict-">" top-n insertion
//   topn_merge_kernel .cc:129-155 final top-n across bin ranges, ang/lvl outputs
//
// Precision contract: inputs and the steering table stay fp32 in HBM
// (that is what the reference sees); every accumulation is fp64 (exact widening, like the
// reference's static_cast<gr_complexd>, .cc:77).  No CUDA-compat layer, gfx950 only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bazmusic {

typedef double v4f64 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wave_lds_fence()
{
    // LDS traffic of one wave is issued in order; this only stops the compiler from moving
    // LDS accesses across the hand-over point between "lane = producer" and "lane = consumer".
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// =====================================================================================
// 1. Spatial covariance R = x x^H / K as an fp64 MFMA outer-product reduction.
//
// One item is K time columns of M complex64 = ROWS = 2M floats per column, laid out in HBM
// exactly as the real 2M x K matrix X~ (rows re0,im0,re1,im1,...; lib/baz_music_doa.cc:82-84:
// x(r,c) = in[c*m + r]).  The real Gram matrix X~ X~^T (2M x 2M) holds everything:
//     Re R_ab = G[2a][2b] + G[2a+1][2b+1]      Im R_ab = G[2a+1][2b] - G[2a][2b+1]
// v_mfma_f64_16x16x4_f64 computes D(16x16) += A(16x4) B(4x16); lane l supplies A[l&15][l>>4]
// and B[l>>4][l&15].  With B = A^T both operands are the SAME register, so one fp32 load
// (widened exactly) feeds the instruction.  IPT = 16/ROWS items share a tile (block diagonal):
// M=4 -> 2 items/tile, M=8 -> 1.  Per MFMA a wave reads, per item, 4 columns x ROWS floats
// = one fully used contiguous segment (128 B at M=4, 256 B at M=8).
// =====================================================================================
template <int M>
__global__ __launch_bounds__(256) void cov_mfma_kernel(const float* __restrict__ in,
                                                        double2* __restrict__ R,
                                                        uint32_t batch, uint32_t K)
{
    constexpr int ROWS = 2 * M;
    constexpr int IPT = 16 / ROWS;
    constexpr int MM = M * M;
    constexpr int CH = 32;            // MFMA steps per chunk = loads in flight per lane per buffer
    static_assert(ROWS <= 16, "single-tile covariance handles m <= 8");
    __shared__ double gram[4][16 * 17];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15;          // tile row / column
    const int kk = lane >> 4;         // k index inside the MFMA (time column mod 4)
    const int sub = i / ROWS;         // item inside the tile
    const int row = i - sub * ROWS;   // float offset inside a time column
    const uint32_t ntiles = (batch + IPT - 1) / IPT;
    const size_t item_floats = (size_t)K * ROWS;
    const uint32_t steps = (K + 3) >> 2;
    const bool fast = (K % (4 * CH)) == 0;   // whole chunks, no column tail
    double* g = gram[wave];

    for (uint32_t tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        const uint32_t item = tile * IPT + sub;
        const bool live = (sub < IPT) && (item < batch);
        const float* p = in + (size_t)(live ? item : 0) * item_floats + kk * ROWS + row;
        const float keep = live ? 1.0f : 0.0f;
        v4f64 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};

        if (fast) {
            // Two register buffers of CH loads each: the next chunk's 32 loads (8 KiB per wave) are in
            // flight while the current chunk's 32 MFMAs (~2k cycles) retire.
            float va[CH], vb[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) va[u] = p[(size_t)u * (4 * ROWS)];
            for (uint32_t t = 0; t < steps; t += 2 * CH) {
                const bool more_b = (t + CH) < steps;
                if (more_b) {
#pragma unroll
                    for (int u = 0; u < CH; ++u) vb[u] = p[(size_t)(t + CH + u) * (4 * ROWS)];
                }
#pragma unroll
                for (int u = 0; u < CH; u += 2) {
                    const double a0 = (double)(va[u] * keep), a1 = (double)(va[u + 1] * keep);
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, acc1, 0, 0, 0);
                }
                if (more_b) {
                    if ((t + 2 * CH) < steps) {
#pragma unroll
                        for (int u = 0; u < CH; ++u) va[u] = p[(size_t)(t + 2 * CH + u) * (4 * ROWS)];
                    }
#pragma unroll
                    for (int u = 0; u < CH; u += 2) {
                        const double a0 = (double)(vb[u] * keep), a1 = (double)(vb[u + 1] * keep);
                        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, acc1, 0, 0, 0);
                    }
                }
            }
        } else {
            for (uint32_t t = 0; t < steps; ++t) {
                const bool ok = live && (4 * t + kk < K);
                const float v = ok ? p[(size_t)t * (4 * ROWS)] : 0.0f;
                const double a = (double)v;
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, a, acc0, 0, 0, 0);
            }
        }
        const v4f64 acc = acc0 + acc1;

        // C/D layout of the f64 MFMA: col = lane&15, row = (lane>>4) + 4*reg.
#pragma unroll
        for (int r = 0; r < 4; ++r) g[(kk + 4 * r) * 17 + i] = acc[r];
        wave_lds_fence();

        const double dK = (double)K;
        for (int e = lane; e < IPT * MM; e += 64) {
            const int s2 = e / MM, ab = e - s2 * MM;
            const int a = ab / M, b = ab - a * M;
            const int base = s2 * ROWS;
            const double re = g[(base + 2 * a) * 17 + base + 2 * b] + g[(base + 2 * a + 1) * 17 + base + 2 * b + 1];
            const double im = g[(base + 2 * a + 1) * 17 + base + 2 * b] - g[(base + 2 * a) * 17 + base + 2 * b + 1];
            const uint32_t it2 = tile * IPT + s2;
            if (it2 < batch) R[(size_t)it2 * MM + ab] = make_double2(re / dK, im / dK);   // .cc:85 "/ (double)average_over"
        }
        wave_lds_fence();
    }
}

// =====================================================================================
// 2. Batched Hermitian EVD (cyclic complex Jacobi, fp64), one item per lane, then the
//    noise-subspace projector Q = G G^H, G = eigenvectors of the (m-n) smallest eigenvalues
//    (lib/baz_music_doa.cc:88-93).  Only the subspace matters: eigenvector phase / basis
//    inside the noise subspace cancels in Q.
//
//    Output: MM real coefficients per item, item-minor (Qs[e*qstride + item]) so that the
//    scan kernel's lane-per-item loads are coalesced:
//        e = i*M+i : Q_ii          e = i*M+j (i<j) : 2 Re Q_ij       e = j*M+i (i<j) : -2 Im Q_ij
//    so that  a^H Q a = sum_e q[e] * F[e]  with the table F built by build_F() on the host.
// =====================================================================================
template <int M>
struct Herm {
    double Ar[M][M], Ai[M][M], Vr[M][M], Vi[M][M];
};

template <int M>
__device__ __forceinline__ void jacobi_rotate(Herm<M>& h, const int p, const int q)
{
    const double apr = h.Ar[p][q], api = h.Ai[p][q];
    const double g2 = apr * apr + api * api;
    // The matrix is normalised to max diagonal in [0.5,1) (see evd_proj_kernel), so this absolute
    // threshold (|a_pq| <= 1e-20) is 4 orders below fp64 resolution.  It also keeps lanes that
    // converged early (while the rest of the wave still sweeps) from rotating on off-diagonals that
    // have shrunk quadratically into the denormal range, where u = a_pq/|a_pq| loses unit modulus.
    const bool rot = g2 > 1e-40;
    const double gg = sqrt(g2);
    const double ig = rot ? 1.0 / gg : 0.0;
    const double ur = rot ? apr * ig : 1.0;
    const double ui = rot ? api * ig : 0.0;
    const double tau = (h.Ar[q][q] - h.Ar[p][p]) * 0.5 * ig;
    double t = copysign(1.0, tau) / (fabs(tau) + sqrt(1.0 + tau * tau));
    t = rot ? t : 0.0;
    const double c = 1.0 / sqrt(1.0 + t * t);
    const double s = t * c;
    const double sur = s * ur, sui = s * ui, cur = c * ur, cui = c * ui;
    // J = [[c, s],[-s conj(u), c conj(u)]] on (p,q);   A <- J^H A J,  V <- V J
#pragma unroll
    for (int k = 0; k < M; ++k) {   // A J : columns p,q
        const double xr = h.Ar[k][p], xi = h.Ai[k][p], yr = h.Ar[k][q], yi = h.Ai[k][q];
        h.Ar[k][p] = c * xr - (sur * yr + sui * yi);
        h.Ai[k][p] = c * xi - (sur * yi - sui * yr);
        h.Ar[k][q] = s * xr + (cur * yr + cui * yi);
        h.Ai[k][q] = s * xi + (cur * yi - cui * yr);
    }
#pragma unroll
    for (int k = 0; k < M; ++k) {   // J^H (A J) : rows p,q
        const double xr = h.Ar[p][k], xi = h.Ai[p][k], yr = h.Ar[q][k], yi = h.Ai[q][k];
        h.Ar[p][k] = c * xr - (sur * yr - sui * yi);
        h.Ai[p][k] = c * xi - (sur * yi + sui * yr);
        h.Ar[q][k] = s * xr + (cur * yr - cui * yi);
        h.Ai[q][k] = s * xi + (cur * yi + cui * yr);
    }
    h.Ar[p][q] = 0.0; h.Ai[p][q] = 0.0; h.Ar[q][p] = 0.0; h.Ai[q][p] = 0.0;
    h.Ai[p][p] = 0.0; h.Ai[q][q] = 0.0;
#pragma unroll
    for (int k = 0; k < M; ++k) {   // V J
        const double xr = h.Vr[k][p], xi = h.Vi[k][p], yr = h.Vr[k][q], yi = h.Vi[k][q];
        h.Vr[k][p] = c * xr - (sur * yr + sui * yi);
        h.Vi[k][p] = c * xi - (sur * yi - sui * yr);
        h.Vr[k][q] = s * xr + (cur * yr + cui * yi);
        h.Vi[k][q] = s * xi + (cur * yi - cui * yr);
    }
}

template <int M, bool UNROLL>
__device__ __forceinline__ void jacobi_sweep(Herm<M>& h)
{
    if constexpr (UNROLL) {
#pragma unroll
        for (int p = 0; p < M - 1; ++p)
#pragma unroll
            for (int q = p + 1; q < M; ++q) jacobi_rotate<M>(h, p, q);
    } else {
#pragma nounroll
        for (int p = 0; p < M - 1; ++p)
#pragma nounroll
            for (int q = p + 1; q < M; ++q) jacobi_rotate<M>(h, p, q);
    }
}

template <int M>
__global__ __launch_bounds__(64) void evd_proj_kernel(const double2* __restrict__ R,
                                                       double* __restrict__ Qs,
                                                       uint32_t batch, uint32_t n, uint32_t qstride)
{
    constexpr int MM = M * M;
    constexpr bool UNROLL = (M <= 4);   // register-resident, statically indexed; larger M uses private arrays
    constexpr int MAX_SWEEPS = 16;
    const uint32_t item = blockIdx.x * 64 + threadIdx.x;
    const bool valid = item < batch;
    const uint32_t itc = valid ? item : (batch - 1);

    Herm<M> h;
    const double2* Rp = R + (size_t)itc * MM;
    // The projector is invariant under R -> s R (s > 0): scale by an exact power of two so that the
    // largest diagonal entry lies in [0.5,1) (R is PSD, so every |R_ij| <= that).  LAPACK's zheev
    // (behind the reference's eig_sym, .cc:90) likewise rescales out-of-range matrices.
    double dmax = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i) dmax = fmax(dmax, fabs(Rp[i * M + i].x));
    int ex = 0;
    (void)frexp(dmax, &ex);
    const double scl = (dmax > 0.0 && dmax < __builtin_huge_val()) ? ldexp(1.0, -ex) : 1.0;
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const double2 v = Rp[i * M + j];
            h.Ar[i][j] = v.x * scl;
            h.Ai[i][j] = (i == j) ? 0.0 : v.y * scl;
            h.Vr[i][j] = (i == j) ? 1.0 : 0.0;
            h.Vi[i][j] = 0.0;
        }

    for (int sweep = 0; sweep < MAX_SWEEPS; ++sweep) {
        double off = 0.0, dia = 0.0;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            dia += h.Ar[i][i] * h.Ar[i][i];
#pragma unroll
            for (int j = i + 1; j < M; ++j) off += h.Ar[i][j] * h.Ar[i][j] + h.Ai[i][j] * h.Ai[i][j];
        }
        const bool done = !(off > 1e-33 * dia);   // also true for NaN input -> bounded loop either way
        if (__all(done)) break;
        jacobi_sweep<M, UNROLL>(h);
    }

    // ascending rank of each eigenvalue (ties -> lower column first), noise = rank < m-n
    double wk[M];
#pragma unroll
    for (int k = 0; k < M; ++k) wk[k] = h.Ar[k][k];
    double msk[M];
    const int nnoise = (int)M - (int)n;
#pragma unroll
    for (int k = 0; k < M; ++k) {
        int rank = 0;
#pragma unroll
        for (int j = 0; j < M; ++j) rank += (wk[j] < wk[k] || (wk[j] == wk[k] && j < k)) ? 1 : 0;
        msk[k] = (rank < nnoise) ? 1.0 : 0.0;
    }

#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = i; j < M; ++j) {
            double re = 0.0, im = 0.0;
#pragma unroll
            for (int k = 0; k < M; ++k) {
                re += msk[k] * (h.Vr[i][k] * h.Vr[j][k] + h.Vi[i][k] * h.Vi[j][k]);
                im += msk[k] * (h.Vi[i][k] * h.Vr[j][k] - h.Vr[i][k] * h.Vi[j][k]);
            }
            if (valid) {
                if (i == j) {
                    Qs[(size_t)(i * M + i) * qstride + item] = re;
                } else {
                    Qs[(size_t)(i * M + j) * qstride + item] = 2.0 * re;
                    Qs[(size_t)(j * M + i) * qstride + item] = -2.0 * im;
                }
            }
        }
}

// =====================================================================================
// 3. Top-n bookkeeping shared by the scan and merge kernels (lib/baz_music_doa.cc:95,129-141).
//    Lists are kept in d = 1/strength ascending order; strict "<" on d while bins arrive in
//    ascending order reproduces the reference's strict ">" insertion on strength (earliest bin
//    wins ties); NaN never inserts; untouched slots stay (d = +inf, bin 0) = (strength 0, angle 0).
// =====================================================================================
template <int NMAX>
__device__ __forceinline__ void topn_insert(double (&td)[NMAX], uint32_t (&tb)[NMAX], const double d, const uint32_t bin)
{
#pragma unroll
    for (int i = NMAX - 1; i >= 0; --i) {
        const bool ci = d < td[i];
        const bool cp = (i > 0) ? (d < td[(i > 0) ? i - 1 : 0]) : false;
        td[i] = ci ? (cp ? td[(i > 0) ? i - 1 : 0] : d) : td[i];
        tb[i] = ci ? (cp ? tb[(i > 0) ? i - 1 : 0] : bin) : tb[i];
    }
}

template <int NMAX>
__device__ __forceinline__ void topn_insert_lex(double (&td)[NMAX], uint32_t (&tb)[NMAX], const double d, const uint32_t bin)
{
    // merge step: candidates arrive out of bin order -> explicit (d, bin) lexicographic order
#pragma unroll
    for (int i = NMAX - 1; i >= 0; --i) {
        const int im = (i > 0) ? i - 1 : 0;
        const bool ci = (d < td[i]) || (d == td[i] && bin < tb[i]);
        const bool cp = (i > 0) ? ((d < td[im]) || (d == td[im] && bin < tb[im])) : false;
        td[i] = ci ? (cp ? td[im] : d) : td[i];
        tb[i] = ci ? (cp ? tb[im] : bin) : tb[i];
    }
}

__device__ __forceinline__ float strength_f32(const double d)
{
    // (float)(1.0/d) of the reference (.cc:114-121) evaluated as rcp_f32((float)d): <= ~2 ulp_f32
    // (2.4e-7 relative) from the correctly rounded value; the parity budget is 1e-5.
    return __builtin_amdgcn_rcpf((float)d);
}

// =====================================================================================
// 4. Pseudo-spectrum scan as an fp64 MFMA GEMM  D[bins x items] = F[bins x MM] * Q^T[MM x items].
//
//    Measured on gfx950 (scripts/ubench.hip, profiles/): v_mfma_f64_16x16x4_f64 retires 1024 FMAs in
//    65 cycles/SIMD (77 TFLOP/s) -- the same rate as v_fma_f64 (16 wave-instructions = 78 cycles) --
//    but its operands are lane-distributed, so the shared table F needs NO wave-uniform broadcast:
//    it streams through plain coalesced vector loads (deeply pipelined with vmcnt).  (A lane=item
//    v_fma_f64 form fed by scalar loads of F was measured first: 0.495 ms per 65,536 cfg-2 items,
//    72 % of its wave cycles in s_waitcnt on scalar-cache misses; this form: 0.289 ms. DESIGN.md 5.)
//
//    Tile: A = F (16 bin-rows x 4 k), B = Q^T (4 k x 16 items), K = MM (KS = ceil(MM/4) steps).
//    A-operand rows are permuted on the host (build_FA) so that accumulator register r of lane
//    l = (g = l>>4, c = l&15) is bin  bin0 + 4g + r  of item  item0 + 16*ti + c :
//    every lane owns 4 CONSECUTIVE bins of one item -> one 16-B store, 64 B contiguous per item
//    per store instruction, no LDS transpose.  A wave owns IT item-tiles (16*IT items) and a
//    contiguous range of 16-bin tiles; its per-lane top-n lists (over the bins it sees, ascending)
//    are merged across the 4 lanes of an item with wave shuffles, then across bin ranges by
//    topn_merge_kernel.
// =====================================================================================
template <int NMAX>
__device__ __forceinline__ void topn_merge_xor(double (&td)[NMAX], uint32_t (&tb)[NMAX], const int mask)
{
    double od[NMAX];
    uint32_t ob[NMAX];
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
        od[i] = __shfl_xor(td[i], mask, 64);
        ob[i] = (uint32_t)__shfl_xor((int)tb[i], mask, 64);
    }
#pragma unroll
    for (int i = 0; i < NMAX; ++i) topn_insert_lex<NMAX>(td, tb, od[i], ob[i]);
}

template <int M, int NMAX, int IT, bool SPEC, bool VEC4>
__global__ __launch_bounds__(256) void scan_mfma_kernel(const double* __restrict__ Qs,
                                                         const double* __restrict__ FA,
                                                         float* __restrict__ spec,
                                                         float* __restrict__ ang,
                                                         float* __restrict__ lvl,
                                                         double* __restrict__ cand_d,
                                                         uint32_t* __restrict__ cand_b,
                                                         uint32_t batch, uint32_t res, uint32_t n,
                                                         uint32_t qstride, uint32_t ntiles, uint32_t nsplit)
{
    constexpr int MM = M * M;
    constexpr int KS = (MM + 3) / 4;        // MFMA k-steps
    constexpr int ITEMS = 16 * IT;          // items per wave

    const int lane = threadIdx.x & 63;
    const int c = lane & 15, g = lane >> 4;
    const uint32_t wglob = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t igroup = wglob / nsplit;
    const uint32_t split = wglob - igroup * nsplit;
    const uint32_t item0 = igroup * ITEMS;
    if (item0 >= batch) return;             // wave-uniform
    const uint32_t t_begin = (uint32_t)(((uint64_t)ntiles * split) / nsplit);
    const uint32_t t_end = (uint32_t)(((uint64_t)ntiles * (split + 1)) / nsplit);

    // B operand: q[item0 + 16 ti + c][e = 4 s + g]   (zero for the K padding e >= MM)
    double qb[IT][KS];
#pragma unroll
    for (int ti = 0; ti < IT; ++ti) {
        const uint32_t it = item0 + 16 * ti + c;
        const uint32_t itc = (it < batch) ? it : (batch - 1);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int e = 4 * s + g;
            qb[ti][s] = (e < MM) ? Qs[(size_t)e * qstride + itc] : 0.0;
        }
    }

    double td[IT][NMAX];
    uint32_t tb[IT][NMAX];
#pragma unroll
    for (int ti = 0; ti < IT; ++ti)
#pragma unroll
        for (int i = 0; i < NMAX; ++i) { td[ti][i] = __builtin_huge_val(); tb[ti][i] = 0; }   // .cc:95

    // A operand stream: FA[tile][lane][s] (KS doubles per lane per tile, contiguous per wave)
    const double* __restrict__ fa = FA + ((size_t)t_begin * 64 + lane) * KS;
    double a_cur[KS], a_nxt[KS];
    if (t_begin < t_end) {
#pragma unroll
        for (int s = 0; s < KS; ++s) a_cur[s] = fa[s];
    }

    for (uint32_t t = t_begin; t < t_end; ++t) {
        fa += (size_t)64 * KS;
        // FA carries one extra (NaN) tile at the end, so the prefetch past the last tile is in bounds
#pragma unroll
        for (int s = 0; s < KS; ++s) a_nxt[s] = fa[s];

        v4f64 acc[IT];
#pragma unroll
        for (int ti = 0; ti < IT; ++ti) acc[ti] = (v4f64){0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int ti = 0; ti < IT; ++ti)
                acc[ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[s], qb[ti][s], acc[ti], 0, 0, 0);

        const uint32_t bin = t * 16 + 4 * g;          // this lane's first bin in the tile
        bool hit = false;
        double d[IT][4];
#pragma unroll
        for (int ti = 0; ti < IT; ++ti) {
            float sv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                d[ti][r] = fabs(acc[ti][r]);          // ||G^H a||^2 >= 0; padding bins are NaN (never insert)
                sv[r] = strength_f32(d[ti][r]);
                hit |= d[ti][r] < td[ti][NMAX - 1];
            }
            if constexpr (SPEC) {
                const uint32_t it = item0 + 16 * ti + c;
                if constexpr (VEC4) {                 // res % 4 == 0: the 4 bins are all in or all out
                    if (it < batch && bin < res)
                        *reinterpret_cast<float4*>(&spec[(size_t)it * res + bin]) = make_float4(sv[0], sv[1], sv[2], sv[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (it < batch && bin + r < res) spec[(size_t)it * res + bin + r] = sv[r];
                }
            }
        }
        if (__any(hit)) {
#pragma unroll
            for (int ti = 0; ti < IT; ++ti)
#pragma unroll
                for (int r = 0; r < 4; ++r) topn_insert<NMAX>(td[ti], tb[ti], d[ti][r], bin + r);
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) a_cur[s] = a_nxt[s];
    }

    // merge the 4 lanes (g = 0..3) that share an item, then emit
#pragma unroll
    for (int ti = 0; ti < IT; ++ti) {
        topn_merge_xor<NMAX>(td[ti], tb[ti], 16);
        topn_merge_xor<NMAX>(td[ti], tb[ti], 32);
        const uint32_t it = item0 + 16 * ti + c;
        if (g == 0 && it < batch) {
            if (nsplit == 1) {
#pragma unroll
                for (int i = 0; i < NMAX; ++i)
                    if (i < (int)n) {
                        ang[(size_t)it * n + i] = (float)((double)tb[ti][i] * 360.0 / (double)res);   // .cc:134,152
                        if (lvl) lvl[(size_t)it * n + i] = strength_f32(td[ti][i]);                    // .cc:153
                    }
            } else {
#pragma unroll
                for (int i = 0; i < NMAX; ++i) {
                    cand_d[((size_t)it * nsplit + split) * NMAX + i] = td[ti][i];
                    cand_b[((size_t)it * nsplit + split) * NMAX + i] = tb[ti][i];
                }
            }
        }
    }
}

// Final top-n over the per-bin-range candidate lists (one thread per item).
template <int NMAX>
__global__ __launch_bounds__(256) void topn_merge_kernel(const double* __restrict__ cand_d,
                                                          const uint32_t* __restrict__ cand_b,
                                                          float* __restrict__ ang, float* __restrict__ lvl,
                                                          uint32_t batch, uint32_t res, uint32_t n, uint32_t nsplit)
{
    const uint32_t it = blockIdx.x * 256 + threadIdx.x;
    if (it >= batch) return;
    double fd[NMAX];
    uint32_t fb[NMAX];
#pragma unroll
    for (int i = 0; i < NMAX; ++i) { fd[i] = __builtin_huge_val(); fb[i] = 0; }
    const size_t base = (size_t)it * nsplit * NMAX;
    for (uint32_t k = 0; k < nsplit * NMAX; ++k) {
        const double d = cand_d[base + k];
        const uint32_t b = cand_b[base + k];
        if (d < __builtin_huge_val()) topn_insert_lex<NMAX>(fd, fb, d, b);
    }
#pragma unroll
    for (int i = 0; i < NMAX; ++i)
        if (i < (int)n) {
            ang[(size_t)it * n + i] = (float)((double)fb[i] * 360.0 / (double)res);   // .cc:134,152
            if (lvl) lvl[(size_t)it * n + i] = strength_f32(fd[i]);                    // .cc:153
        }
}

}  // namespace bazmusic
