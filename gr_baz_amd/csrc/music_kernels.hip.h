// music_kernels.hip.h -- hand-written gfx950 (CDNA4, wave64) kernels for the MUSIC-DoA hot path.
//
// Replaces the arithmetic of baz_music_doa::work (/root/reference/lib/baz_music_doa.cc:72-161):
//   cov_mfma_kernel   .cc:74-85   widen c64->c128, x = reshape(m,K), R = x x^H / K
//   evd_proj_kernel   .cc:88-93   Hermitian EVD (ascending), noise basis G = first m-n eigenvectors;
//                                 emitted as the real coefficients of the projector Q = G G^H
//   scan_mfma_kernel  .cc:101-141 per-bin strength 1/||G^H a||^2 (as 1/(a^H Q a), an fp64 MFMA GEMM; near-null
//                                 tiles again in the reference's literal form ||G^H a||^2, literal_tile()),
//                                 optional spectrum port, per-range top-n candidates
//   topn_merge_kernel .cc:129-155 final top-n across bin ranges, ang/lvl outputs
//   peak_pick_kernel  (opt-in extension, no reference counterpart) n strongest local maxima
//
// Precision contract: inputs and the steering table stay fp32 in HBM
// (that is what the reference sees); every accumulation is fp64 (exact widening, like the
// reference's static_cast<gr_complexd>, .cc:77).  No CUDA-compat layer, gfx950 only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bazmusic {

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef float v4f32 __attribute__((ext_vector_type(4)));
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));

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
        constexpr bool PAD = (16 % ROWS) != 0;       // tile rows beyond IPT*ROWS carry no item (m = 3, 5, 6, 7)
        const uint32_t item = tile * IPT + (PAD ? ((sub < IPT) ? sub : 0) : sub);
        const bool live = (sub < IPT) && (item < batch);
        // batch tail: a missing item re-reads the last one; its Gram block is simply not written back
        const float* p = in + (size_t)((item < batch) ? item : (batch - 1)) * item_floats + kk * ROWS + row;
        const float keep = (!PAD || sub < IPT) ? 1.0f : 0.0f;
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
                    const double a0 = PAD ? (double)(va[u] * keep) : (double)va[u];
                    const double a1 = PAD ? (double)(va[u + 1] * keep) : (double)va[u + 1];
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
                        const double a0 = PAD ? (double)(vb[u] * keep) : (double)vb[u];
                        const double a1 = PAD ? (double)(vb[u + 1] * keep) : (double)vb[u + 1];
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

// -------------------------------------------------------------------------------------
// 1b. The same covariance for 9 <= m <= 16: the realified item has 2m = 18..32 rows = two 16-row tiles
//     X0 (rows 0..15) and X1 (rows 16..31, zero beyond 2m).  Gram blocks G00 = X0 X0^T, G11 = X1 X1^T and
//     G10 = X1 X0^T (G01 is its transpose): three MFMAs per k-step from two fp32 loads per lane.
// -------------------------------------------------------------------------------------
template <int M>
__global__ __launch_bounds__(256) void cov_mfma2_kernel(const float* __restrict__ in,
                                                         double2* __restrict__ R,
                                                         uint32_t batch, uint32_t K)
{
    constexpr int ROWS = 2 * M;
    constexpr int MM = M * M;
    constexpr int CH = 16;
    static_assert(ROWS > 16 && ROWS <= 32, "two-tile covariance handles 9 <= m <= 16");
    __shared__ double gram[4][32 * 33];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = lane & 15;
    const int kk = lane >> 4;
    const bool hi_row = (16 + i) < ROWS;          // row 16+i exists
    const size_t item_floats = (size_t)K * ROWS;
    const uint32_t steps = (K + 3) >> 2;
    const bool fast = (K % (4 * CH)) == 0;
    double* g = gram[wave];

    for (uint32_t item = blockIdx.x * 4 + wave; item < batch; item += gridDim.x * 4) {
        const float* p0 = in + (size_t)item * item_floats + kk * ROWS + i;
        const float* p1 = p0 + (hi_row ? 16 : 0);
        const float keep1 = hi_row ? 1.0f : 0.0f;
        v4f64 a00 = {0, 0, 0, 0}, a11 = {0, 0, 0, 0}, a10 = {0, 0, 0, 0};
        if (fast) {
            for (uint32_t t = 0; t < steps; t += CH) {
                float v0[CH], v1[CH];
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    v0[u] = p0[(size_t)(t + u) * (4 * ROWS)];
                    v1[u] = p1[(size_t)(t + u) * (4 * ROWS)];
                }
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    const double x0 = (double)v0[u], x1 = (double)(v1[u] * keep1);
                    a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, a00, 0, 0, 0);
                    a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x1, a11, 0, 0, 0);
                    a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x0, a10, 0, 0, 0);
                }
            }
        } else {
            for (uint32_t t = 0; t < steps; ++t) {
                const bool ok = (4 * t + kk) < K;
                const double x0 = ok ? (double)p0[(size_t)t * (4 * ROWS)] : 0.0;
                const double x1 = (ok && hi_row) ? (double)p1[(size_t)t * (4 * ROWS)] : 0.0;
                a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, a00, 0, 0, 0);
                a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x1, a11, 0, 0, 0);
                a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x0, a10, 0, 0, 0);
            }
        }
        // D layout: col = lane&15 (i), row = kk + 4r.   G10[row][col] = sum X1[row] X0[col]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = kk + 4 * r;
            g[row * 33 + i] = a00[r];
            g[(16 + row) * 33 + 16 + i] = a11[r];
            g[(16 + row) * 33 + i] = a10[r];
            g[i * 33 + 16 + row] = a10[r];         // G01 = G10^T
        }
        wave_lds_fence();
        const double dK = (double)K;
        for (int e = lane; e < MM; e += 64) {
            const int a = e / M, b = e - a * M;
            const double re = g[(2 * a) * 33 + 2 * b] + g[(2 * a + 1) * 33 + 2 * b + 1];
            const double im = g[(2 * a + 1) * 33 + 2 * b] - g[(2 * a) * 33 + 2 * b + 1];
            R[(size_t)item * MM + e] = make_double2(re / dK, im / dK);   // .cc:85
        }
        wave_lds_fence();
    }
}

// -------------------------------------------------------------------------------------
// 1c. Covariance for m = 4, K % 256 == 0 (cfg2, the headline shape) -- round 2.
//
//     What bounds the kernels above at m = 4 is not HBM but how they ask for it: one dword per lane = 256 B per
//     wave-instruction, 4x the load instructions and address cycles per byte of a 16-B-per-lane stream (measured,
//     scripts/ubench_hbm.hip: 5.5-6.0 TB/s for dword streams against 7.0 TB/s for dwordx4; cov_mfma_kernel<4> itself
//     4.9-5.5 TB/s), and the 16x16x4 tiles spend half their flops on the zero blocks between the two items they hold.
//     Here a wave streams ONE item at a time with global_load_dwordx4 (1 KiB = 32 time columns per instruction, 8
//     instructions = 8 KiB always in flight per wave; ONE workgroup per CU, see baz_music_create), and the MFMA operand layout -- one matrix row per lane, which a
//     16-B load of 4 consecutive rows cannot supply -- is produced by a per-wave LDS transpose:
//         lane l of a chunk holds rows 4(l&1)..+3 of column l>>1  ->  widened (exactly) to fp64  ->  T[row][col] in LDS
//         operand P: lane (i = l&3, h = (l>>2)&1, w = (l>>3)&1, k = l>>4) reads T[4h+i][8k+4w .. +3]   (4 MFMA steps)
//         operand Q: the same with the other half of the rows, T[4(1-h)+i][..]
//     v_mfma_f64_4x4x4_4b_f64 multiplies four independent 4x4x4 blocks (block = (lane>>2)&3 = (w, h); layout measured
//     with scripts/probe_mfma4.hip: A[i][k] and B[k][j] in lane i|j + 4*block + 16*k, D[i][j] in lane j + 4*block
//     + 16*i; CBSZ/ABID do not broadcast for this opcode).  With X_h = rows 4h..4h+3 of the realified item:
//         D1 += P P^T  -> block (w, h) accumulates X_h X_h^T over its quarter of the columns
//         D2 += P Q^T  -> block (w, 0) accumulates X_0 X_1^T  (block (w, 1) the transposed duplicate)
//     i.e. every issued flop but that duplicate is useful: 2 x 20.3 cycles per 8 columns instead of 65.
//     The Gram blocks go through LDS once per item to form R (.cc:85), exactly Hermitian by construction.
// -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cov4_x4_kernel(const float* __restrict__ in, double2* __restrict__ R,
                                                      uint32_t batch, uint32_t K)
{
    // Row stride of the transposed chunk (8 rows x 32 columns of doubles).  34: the 16-B operand reads below then
    // start at 4-bank slot (17*row + 4k + 2w) mod 16 -- 4 lanes per slot, the minimum for a 1-KiB wave read (36 puts
    // 8 lanes on each of 8 slots).  Measured: no difference (0.396 ms either way); the LDS is not what bounds this kernel.
    constexpr int RSD = 34;
    __shared__ double stage[4][2][8 * RSD];       // per wave, double-buffered
    __shared__ double gram[4][2][64];             // per wave: D1, D2
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t chunks = K >> 5;               // 1-KiB chunks per item
    // writer role: column l>>1, rows 4(l&1)..+3
    const int wcol = lane >> 1, wrow = 4 * (lane & 1);
    // reader role (MFMA operand layout)
    const int ri = lane & 3, rh = (lane >> 2) & 1, rw = (lane >> 3) & 1, rk = lane >> 4;
    const int p_off = (4 * rh + ri) * RSD + 8 * rk + 4 * rw;
    const int q_off = (4 * (1 - rh) + ri) * RSD + 8 * rk + 4 * rw;
    double* const g1 = gram[wave][0];
    double* const g2 = gram[wave][1];
    const double dK = (double)K;

    // The 8-deep load ring runs ACROSS items (a slot is re-armed, unconditionally, with the next 8-chunk group of
    // this item or the first group of the wave's next item; past the end it re-reads the last item), so the stream
    // never drains at an item boundary and the compiler can count vmcnt exactly.  K % 256 == 0: 8-chunk groups.
    const uint32_t groups = chunks >> 3;
    const uint32_t stride = gridDim.x * 4;
    uint32_t item = blockIdx.x * 4 + wave;
    if (item >= batch) return;
    const v4f32* __restrict__ src = reinterpret_cast<const v4f32*>(in + (size_t)item * K * 8) + lane;
    v4f32 pf[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) pf[u] = __builtin_nontemporal_load(src + (size_t)u * 64);
    for (; item < batch; item += stride) {
        const uint32_t nitem = (item + stride < batch) ? item + stride : item;
        const v4f32* __restrict__ nsrc = reinterpret_cast<const v4f32*>(in + (size_t)nitem * K * 8) + lane;
        double a1 = 0.0, b1 = 0.0, a2 = 0.0, b2 = 0.0;       // two accumulator pairs: independent MFMA chains
        for (uint32_t cg = 0; cg < groups; ++cg) {
            const v4f32* __restrict__ rearm = (cg + 1 < groups) ? src + (size_t)(cg + 1) * 8 * 64 : nsrc;   // wave-uniform
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                double* __restrict__ T = stage[wave][u & 1];
#pragma unroll
                for (int j = 0; j < 4; ++j) T[(wrow + j) * RSD + wcol] = (double)pf[u][j];   // exact widening (.cc:77)
                // re-arm the slot only after its values are consumed: the load lands in the SAME registers (a copy at
                // the loop's back edge would have to wait for every load in flight)
                asm volatile("" ::: "memory");
                pf[u] = __builtin_nontemporal_load(rearm + (size_t)u * 64);
                wave_lds_fence();
                const v4f64 P = *reinterpret_cast<const v4f64*>(T + p_off);
                const v4f64 Q = *reinterpret_cast<const v4f64*>(T + q_off);
                a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(P[0], P[0], a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f64_4x4x4f64(P[0], Q[0], a2, 0, 0, 0);
                b1 = __builtin_amdgcn_mfma_f64_4x4x4f64(P[1], P[1], b1, 0, 0, 0);
                b2 = __builtin_amdgcn_mfma_f64_4x4x4f64(P[1], Q[1], b2, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(P[2], P[2], a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f64_4x4x4f64(P[2], Q[2], a2, 0, 0, 0);
                b1 = __builtin_amdgcn_mfma_f64_4x4x4f64(P[3], P[3], b1, 0, 0, 0);
                b2 = __builtin_amdgcn_mfma_f64_4x4x4f64(P[3], Q[3], b2, 0, 0, 0);
                wave_lds_fence();     // (the buffer is rewritten two chunks later; LDS ops of a wave retire in order)
            }
        }
        src = nsrc;
        // D layout: lane = j + 4*(2w+h) + 16*i.  G[x][y] (x, y < 8 realified rows):
        //   same half h:   sum_w D1[(y&3) + 4(2w+h) + 16(x&3)]      x < 4 <= y:   sum_w D2[(y-4) + 8w + 16x]
        g1[lane] = a1 + b1;
        g2[lane] = a2 + b2;
        wave_lds_fence();
        if (lane < 16) {
            const int a = lane >> 2, b = lane & 3;
            auto G = [&](int x, int y) -> double {
                if ((x >> 2) == (y >> 2)) {
                    const int hh = x >> 2, o = (y & 3) + 4 * hh + 16 * (x & 3);
                    return g1[o] + g1[o + 8];
                }
                if (x > y) { const int t = x; x = y; y = t; }      // G is symmetric: the X_1 X_0^T block is the transpose
                const int o = (y - 4) + 16 * x;
                return g2[o] + g2[o + 8];
            };
            // Re R_ab = G[2a][2b] + G[2a+1][2b+1]      Im R_ab = G[2a+1][2b] - G[2a][2b+1]
            const double re = G(2 * a, 2 * b) + G(2 * a + 1, 2 * b + 1);
            const double im = G(2 * a + 1, 2 * b) - G(2 * a, 2 * b + 1);
            R[(size_t)item * 16 + lane] = make_double2(re / dK, im / dK);    // .cc:85 "/ (double)average_over"
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
    // A = the Hermitian iterate: real diagonal D and the strict UPPER triangle U (entries i < j only; the lower one
    // is its conjugate and is never formed -- round 1 carried and rotated the full matrix: 16 instead of 4 complex
    // entry updates per rotation at m = 4).  V = accumulated rotations (eigenvectors in its columns).
    double D[M], Ur[M][M], Ui[M][M], Vr[M][M], Vi[M][M];
};

// One complex Jacobi rotation on the pair (p, q), p < q, both compile-time after unrolling:
//     J = [[c, s], [-s conj(u), c conj(u)]] on (p, q),  u = a_pq / |a_pq|,  A <- J^H A J,  V <- V J
// with t = tan of the rotation angle chosen to annihilate a_pq (the smaller root), so that
//     a_pp <- a_pp - t |a_pq|,   a_qq <- a_qq + t |a_pq|,   a_pq <- 0,
// and for every k outside the pair the two entries that couple k with p and q, taken from / stored into the upper
// triangle:   k < p:      (U_kp, U_kq) <- (c U_kp - conj(s u) U_kq,   s U_kp + conj(c u) U_kq)
//             p < k < q:  (U_pk, U_kq) <- (c U_pk - (s u) conj(U_kq), s conj(U_pk) + conj(c u) U_kq)
//             k > q:      (U_pk, U_qk) <- (c U_pk - (s u) U_qk,       s U_pk + (c u) U_qk)
template <int M>
__device__ __forceinline__ void jacobi_rotate(Herm<M>& h, const int p, const int q, const bool active)
{
    const double apr = h.Ur[p][q], api = h.Ui[p][q];
    const double g2 = apr * apr + api * api;
    // The matrix is normalised to max diagonal in [0.5,1) (see evd_project_lane), so this absolute
    // threshold (|a_pq| <= 1e-20) is 4 orders below fp64 resolution.  It also keeps lanes that
    // converged early (while the rest of the wave still sweeps) from rotating on off-diagonals that
    // have shrunk quadratically into the denormal range, where u = a_pq/|a_pq| loses unit modulus.
    // `active` is false for a lane whose item has already converged while its wave-mates still sweep: it then applies
    // the exact identity (c = 1, s = 0), so an item's result never depends on which items share its wave
    const bool rot = active && g2 > 1e-40;
    // (Replacing the IEEE divide / sqrt sequences below by v_rcp_f64 / v_rsq_f64 seeds + Newton steps removed 14 % of
    // the kernel's VALU instructions and not one microsecond: 0.065 vs 0.066 ms per 262,144 items.)
    const double gg = sqrt(g2);
    const double ig = rot ? 1.0 / gg : 0.0;
    const double tau = (h.D[q] - h.D[p]) * 0.5 * ig;
    double t = copysign(1.0, tau) / (fabs(tau) + sqrt(1.0 + tau * tau));
    t = rot ? t : 0.0;
    const double c = 1.0 / sqrt(1.0 + t * t);
    const double ur = rot ? apr * ig : 1.0;
    const double ui = rot ? api * ig : 0.0;
    const double s = t * c;
    const double sur = s * ur, sui = s * ui, cur = c * ur, cui = c * ui;
    const double shift = rot ? t * gg : 0.0;
    h.D[p] -= shift;
    h.D[q] += shift;
    h.Ur[p][q] = rot ? 0.0 : apr;
    h.Ui[p][q] = rot ? 0.0 : api;
#pragma unroll
    for (int k = 0; k < M; ++k) {
        if (k == p || k == q) continue;
        if (k < p) {
            const double xr = h.Ur[k][p], xi = h.Ui[k][p], yr = h.Ur[k][q], yi = h.Ui[k][q];
            h.Ur[k][p] = c * xr - (sur * yr + sui * yi);
            h.Ui[k][p] = c * xi - (sur * yi - sui * yr);
            h.Ur[k][q] = s * xr + (cur * yr + cui * yi);
            h.Ui[k][q] = s * xi + (cur * yi - cui * yr);
        } else if (k < q) {
            const double xr = h.Ur[p][k], xi = h.Ui[p][k], yr = h.Ur[k][q], yi = h.Ui[k][q];
            h.Ur[p][k] = c * xr - (sur * yr + sui * yi);       // c x - (s u) conj(y)
            h.Ui[p][k] = c * xi - (sui * yr - sur * yi);
            h.Ur[k][q] = s * xr + (cur * yr + cui * yi);       // s conj(x) + conj(c u) y
            h.Ui[k][q] = -s * xi + (cur * yi - cui * yr);
        } else {
            const double xr = h.Ur[p][k], xi = h.Ui[p][k], yr = h.Ur[q][k], yi = h.Ui[q][k];
            h.Ur[p][k] = c * xr - (sur * yr - sui * yi);
            h.Ui[p][k] = c * xi - (sur * yi + sui * yr);
            h.Ur[q][k] = s * xr + (cur * yr - cui * yi);
            h.Ui[q][k] = s * xi + (cur * yi + cui * yr);
        }
    }
#pragma unroll
    for (int k = 0; k < M; ++k) {   // V J
        const double xr = h.Vr[k][p], xi = h.Vi[k][p], yr = h.Vr[k][q], yi = h.Vi[k][q];
        h.Vr[k][p] = c * xr - (sur * yr + sui * yi);
        h.Vi[k][p] = c * xi - (sur * yi - sui * yr);
        h.Vr[k][q] = s * xr + (cur * yr + cui * yi);
        h.Vi[k][q] = s * xi + (cur * yi - cui * yr);
    }
}

template <int M>
__device__ __forceinline__ void jacobi_sweep(Herm<M>& h, const bool active)
{
#pragma unroll
    for (int p = 0; p < M - 1; ++p)
#pragma unroll
        for (int q = p + 1; q < M; ++q) jacobi_rotate<M>(h, p, q, active);
}

// The per-lane EVD + projector of one item (m <= 4, register resident, statically indexed), in pieces:
//     evd_begin (scale, load)  ->  { evd_check ; jacobi_sweep } per sweep  ->  evd_finish (ranks, G, Q)
// evd_project_lane runs them.  Every lane of the wave must take part (the early exit is wave-uniform).  Shared by evd_proj_kernel and the fused cov4_evd_kernel,
// which produce the same bits.
constexpr int EVD_MAX_SWEEPS = 16;

template <int M>
struct EvdState {
    Herm<M> h;
    double poison;     // NaN iff some entry of R is NaN or +-Inf, else +-0
    bool done;         // this lane's item has converged: its rotations are exact identities from now on
};

// getR(i, j) returns R_ij as double2
template <int M, class GetR>
__device__ __forceinline__ void evd_begin(GetR getR, EvdState<M>& st)
{
    static_assert(M <= 4, "register-resident, statically indexed (m >= 5 uses evd_proj_lds_kernel)");
    Herm<M>& h = st.h;
    // The projector is invariant under R -> s R (s > 0): scale by an exact power of two so that the
    // largest diagonal entry lies in [0.5,1) (R is PSD, so every |R_ij| <= that).  LAPACK's zheev
    // (behind the reference's eig_sym, .cc:90) likewise rescales out-of-range matrices.
    double dmax = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i) dmax = fmax(dmax, fabs(getR(i, i).x));
    int ex = 0;
    (void)frexp(dmax, &ex);
    const double scl = (dmax > 0.0 && dmax < __builtin_huge_val()) ? ldexp(1.0, -ex) : 1.0;
    double psum = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i)
#pragma unroll
        for (int j = 0; j < M; ++j) {
            const double2 v = getR(i, j);
            psum += v.x + v.y;                     // (every entry, also the lower triangle: NaN / Inf anywhere poisons)
            if (i == j) h.D[i] = v.x * scl;
            if (i < j) { h.Ur[i][j] = v.x * scl; h.Ui[i][j] = v.y * scl; }
            h.Vr[i][j] = (i == j) ? 1.0 : 0.0;
            h.Vi[i][j] = 0.0;
        }
    // A covariance with NaN/Inf entries has no eigen-decomposition (the reference's eig_sym fails there): poison the
    // projector so that the item's spectrum is NaN and no bin is ever inserted (.cc:131) -> (0, 0) outputs.
    st.poison = psum * 0.0;
    st.done = false;
}

// Convergence test at the head of a sweep: sets st.done, returns true when every lane of the wave is done.
template <int M>
__device__ __forceinline__ bool evd_check(EvdState<M>& st)
{
    const Herm<M>& h = st.h;
    double off = 0.0, dia = 0.0;
#pragma unroll
    for (int i = 0; i < M; ++i) {
        dia += h.D[i] * h.D[i];
#pragma unroll
        for (int j = i + 1; j < M; ++j) off += h.Ur[i][j] * h.Ur[i][j] + h.Ui[i][j] * h.Ui[i][j];
    }
    st.done = !(off > 1e-33 * dia);               // also true for NaN input -> bounded loop either way
    return __all(st.done);
}

template <int M>
__device__ __forceinline__ void evd_finish(const EvdState<M>& st, const bool valid, const uint32_t item, const uint32_t n,
                                           const uint32_t qstride, double* __restrict__ Qs, double* __restrict__ Gs)
{
    const Herm<M>& h = st.h;
    const double poison = st.poison;
    // ascending rank of each eigenvalue (ties -> lower column first), noise = rank < m-n
    double wk[M];
#pragma unroll
    for (int k = 0; k < M; ++k) wk[k] = h.D[k];
    double msk[M];
    const int nnoise = (int)M - (int)n;
#pragma unroll
    for (int k = 0; k < M; ++k) {
        int rank = 0;
#pragma unroll
        for (int j = 0; j < M; ++j) rank += (wk[j] < wk[k] || (wk[j] == wk[k] && j < k)) ? 1 : 0;
        msk[k] = (rank < nnoise) ? 1.0 : 0.0;
        // the noise eigenvectors themselves, for the literal-form refinement of near-null tiles (literal_tile() in the scan):
        // Gs[((rank*M + i)*2 + {re,im}) * qstride + item] = V[i][k]
        if (valid && Gs && rank < nnoise) {
#pragma unroll
            for (int i = 0; i < M; ++i) {
                Gs[(size_t)((rank * M + i) * 2) * qstride + item] = h.Vr[i][k];
                Gs[(size_t)((rank * M + i) * 2 + 1) * qstride + item] = h.Vi[i][k];
            }
        }
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
                    Qs[(size_t)(i * M + i) * qstride + item] = re + poison;
                } else {
                    Qs[(size_t)(i * M + j) * qstride + item] = 2.0 * re + poison;
                    Qs[(size_t)(j * M + i) * qstride + item] = -2.0 * im + poison;
                }
            }
        }
}

// (An orthogonal-iteration form of this function for n <= m/2 -- the m <= 4 counterpart of evd_sub_kernel, section 2c --
// was built and measured: 7 steps of ~380 instructions instead of ~4,900 at cfg2 / 20 dB, yet only 1.127 -> 1.120 ms per
// step, because its inner products are dependent fp64 chains where the rotations have 4-way ILP, and 1.3 % slower at
// 0 dB where the iteration is abandoned after 3 steps.  Not shipped; profiles/r02_subspace_iteration.txt.)
template <int M, class GetR>
__device__ __forceinline__ void evd_project_lane(GetR getR, const bool valid, const uint32_t item, const uint32_t n,
                                                 const uint32_t qstride, double* __restrict__ Qs, double* __restrict__ Gs)
{
    EvdState<M> st;
    evd_begin<M>(getR, st);
    for (int sweep = 0; sweep < EVD_MAX_SWEEPS; ++sweep) {
        if (evd_check<M>(st)) break;
        jacobi_sweep<M>(st.h, !st.done);
    }
    evd_finish<M>(st, valid, item, n, qstride, Qs, Gs);
}

template <int M>
__global__ __launch_bounds__(64) void evd_proj_kernel(const double2* __restrict__ R,
                                                       double* __restrict__ Qs,
                                                       uint32_t batch, uint32_t n, uint32_t qstride,
                                                       double* __restrict__ Gs)
{
    constexpr int MM = M * M;
    const uint32_t item = blockIdx.x * 64 + threadIdx.x;
    const bool valid = item < batch;
    const uint32_t itc = valid ? item : (batch - 1);
    const double2* Rp = R + (size_t)itc * MM;
    evd_project_lane<M>([&](int i, int j) { return Rp[i * M + j]; }, valid, item, n, qstride, Qs, Gs);
}

// -------------------------------------------------------------------------------------
// 2a. Covariance AND EVD in one kernel for m = 4, K % 256 == 0: a wave streams 64 consecutive items through the
//     covariance of cov4_x4_kernel (1c), parks each R (upper triangle, 16 doubles) in LDS, then runs the lane-per-item
//     Jacobi of evd_project_lane on the 64 of them and writes Q (and G).  R never touches HBM, and nothing is written
//     while the wave streams: the stand-alone covariance lost 0.06 ms to its 67 MB of 256-B R stores interleaved with
//     the read stream (its arithmetic is free at 2 waves/SIMD), see profiles/r02_fused_covevd_priorities.txt.
//     The EVD is NOT hidden: every wave streams at the same rate, so all of them reach their EVD phase together --
//     2 x (153 us of stream + 40 us of rotations) per 262,144 items.  Staggered phases and a rotation-per-item state
//     machine were measured slower (the rotations' fp64 VALU work and the stream's fp64 MFMAs share the DP pipe).
//     A streaming wave runs at s_setprio 3, an EVD phase at 0 (worth 0.04 ms against equal priorities once the waves
//     drift apart).
// -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cov4_evd_kernel(const float* __restrict__ in, double* __restrict__ Qs,
                                                       double* __restrict__ Gs, double2* __restrict__ Rdbg,
                                                       uint32_t batch, uint32_t K, uint32_t n, uint32_t qstride,
                                                       uint32_t task_items = 64)
{
    // task_items (64, 32 or 16; round 5): items per wave task.  64 fills the lane-per-item EVD; a SMALL batch -- a host-fed work() call of
    // 1,024 items is 16 tasks of 64 = 16 waves with 8 KiB in flight each, too little to keep a PCIe link (or HBM) busy -- is cut into more,
    // shorter tasks (the EVD then runs on fewer lanes: its latency is what it was).  Items are independent: no result depends on it.
    constexpr int RSD = 34;                       // see cov4_x4_kernel
    constexpr int RING = 8;                       // chunk loads in flight per wave (8 KiB)
    __shared__ double stage[4][2][8 * RSD];       // per wave, double-buffered
    __shared__ double gram[4][2][64];             // per wave: D1, D2
    __shared__ double rtab[4][16][64];            // per wave: R of 64 items, [slot][item]: 4 diagonals, 6 x (re, im)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t chunks = K >> 5;               // 1-KiB chunks per item (multiple of 8)
    const int wcol = lane >> 1, wrow = 4 * (lane & 1);
    const int ri = lane & 3, rh = (lane >> 2) & 1, rw = (lane >> 3) & 1, rk = lane >> 4;
    const int p_off = (4 * rh + ri) * RSD + 8 * rk + 4 * rw;
    const int q_off = (4 * (1 - rh) + ri) * RSD + 8 * rk + 4 * rw;
    double* const g1 = gram[wave][0];
    double* const g2 = gram[wave][1];
    double(*const rt)[64] = rtab[wave];
    const double dK = (double)K;
    const uint32_t ntasks = (batch + task_items - 1) / task_items;
    const uint32_t tstride = gridDim.x * 4;
    // slot of the upper-triangle entry this lane (< 16: a = lane>>2, b = lane&3) produces: diagonal a -> a;
    // pair (a < b) -> 4 + 2p (re), 5 + 2p (im), p = index of (a, b) in (0,1)(0,2)(0,3)(1,2)(1,3)(2,3)
    const int ea = (lane >> 2) & 3, eb = lane & 3;
    const int pidx = (ea == 0) ? eb - 1 : (ea == 1 ? eb + 1 : 5);

    for (uint32_t task = blockIdx.x * 4 + wave; task < ntasks; task += tstride) {
        __builtin_amdgcn_s_setprio(3);
        const uint32_t item0 = task * task_items;
        const uint32_t nit = (batch - item0 < task_items) ? batch - item0 : task_items;
        // the stream of this task: nit items x chunks, contiguous in HBM; ring slot u holds the chunks q = u (mod 8)
        const v4f32* __restrict__ src = reinterpret_cast<const v4f32*>(in + (size_t)item0 * K * 8) + lane;
        const uint32_t total = nit * chunks;                 // multiple of 8
        v4f32 pf[RING];
#pragma unroll
        for (int u = 0; u < RING; ++u) pf[u] = __builtin_nontemporal_load(src + (size_t)u * 64);
        uint32_t q = 0;                                      // chunk index inside the task
        for (uint32_t it = 0; it < nit; ++it) {
            double a1 = 0.0, b1 = 0.0, a2 = 0.0, b2 = 0.0;
            for (uint32_t cg = 0; cg < chunks; cg += 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    double* __restrict__ T = stage[wave][u & 1];
#pragma unroll
                    for (int j = 0; j < 4; ++j) T[(wrow + j) * RSD + wcol] = (double)pf[u][j];   // exact widening (.cc:77)
                    // re-arm the slot only after its values are consumed (see cov4_x4_kernel); past the end of the
                    // task the loads repeat its last chunk, so that they stay unconditional
                    asm volatile("" ::: "memory");
                    const uint32_t qn = q + u + RING;
                    pf[u] = __builtin_nontemporal_load(src + (size_t)(qn < total ? qn : total - 1) * 64);
                    wave_lds_fence();
                    const v4f64 P = *reinterpret_cast<const v4f64*>(T + p_off);
                    const v4f64 Q = *reinterpret_cast<const v4f64*>(T + q_off);
                    a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(P[0], P[0], a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f64_4x4x4f64(P[0], Q[0], a2, 0, 0, 0);
                    b1 = __builtin_amdgcn_mfma_f64_4x4x4f64(P[1], P[1], b1, 0, 0, 0);
                    b2 = __builtin_amdgcn_mfma_f64_4x4x4f64(P[1], Q[1], b2, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(P[2], P[2], a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f64_4x4x4f64(P[2], Q[2], a2, 0, 0, 0);
                    b1 = __builtin_amdgcn_mfma_f64_4x4x4f64(P[3], P[3], b1, 0, 0, 0);
                    b2 = __builtin_amdgcn_mfma_f64_4x4x4f64(P[3], Q[3], b2, 0, 0, 0);
                    wave_lds_fence();
                }
                q += 8;
            }
            // Gram blocks -> R (see cov4_x4_kernel), upper triangle into the wave's table
            g1[lane] = a1 + b1;
            g2[lane] = a2 + b2;
            wave_lds_fence();
            if (lane < 16) {
                auto G = [&](int x, int y) -> double {
                    if ((x >> 2) == (y >> 2)) {
                        const int hh = x >> 2, o = (y & 3) + 4 * hh + 16 * (x & 3);
                        return g1[o] + g1[o + 8];
                    }
                    if (x > y) { const int t = x; x = y; y = t; }
                    const int o = (y - 4) + 16 * x;
                    return g2[o] + g2[o + 8];
                };
                const double re = (G(2 * ea, 2 * eb) + G(2 * ea + 1, 2 * eb + 1)) / dK;     // .cc:85
                const double im = (G(2 * ea + 1, 2 * eb) - G(2 * ea, 2 * eb + 1)) / dK;
                if (ea == eb) rt[ea][it] = re;
                else if (ea < eb) { rt[4 + 2 * pidx][it] = re; rt[5 + 2 * pidx][it] = im; }
                if (Rdbg) Rdbg[(size_t)(item0 + it) * 16 + lane] = make_double2(re, im);
            }
            wave_lds_fence();
        }
        // EVD of the task's items, one per lane (lanes beyond nit redo the last item and write nothing), at low priority
        __builtin_amdgcn_s_setprio(0);
        {
            const int li = ((uint32_t)lane < nit) ? lane : (int)nit - 1;
            auto getR = [&](int i, int j) -> double2 {
                if (i == j) return make_double2(rt[i][li], 0.0);
                const int lo = i < j ? i : j, hi2 = i < j ? j : i;
                const int p = (lo == 0) ? hi2 - 1 : (lo == 1 ? hi2 + 1 : 5);
                const double re = rt[4 + 2 * p][li], im = rt[5 + 2 * p][li];
                return make_double2(re, i < j ? im : -im);
            };
            evd_project_lane<4>(getR, (uint32_t)lane < nit, item0 + lane, n, qstride, Qs, Gs);
        }
        wave_lds_fence();
    }
}

// -------------------------------------------------------------------------------------
// 2b. The same EVD for m >= 5: M lanes per item, A lives in LDS (complex128), V in registers (one row per lane).
//     A rotation is two lane-parallel phases: lane j applies the column operation to ITS ROW of A and V
//     (A J, V J), then -- after a wave-level LDS hand-over -- the row operation to ITS COLUMN of A (J^H (A J)).
//     Rotations run in tournament rounds of ME/2 disjoint pairs.  Compact code (runtime rounds) instead of a
//     >64 KiB fully unrolled register kernel or the private-array (scratch) form, which took 1.17 ms per 4,096
//     8x8 items.
// -------------------------------------------------------------------------------------
// position -> position map of the tournament movement between two rounds (Brent-Luk "musical chairs"): the pairs of
// a round are always the positions (2k, 2k+1); position 0 stays, the even positions move up, the odd ones down.
// new[pos] = old[tour_src(pos)].
template <int ME>
__device__ __host__ constexpr int tour_src(int pos)
{
    if (ME <= 2 || pos == 0) return pos;
    if (pos == 2) return 1;
    if (pos == ME - 1) return ME - 2;
    return (pos & 1) ? pos + 2 : pos - 2;
}

// original index sitting at position pos in round r (closed form of the movement above; r is wave-uniform and pos a
// compile-time constant at every use, so this is scalar-ALU work): the positions 1,2,4,..,ME-2,ME-1,ME-3,..,3 form one
// cycle along which the contents advance by one step per round.
template <int ME>
__device__ __forceinline__ int tour_idx(int r, int pos)
{
    if (ME <= 2 || pos == 0) return pos;
    constexpr int P = ME - 1;
    const int t0 = (pos == 1) ? 0 : ((pos & 1) ? P - (pos - 1) / 2 : pos / 2);
    int t = t0 - r;
    t += (t < 0) ? P : 0;
    return t == 0 ? 1 : (t <= ME / 2 - 1 ? 2 * t : 2 * (P - t) + 1);
}

template <int M>
__global__ __launch_bounds__(64) void evd_proj_lds_kernel(const double2* __restrict__ R,
                                                           double* __restrict__ Qs,
                                                           uint32_t batch, uint32_t n, uint32_t qstride,
                                                           double* __restrict__ Gs,
                                                           const uint8_t* __restrict__ only = nullptr,
                                                           double* __restrict__ Ss = nullptr)
{
    constexpr int MM = M * M;
    constexpr int IPW = 64 / M;           // items per wave
    constexpr int MAX_SWEEPS = 24;
    constexpr int ME = M + (M & 1);       // even size of the round-robin schedule (odd M: one phantom index)
    // A lives in LDS (index space, dynamically addressed); V stays in REGISTERS: lane j holds row j of V with its
    // columns kept in tournament-position order, so the column pair of round-pair k is always registers 2k, 2k+1
    // (static), and is re-ordered between rounds by register moves.  V enters LDS only for the final projector
    // (it reuses A's storage).  Halving the LDS footprint doubles the resident waves at m >= 9.
    __shared__ double2 sA[IPW][M][M + 1]; // +1: rows of different lanes start on different banks
    __shared__ double sPart[IPW][M];
    __shared__ double sPar[IPW][ME / 2][6];
    __shared__ int sSel[IPW][M];          // eigenvalue index by ascending rank

    const int lane = threadIdx.x;
    const int slot = lane / M;            // item within the wave
    const int j = lane - slot * M;        // this lane's row (phase 1) / column (phase 2)
    const bool lane_used = slot < IPW;
    const int sl = lane_used ? slot : 0;
    const uint32_t item = blockIdx.x * IPW + sl;
    const uint32_t itc = (item < batch) ? item : (batch - 1);
    // `only` (the pass behind evd_sub_kernel): just the items that kernel handed back; a wave with none of them leaves
    const bool wanted = !only || only[itc] != 0;
    if (only && !__any(wanted && lane_used && item < batch)) return;
    const bool valid = lane_used && item < batch && wanted;
    double2(*A)[M + 1] = sA[sl];

    double2 Vrow[ME];
#pragma unroll
    for (int k = 0; k < ME; ++k) Vrow[k] = make_double2(k == j ? 1.0 : 0.0, 0.0);
    if (lane_used) {
        const double2* Rp = R + (size_t)itc * MM + j * M;
        double rsum = 0.0;
#pragma unroll
        for (int k = 0; k < M; ++k) {
            double2 v = Rp[k];
            rsum += v.x + v.y;
            if (k == j) v.y = 0.0;
            A[j][k] = v;
        }
        sPart[sl][j] = rsum * 0.0;      // NaN iff this row holds a NaN / Inf
    }
    wave_lds_fence();
    // non-finite covariance -> poisoned projector (see evd_proj_kernel)
    double poison = 0.0;
#pragma unroll
    for (int k = 0; k < M; ++k) poison += sPart[sl][k];
    // exact power-of-two normalisation (see evd_proj_kernel)
    double dmax = 0.0;
#pragma unroll
    for (int k = 0; k < M; ++k) dmax = fmax(dmax, fabs(A[k][k].x));
    int ex = 0;
    (void)frexp(dmax, &ex);
    const double scl = (dmax > 0.0 && dmax < __builtin_huge_val()) ? ldexp(1.0, -ex) : 1.0;
    wave_lds_fence();
    if (lane_used) {
#pragma unroll
        for (int k = 0; k < M; ++k) {
            double2 v = A[j][k];
            v.x *= scl; v.y *= scl;
            A[j][k] = v;
        }
    }
    wave_lds_fence();

    for (int sweep = 0; sweep < MAX_SWEEPS; ++sweep) {
        double off = 0.0;
        if (lane_used) {
#pragma unroll
            for (int k = 0; k < M; ++k) {
                const double2 v = A[j][k];
                if (k != j) off += v.x * v.x + v.y * v.y;
            }
            sPart[sl][j] = off;
        }
        wave_lds_fence();
        double offsum = 0.0, dia = 0.0;
#pragma unroll
        for (int k = 0; k < M; ++k) { offsum += sPart[sl][k]; const double a = A[k][k].x; dia += a * a; }
        const bool done = !(offsum > 2e-33 * dia);   // offsum counts every off-diagonal twice
        wave_lds_fence();
        if (__all(done || !lane_used)) break;

        // One sweep = ME-1 rounds of the round-robin (tournament) ordering; the <= ME/2 pairs of a round are disjoint,
        // so their rotations commute and read only their own 2x2 block: parameters of all pairs are computed at once
        // (lane k of the item takes pair k), then every lane applies ALL column operations of the round to its row
        // of A (LDS) and V (registers), then ALL row operations to its column of A.  3 LDS hand-overs per round
        // instead of 2 per rotation, one parameter evaluation per lane per round instead of one per lane per
        // rotation.  (Row-cyclic form: 2.05 ms per 16,384 16x16 items, 57 % of the config-5 step.)
        for (int r = 0; r < ME - 1; ++r) {
            if (lane_used && j < ME / 2) {
                const int k = j;
                const int pp = tour_idx<ME>(r, 2 * k), qq = tour_idx<ME>(r, 2 * k + 1);
                double c = 1.0, sn = 0.0, ur = 1.0, ui = 0.0;
                if (pp < M && qq < M) {                  // (a pair with the phantom index of an odd M idles)
                    const double2 apq = A[pp][qq];
                    const double app = A[pp][pp].x, aqq = A[qq][qq].x;
                    const double g2 = apq.x * apq.x + apq.y * apq.y;
                    const bool rot = !done && g2 > 1e-40;   // a converged item freezes (exact identity) while wave-mates sweep
                    const double gg = sqrt(g2);
                    const double ig = rot ? 1.0 / gg : 0.0;
                    ur = rot ? apq.x * ig : 1.0;
                    ui = rot ? apq.y * ig : 0.0;
                    const double tau = (aqq - app) * 0.5 * ig;
                    double t = copysign(1.0, tau) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    t = rot ? t : 0.0;
                    c = 1.0 / sqrt(1.0 + t * t);
                    sn = t * c;
                }
                double* par = sPar[sl][k];
                par[0] = c; par[1] = sn; par[2] = sn * ur; par[3] = sn * ui; par[4] = c * ur; par[5] = c * ui;
            }
            wave_lds_fence();
            // phase 1: this lane's row j of A and V, columns p_k and q_k of every pair  (A J, V J).  All operands of
            // the round are fetched before the first result is stored (the pairs touch disjoint columns, which the
            // compiler cannot know): one LDS round trip per phase instead of one per pair.
            if (lane_used) {
                constexpr int HB = (ME / 2 + 1) / 2;          // two operand batches: bounds the live registers
#pragma unroll
                for (int h = 0; h < ME / 2; h += HB) {
                    double2 ax[HB], ay[HB];
#pragma unroll
                    for (int kk = 0; kk < HB; ++kk) {
                        const int k = h + kk;
                        if (k < ME / 2) {
                            const int pp = tour_idx<ME>(r, 2 * k), qq = tour_idx<ME>(r, 2 * k + 1);
                            if (pp < M && qq < M) { ax[kk] = A[j][pp]; ay[kk] = A[j][qq]; }
                        }
                    }
#pragma unroll
                    for (int kk = 0; kk < HB; ++kk) {
                        const int k = h + kk;
                        if (k >= ME / 2) continue;
                        const int pp = tour_idx<ME>(r, 2 * k), qq = tour_idx<ME>(r, 2 * k + 1);
                        if (pp >= M || qq >= M) continue;
                        const double* par = sPar[sl][k];
                        const double c = par[0], s = par[1], sur = par[2], sui = par[3], cur = par[4], cui = par[5];
                        const double2 x = ax[kk], y = ay[kk], vx = Vrow[2 * k], vy = Vrow[2 * k + 1];
                        A[j][pp] = make_double2(c * x.x - (sur * y.x + sui * y.y), c * x.y - (sur * y.y - sui * y.x));
                        A[j][qq] = make_double2(s * x.x + (cur * y.x + cui * y.y), s * x.y + (cur * y.y - cui * y.x));
                        Vrow[2 * k] = make_double2(c * vx.x - (sur * vy.x + sui * vy.y), c * vx.y - (sur * vy.y - sui * vy.x));
                        Vrow[2 * k + 1] = make_double2(s * vx.x + (cur * vy.x + cui * vy.y), s * vx.y + (cur * vy.y - cui * vy.x));
                    }
                }
            }
            wave_lds_fence();
            // phase 2: this lane's column j of A, rows p_k and q_k of every pair  (J^H (A J))
            if (lane_used) {
                double2 ax[ME / 2], ay[ME / 2];
#pragma unroll
                for (int k = 0; k < ME / 2; ++k) {
                    const int pp = tour_idx<ME>(r, 2 * k), qq = tour_idx<ME>(r, 2 * k + 1);
                    if (pp < M && qq < M) { ax[k] = A[pp][j]; ay[k] = A[qq][j]; }
                }
#pragma unroll
                for (int k = 0; k < ME / 2; ++k) {
                    const int pp = tour_idx<ME>(r, 2 * k), qq = tour_idx<ME>(r, 2 * k + 1);
                    if (pp >= M || qq >= M) continue;
                    const double* par = sPar[sl][k];
                    const double c = par[0], s = par[1], sur = par[2], sui = par[3], cur = par[4], cui = par[5];
                    const double2 x = ax[k], y = ay[k];
                    double2 np = make_double2(c * x.x - (sur * y.x - sui * y.y), c * x.y - (sur * y.y + sui * y.x));
                    double2 nq = make_double2(s * x.x + (cur * y.x - cui * y.y), s * x.y + (cur * y.y + cui * y.x));
                    if (j == qq) np = make_double2(0.0, 0.0);                    // a_pq := 0
                    if (j == pp) { nq = make_double2(0.0, 0.0); np.y = 0.0; }    // a_qp := 0, real diagonal
                    if (j == qq) nq.y = 0.0;
                    A[pp][j] = np;
                    A[qq][j] = nq;
                }
            }
            wave_lds_fence();
            // tournament movement of V's columns (registers): position pos now holds what tour_src(pos) held
            {
                double2 t[ME];
#pragma unroll
                for (int k = 0; k < ME; ++k) t[k] = Vrow[tour_src<ME>(k)];
#pragma unroll
                for (int k = 0; k < ME; ++k) Vrow[k] = t[k];
            }
        }
    }
    // (sweeps are whole periods of the tournament: position == original index again)

    // Ascending rank of the eigenvalues (ties -> lower column first; the noise space is rank < m-n, .cc:93): lane j
    // ranks eigenvalue j and publishes sSel[rank] = j.  The projector is then summed over the SMALLER of the two sets:
    // Q = sum_noise v v^H  or  Q = I - sum_signal v v^H  (V is unitary) -- n = 2 of 16 columns at config 5, which
    // takes the epilogue from ~m^3/2 to ~m^2 n complex MACs per item.
    if (lane_used) sSel[sl][j] = j;
    wave_lds_fence();
    if (lane_used) {
        const double wj = A[j][j].x;
        int rank = 0;
#pragma unroll
        for (int l = 0; l < M; ++l) {
            const double wl = A[l][l].x;
            rank += (wl < wj || (wl == wj && l < j)) ? 1 : 0;
        }
        sSel[sl][rank] = j;                    // (NaN eigenvalues: every rank is 0; the projector is poisoned anyway)
    }
    const int nnoise = (int)M - (int)n;
    const bool use_noise = nnoise <= (int)n;
    const int cnt = use_noise ? nnoise : (int)n;
    const int base = use_noise ? 0 : nnoise;
    // V rows go to LDS (A's storage) for the cross-lane projector
    wave_lds_fence();
    double2(*V)[M + 1] = sA[sl];
    if (lane_used) {
#pragma unroll
        for (int k = 0; k < M; ++k) V[j][k] = Vrow[k];
    }
    wave_lds_fence();
    // the noise eigenvectors themselves (see evd_proj_kernel): lane j writes component j of every noise vector
    if (valid && Gs) {
        for (int r = 0; r < nnoise; ++r) {
            const double2 v = V[j][sSel[sl][r] & 15];
            Gs[(size_t)((r * M + j) * 2) * qstride + item] = v.x;
            Gs[(size_t)((r * M + j) * 2 + 1) * qstride + item] = v.y;
        }
    }
    // the two signal eigenvectors as the coefficient vectors of the scan's short form (scan_mfma_kernel, SIG): output
    // 2c = Re s_c^H a, 2c+1 = Im s_c^H a over the real coordinates (re a_0, im a_0, re a_1, ...)
    if (valid && Ss && n <= 2) {
        for (int cI = 0; cI < (int)n; ++cI) {
            const double2 v = V[j][sSel[sl][nnoise + cI] & 15];
            const double vr = v.x + poison, vi = v.y + poison;
            Ss[(size_t)((2 * cI) * 2 * M + 2 * j) * qstride + item] = vr;
            Ss[(size_t)((2 * cI) * 2 * M + 2 * j + 1) * qstride + item] = vi;
            Ss[(size_t)((2 * cI + 1) * 2 * M + 2 * j) * qstride + item] = -vi;
            Ss[(size_t)((2 * cI + 1) * 2 * M + 2 * j + 1) * qstride + item] = vr;
        }
    }
    // lane j emits row j of Q (upper part): Q_jl = sum_{k in set} V[j][k] conj(V[l][k])
    // (Qs == nullptr: the scan runs the short form from Ss and never reads the projector)
    if (valid && Qs) {
        for (int l = j; l < M; ++l) {
            double re = 0.0, im = 0.0;
            for (int i = 0; i < cnt; ++i) {
                const int k = sSel[sl][base + i] & 15;
                const double2 vj = V[j][k], vl = V[l][k];
                re += vj.x * vl.x + vj.y * vl.y;
                im += vj.y * vl.x - vj.x * vl.y;
            }
            if (!use_noise) { re = ((l == j) ? 1.0 : 0.0) - re; im = -im; }
            if (l == j) {
                Qs[(size_t)(j * M + j) * qstride + item] = re + poison;
            } else {
                Qs[(size_t)(j * M + l) * qstride + item] = 2.0 * re + poison;
                Qs[(size_t)(l * M + j) * qstride + item] = -2.0 * im + poison;
            }
        }
    }
}

// -------------------------------------------------------------------------------------
// 2c. The projector WITHOUT the full eigen-decomposition, for m >= 5 and few emitters (n = P <= 4, 2P <= m).
//
//     MUSIC needs only the noise-subspace projector I - S S^H, S = the eigenvectors of the n LARGEST eigenvalues
//     (.cc:88-93 keeps the other m-n columns of eig_sym), and with n of m = 2 of 16 that invariant subspace is found far
//     faster by orthogonal (subspace) iteration than all 16 eigenpairs by Jacobi:
//         Y <- orth(R Y)      convergence: sin(angle to the invariant subspace) shrinks by lambda_{n+1} / lambda_n per step
//     (5-6 steps at 20 dB SNR, 8 at 10 dB, 15 at 0 dB for config 5's shape; ~450 instructions per step against ~43,000
//     for the Jacobi of a 16x16 matrix: config 5's EVD 0.98 -> see DESIGN.md).  An item is DONE when the change
//     || (I - Y Y^H) Y' ||_F of its basis is <= 4e-15 sqrt(n) (projector within ~1e-15 of the eigh-based one, measured);
//     it then freezes (exact) while wave-mates continue, so results never depend on which items share a wave.  An item
//     that does not get there -- a gap lambda_n ~ lambda_{n+1} (fewer emitters than n, noise only, zero input), a
//     non-finite R -- is handed back through `redo` and takes the Jacobi (evd_proj_lds_kernel with only = redo): the
//     choice is a function of the item alone.
//     Layout: GS = 8 or 16 lanes per item (lane j = row j of R and of Y; lanes >= m idle), all reductions over the
//     group are XOR butterflies on DPP (mirror / half-mirror / quad permutes: every lane of the group ends with the
//     SAME bits, each step adds the same two operands in both partners), Y is published through LDS for the R Y product.
//     G (the noise eigenvectors' stand-in for the scan's literal form -- any orthonormal basis of the noise subspace
//     gives the same ||G^H a||^2): the last m-n columns of the Householder completion H_0 .. H_{P-1} of S.
// -------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)b, CTRL, 0xF, 0xF, true);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_mov_dpp((int)(uint32_t)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

template <int GS>
__device__ __forceinline__ double group_allsum(double v)
{
    if constexpr (GS == 16) v += dpp_f64<0x140>(v);     // row_mirror       lane ^ 15
    v += dpp_f64<0x141>(v);                             // row_half_mirror  lane ^ 7
    v += dpp_f64<0x1B>(v);                              // quad_perm [3,2,1,0]  lane ^ 3
    v += dpp_f64<0xB1>(v);                              // quad_perm [1,0,3,2]  lane ^ 1
    return v;
}

template <int GS>
__device__ __forceinline__ double group_allmax(double v)
{
    if constexpr (GS == 16) v = fmax(v, dpp_f64<0x140>(v));
    v = fmax(v, dpp_f64<0x141>(v));
    v = fmax(v, dpp_f64<0x1B>(v));
    v = fmax(v, dpp_f64<0xB1>(v));
    return v;
}

__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 cmulc(double2 a, double2 b) { return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }   // a conj(b)

template <int M, int P>
__global__ __launch_bounds__(64) void evd_sub_kernel(const double2* __restrict__ R, double* __restrict__ Qs, uint32_t batch,
                                                      uint32_t qstride, double* __restrict__ Gs, uint8_t* __restrict__ redo,
                                                      double* __restrict__ Ss = nullptr)
{
    static_assert(M >= 5 && M <= 16 && P >= 1 && 2 * P <= M, "few emitters on the LDS-EVD antenna counts");
    constexpr int GS = M <= 8 ? 8 : 16, IPW = 64 / GS, MM = M * M;
    // give up where the Jacobi is cheaper: ~35 steps' worth at m <= 8, ~95 at m = 16
    constexpr int MAX_IT = M <= 8 ? 28 : 64;
    constexpr double BAIL2 = M <= 8 ? 0.09 : 0.36;        // (estimated rate)^2 beyond which MAX_IT cannot be met
    constexpr double TOL2 = 1.6e-29 * P;                  // (4e-15 sqrt(P))^2
    __shared__ double2 sY[IPW][P][GS];
    const int lane = threadIdx.x, g = lane / GS, j = lane - g * GS;
    const uint32_t item = blockIdx.x * IPW + g;
    const bool item_ok = item < batch, row = j < M;
    const uint32_t itc = item_ok ? item : batch - 1;
    const int jc = row ? j : 0;

    double2 Rrow[M];
    double psum = 0.0, dj = 0.0;
#pragma unroll
    for (int k = 0; k < M; ++k) {
        double2 v = R[(size_t)itc * MM + jc * M + k];
        if (!row) v = make_double2(0.0, 0.0);
        psum += v.x + v.y;
        if (k == jc) { v.y = 0.0; dj = fabs(v.x); }
        Rrow[k] = v;
    }
    const double poison = group_allsum<GS>(psum * 0.0);   // NaN iff R holds a NaN / Inf: such an item goes to the Jacobi
    const double dmax = group_allmax<GS>(dj);
    int ex = 0;
    (void)frexp(dmax, &ex);
    const double scl = (dmax > 0.0 && dmax < __builtin_huge_val()) ? ldexp(1.0, -ex) : 1.0;   // as evd_proj_kernel
#pragma unroll
    for (int k = 0; k < M; ++k) { Rrow[k].x *= scl; Rrow[k].y *= scl; }

    // modified Gram-Schmidt, each projection twice; ok = every norm was a positive finite number
    auto orth = [&](double2 (&z)[P], double2 (&yn)[P]) -> bool {
        bool ok = true;
#pragma unroll
        for (int c = 0; c < P; ++c) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int c2 = 0; c2 < c; ++c2) {
                    const double2 t = cmulc(z[c], yn[c2]);            // conj(yn) z, this row's term
                    const double hr = group_allsum<GS>(t.x), hi = group_allsum<GS>(t.y);
                    z[c].x -= hr * yn[c2].x - hi * yn[c2].y;
                    z[c].y -= hr * yn[c2].y + hi * yn[c2].x;
                }
            const double n2 = group_allsum<GS>(z[c].x * z[c].x + z[c].y * z[c].y);
            const bool good = n2 > 0.0 && n2 < __builtin_huge_val();
            ok = ok && good;
            const double inv = good ? 1.0 / sqrt(n2) : 0.0;
            yn[c] = make_double2(z[c].x * inv, z[c].y * inv);
        }
        return ok;
    };

    double2 y[P], z[P];
#pragma unroll
    for (int c = 0; c < P; ++c) z[c] = Rrow[c];            // one step from the first P unit vectors
    bool ok = orth(z, y) && !(poison != poison);
    bool conv = false;
    double d2prev = __builtin_huge_val();
    for (int it = 0; it < MAX_IT; ++it) {
        wave_lds_fence();
#pragma unroll
        for (int c = 0; c < P; ++c) sY[g][c][j] = y[c];
        wave_lds_fence();
#pragma unroll
        for (int c = 0; c < P; ++c) z[c] = make_double2(0.0, 0.0);
#pragma unroll
        for (int k = 0; k < M; ++k)
#pragma unroll
            for (int c = 0; c < P; ++c) {
                const double2 yk = sY[g][c][k];
                z[c].x += Rrow[k].x * yk.x - Rrow[k].y * yk.y;
                z[c].y += Rrow[k].x * yk.y + Rrow[k].y * yk.x;
            }
        double2 yn[P];
        const bool ok2 = orth(z, yn);
        // D = Y' - Y (Y^H Y'): the part of the new basis outside the old subspace
        double dloc = 0.0;
#pragma unroll
        for (int b = 0; b < P; ++b) {
            double2 d = yn[b];
#pragma unroll
            for (int a = 0; a < P; ++a) {
                const double2 t = cmulc(yn[b], y[a]);                 // conj(y_a) yn_b
                const double cr = group_allsum<GS>(t.x), ci = group_allsum<GS>(t.y);
                d.x -= y[a].x * cr - y[a].y * ci;
                d.y -= y[a].x * ci + y[a].y * cr;
            }
            dloc += d.x * d.x + d.y * d.y;
        }
        const double d2 = group_allsum<GS>(dloc);
        if (!conv && ok) {                                  // (a frozen or failed item keeps its y)
            ok = ok2 && (d2 == d2);
#pragma unroll
            for (int c = 0; c < P; ++c) y[c] = yn[c];
            if (ok && d2 <= TOL2) conv = true;
            else if (it >= 2 && d2 > 100.0 * TOL2 && d2 > BAIL2 * d2prev) ok = false;   // too slow: the Jacobi is cheaper
            d2prev = d2;
        }
        if (__all(conv || !ok || !item_ok)) break;
    }
    if (item_ok && j == 0) redo[item] = conv ? 0 : 1;
    if (!__any(conv && item_ok)) return;

    // ---- outputs of the converged items ----
    wave_lds_fence();
#pragma unroll
    for (int c = 0; c < P; ++c) sY[g][c][j] = y[c];
    wave_lds_fence();
    const bool emit = conv && item_ok && row;
    if constexpr (P <= 2) {
        if (emit && Ss) {      // coefficient vectors of the scan's short form (see evd_proj_lds_kernel)
#pragma unroll
            for (int cI = 0; cI < P; ++cI) {
                Ss[(size_t)((2 * cI) * 2 * M + 2 * j) * qstride + item] = y[cI].x;
                Ss[(size_t)((2 * cI) * 2 * M + 2 * j + 1) * qstride + item] = y[cI].y;
                Ss[(size_t)((2 * cI + 1) * 2 * M + 2 * j) * qstride + item] = -y[cI].y;
                Ss[(size_t)((2 * cI + 1) * 2 * M + 2 * j + 1) * qstride + item] = y[cI].x;
            }
        }
    }
    if (emit && Qs) {   // row j of Q = I - S S^H (upper part), packed as evd_proj_kernel does (not needed by the short-form scan)
        for (int l = j; l < M; ++l) {
            double re = 0.0, im = 0.0;
#pragma unroll
            for (int c = 0; c < P; ++c) {
                const double2 vl = sY[g][c][l];
                re += y[c].x * vl.x + y[c].y * vl.y;
                im += y[c].y * vl.x - y[c].x * vl.y;
            }
            re = ((l == j) ? 1.0 : 0.0) - re; im = -im;
            if (l == j) {
                Qs[(size_t)(j * M + j) * qstride + item] = re;
            } else {
                Qs[(size_t)(j * M + l) * qstride + item] = 2.0 * re;
                Qs[(size_t)(l * M + j) * qstride + item] = -2.0 * im;
            }
        }
    }
    if (!Gs) return;
    // ---- Householder completion: S = H_0 .. H_{P-1} [I_P; 0] D  ->  columns P..M-1 of H_0 .. H_{P-1} span the noise space
    double2 v[P];
    double tau[P];
    double2 w[P];
#pragma unroll
    for (int c = 0; c < P; ++c) w[c] = y[c];
#pragma unroll
    for (int c = 0; c < P; ++c) {
        // x = w[c] on rows >= c; v = x - alpha e_c, alpha = -phase(x_c) ||x||
        const bool in = row && j >= c;
        const double2 x = in ? w[c] : make_double2(0.0, 0.0);
        const double nx2 = group_allsum<GS>(x.x * x.x + x.y * x.y);
        const double xcr = group_allsum<GS>(j == c ? x.x : 0.0), xci = group_allsum<GS>(j == c ? x.y : 0.0);
        const double nx = sqrt(nx2), ax = sqrt(xcr * xcr + xci * xci);
        const double pr = ax > 0.0 ? xcr / ax : 1.0, pi = ax > 0.0 ? xci / ax : 0.0;
        v[c] = x;
        if (j == c) { v[c].x += pr * nx; v[c].y += pi * nx; }
        const double nv2 = group_allsum<GS>(v[c].x * v[c].x + v[c].y * v[c].y);
        tau[c] = nv2 > 0.0 ? 2.0 / nv2 : 0.0;
#pragma unroll
        for (int b = c + 1; b < P; ++b) {                  // w_b <- H_c w_b
            const double2 t = cmulc(w[b], v[c]);           // conj(v) w_b, this row's term
            const double sr = group_allsum<GS>(t.x) * tau[c], si = group_allsum<GS>(t.y) * tau[c];
            w[b].x -= v[c].x * sr - v[c].y * si;
            w[b].y -= v[c].x * si + v[c].y * sr;
        }
    }
    // beta[c][c2] = v_c^H v_c2 (c < c2)
    double2 beta[P][P];
#pragma unroll
    for (int c = 0; c < P; ++c)
#pragma unroll
        for (int c2 = c + 1; c2 < P; ++c2) {
            const double2 t = cmulc(v[c2], v[c]);
            beta[c][c2] = make_double2(group_allsum<GS>(t.x), group_allsum<GS>(t.y));
        }
    wave_lds_fence();
#pragma unroll
    for (int c = 0; c < P; ++c) sY[g][c][j] = v[c];
    wave_lds_fence();
    if (emit) {
        for (int k = P; k < M; ++k) {                      // g_k = H_0 .. H_{P-1} e_k = e_k - sum_c coef_c v_c
            double2 coef[P];
            double2 gk = make_double2(j == k ? 1.0 : 0.0, 0.0);
#pragma unroll
            for (int c = P - 1; c >= 0; --c) {
                const double2 vk = sY[g][c][k];
                double2 s = make_double2(vk.x, -vk.y);     // v_c^H e_k
#pragma unroll
                for (int c2 = c + 1; c2 < P; ++c2) {       // - coef_c2 (v_c^H v_c2)
                    const double2 t = cmul(coef[c2], beta[c][c2]);
                    s.x -= t.x; s.y -= t.y;
                }
                coef[c] = make_double2(tau[c] * s.x, tau[c] * s.y);
                const double2 t = cmul(coef[c], v[c]);
                gk.x -= t.x; gk.y -= t.y;
            }
            const int r = k - P;
            Gs[(size_t)((r * M + j) * 2) * qstride + item] = gk.x;
            Gs[(size_t)((r * M + j) * 2 + 1) * qstride + item] = gk.y;
        }
    }
}

// =====================================================================================
// 3. Top-n bookkeeping (lib/baz_music_doa.cc:95,129-141) on packed fp64 keys.
//
//    key = bits(d) with the low `binbits` mantissa bits replaced by the bin index, d = ||G^H a||^2 >= 0.
//    For non-negative doubles the IEEE order is the integer order of the bits, so keys order by
//    (d truncated by <= 2^-36 relative, then bin): the n SMALLEST keys are the reference's n largest
//    strengths with the earliest bin winning ties -- exactly the outcome of its strict-">" insertion
//    over ascending bins -- except between bins whose d agree to 1.5e-11 relative (the parity rule
//    tolerates 2e-5).  Lists are kept ascending with a branch-free v_min_f64 / v_max_f64 network
//    (2*NMAX-1 instructions per candidate, no divergence, no data-dependent slow path).
//    Non-finite d (NaN / +inf -> strength 0) form NaN keys, which min/max drop: never inserted, like
//    the reference.  EMPTY marks an unused slot = (angle 0, strength 0).
// =====================================================================================
#define BAZ_KEY_EMPTY_BITS 0x7FEFFFFFFFF00000ull   /* ~DBL_MAX, low 20 bits clear */

__device__ __forceinline__ double key_empty() { return __builtin_bit_cast(double, (uint64_t)BAZ_KEY_EMPTY_BITS); }

__device__ __forceinline__ double vmin64(double a, double b)
{
    double r;   // raw v_min_f64: no canonicalisation pass over the hand-built keys
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double vmax64(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// key of a freshly computed d: only the low word is touched (one v_and_or_b32); the sign of d (rounding noise
// around 0) is dropped by the |.| source modifier of the first network operation in key_insert_new().
__device__ __forceinline__ double make_key(const double d, const uint32_t bin, const uint32_t keep_mask)
{
    const uint64_t b = __builtin_bit_cast(uint64_t, d);
    const uint32_t lo = ((uint32_t)b & keep_mask) | bin;
    return __builtin_bit_cast(double, (b & 0xFFFFFFFF00000000ull) | lo);
}

template <int NMAX>
__device__ __forceinline__ void key_insert(double (&t)[NMAX], double k)
{
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
        const double lo = vmin64(k, t[i]);
        if (i + 1 < NMAX) k = vmax64(k, t[i]);
        t[i] = lo;
    }
}

// same, for a key that may still carry a sign bit: |k| is applied inside the first min/max pair
template <int NMAX>
__device__ __forceinline__ void key_insert_new(double (&t)[NMAX], double k)
{
    double lo, hi = 0.0;
    asm("v_min_f64 %0, |%1|, %2" : "=v"(lo) : "v"(k), "v"(t[0]));
    if constexpr (NMAX > 1) asm("v_max_f64 %0, |%1|, %2" : "=v"(hi) : "v"(k), "v"(t[0]));
    t[0] = lo;
    if constexpr (NMAX > 1) {
        k = hi;
#pragma unroll
        for (int i = 1; i < NMAX; ++i) {
            const double l2 = vmin64(k, t[i]);
            if (i + 1 < NMAX) k = vmax64(k, t[i]);
            t[i] = l2;
        }
    }
}

template <int NMAX>
__device__ __forceinline__ void key_merge_xor(double (&t)[NMAX], const int mask)
{
    double o[NMAX];
#pragma unroll
    for (int i = 0; i < NMAX; ++i) o[i] = __shfl_xor(t[i], mask, 64);
#pragma unroll
    for (int i = 0; i < NMAX; ++i) key_insert<NMAX>(t, o[i]);
}

__device__ __forceinline__ float strength_f32(const double d)
{
    // (float)(1.0/d) of the reference (.cc:114-121) evaluated as rcp_f32((float)d): <= ~2 ulp_f32
    // (2.4e-7 relative) from the correctly rounded value; the parity budget is 1e-5.
    return __builtin_amdgcn_rcpf((float)d);
}

// =====================================================================================
// 4. Pseudo-spectrum scan as an fp64 MFMA GEMM  D[items x bins] = Q[items x MM] * F^T[MM x bins].
//
//    Measured on gfx950 (scripts/ubench*.hip, profiles/): v_mfma_f64_16x16x4_f64 retires 1024 FMAs in
//    65 cycles/SIMD (77 TFLOP/s), v_fma_f64 64 FMAs in 4.9; NO other VALU work overlaps with the fp64
//    MFMA (SIMD time = MFMA cycles + VALU cycles).  What the matrix core buys is operand delivery:
//    the shared table F is a lane-distributed B operand, not a wave-uniform scalar (a lane=item
//    v_fma_f64 form fed by s_load spent 72 % of its wave cycles waiting on scalar-cache misses).
//
//    A wave owns 16 items and a contiguous range of 64-bin steps.  Per step: A = q (16 items x 4 k,
//    held in registers for the whole range), B_t = F for 4 bin tiles t = 0..3 whose COLUMNS are
//    permuted on the host (build_FB) so that tile t, column c is bin 64*step + 4c + t.  The fp64 MFMA
//    accumulator layout (col = lane&15, row = (lane>>4) + 4*reg) then gives lane (g, c), register r,
//    tile t  <->  row g + 4r, bin 64*step + 4c + t: a lane holds 4 CONSECUTIVE bins of one item in
//    acc[0..3][r] -> one 16-B store, and the 16 lanes c = 0..15 of a row write 256 B contiguous.
//
//    ROW CLASSES (round 2).  The spectrum port's layout is [item][res] float32 (the GNU Radio buffer): row i starts
//    at byte 4*res*i, which is a multiple of 256 only when res % 64 == 0.  At cfg2 (res = 3600) three rows in
//    four start 64 / 128 / 192 B into a 256-B window, so a step's 256-B pieces straddle 128-B lines and every
//    line is written in two halves by two steps ~1,500 cycles apart: measured 4.4 TB/s for that pattern against
//    5.8-6.0 TB/s for line-aligned rows (scripts/ubench_hbm.hip, profiles/r02_ubench_hbm.txt) -- the round-1
//    scan sat exactly on the lower figure.  Rows are therefore processed BY CLASS: class k = the rows
//    i = nclass*j + k, nclass = 64 / gcd(res, 64), which all share the shift sh = (res*k) % 64 bins, and a wave's
//    step st covers the bins [64*st - sh, 64*st - sh + 64) -- for every row of the class that is one 256-B ALIGNED
//    window.  A wave's 16 rows are 16 consecutive j of one class, the 4 waves of a block share the class (they
//    share the F slices), and the shifted slice is assembled by the stage LOADS: lane (g, c) fetches column
//    c - sh/4 of step st, or column c - sh/4 + 16 of step st - 1 (FB carries one padded step in front and
//    behind).  LDS layout, MFMA loop and epilogue are unchanged; bins outside [0, res) (first / last step of
//    a row) are masked in the stores and carry a never-selected key.
//
//    F delivery: the 4 waves of a workgroup walk the SAME range of bin steps in lockstep and share every piece of
//    FB through a double-buffered LDS stage (each wave fetches a quarter of the next phase from L2 while the
//    current phase's MFMAs run; one s_barrier per phase).  Ordering inside a phase: stage-loads(next) ... MFMAs
//    ... epilogue VALU ... s_waitcnt (only the stage loads and the PREVIOUS step's stores are outstanding) ... LDS
//    write ... this step's stores ... barrier: on gfx9-family ISAs loads and stores share vmcnt and complete out
//    of order with each other, so a wait must never sit right behind fresh stores.
//    A phase is up to SCH = 8 k-steps (8 or 16 KiB of LDS per buffer); m = 4 has one phase per step.
//    ABL is a lab-only ablation mask (scripts/scan_lab.hip); product launches use ABL = 0.
//
//    NEAR-NULL TILES (round 2; replaces the item-level refine pass of round 1).  The projector form's terms are
//    O(||a||^2), so fp64 leaves ~m^2 1e-16 ||a||^2 ABSOLUTE error in d, while the reference's sum of squares
//    ||G^H a||^2 (.cc:110-121) stays relatively accurate however small d is.  When some d of a step falls below
//    `refine_below` (~m 1e-8 max||a||^2: SNR >~ 55 dB, a few steps around each emitter) the wave recomputes THAT
//    16-item x 64-bin tile in the reference's literal form on the matrix core -- c_k = g_k^H a as 2(m-n) real
//    planes of a [16 x 2m] x [2m x 64] GEMM over the raw table image TB, d = sum of the squared planes, same
//    accumulator layout -- and the epilogue continues with the refined d: spectrum, keys and lvl stay consistent
//    by construction, nothing is re-read, and ordinary steps pay one compare per row (shared with the top-n gate).
// =====================================================================================
struct ScanRefine {
    const double* Gs;          // noise eigenvectors, [((k*M + i)*2 + re/im) * qstride + item]; nullptr: refinement off
    const double2* TB;         // raw-table B-operand image (build_TB), same step padding and shift rule as FB
    double below;              // |d| <= below -> the step's tile is recomputed in literal form
    unsigned long long* count; // statistic: (item, bin) values recomputed (may be nullptr)
    const double* A2;          // SIG scan only: ||a||^2 per bin, padded like FB (huge outside [0, res)); points at step 0
};

template <int M>
__device__ __forceinline__ uint32_t literal_tile(v4f64 (&acc)[4], const ScanRefine& rf, const v2f64* __restrict__ tbl,
                                             const uint32_t st, const uint32_t itc, const int g,
                                             const uint32_t qstride, const int nn, const uint32_t bin, const uint32_t res,
                                             const bool (&row_ok)[4])
{
    constexpr int KS2 = (2 * M + 3) / 4;      // k-steps over the 2m real coordinates (re a_0, im a_0, re a_1, ...)
    const double* __restrict__ tb1 = reinterpret_cast<const double*>(tbl + (size_t)st * KS2 * 2 * 64);
    uint32_t cnt = 0;
    if constexpr (KS2 <= 2) {
        // Up to 4 antennas (round 4).  With every item of a wave in another scene and 60 dB of SNR a third of the steps of a
        // config-2 scan come here (the nulls of 16 unrelated items are spread over its 57 steps), and the form below -- every
        // operand fetched from L2 in front of the MFMA that uses it, behind a branch (the select on 4 s + g < 2 m made hipcc
        // branch around each load and wait for it), 16 dependent round trips per step, the first of them behind the previous
        // step's spectrum stores (vmcnt retires in issue order) -- cost 11 us per visit.  The item's coefficients do not depend
        // on the bin tile and the table's do not depend on the eigenvector: 2 nn KS2 UNCONDITIONAL loads (clamped index, the
        // select afterwards) in one batch + KS2 per bin tile that holds a near-null value; the same MFMAs on the same operands
        // in the same order: the same bits.
        constexpr int KMAX = M - 1;
        double x0[KMAX][KS2], x1[KMAX][KS2];
        const double sgn1 = (g & 1) ? 1.0 : -1.0;         // Im c: -gi*ar (g even: comp = im) + gr*ai (g odd: comp = re)
        // (the addresses are formed HERE: hoisted out of the scan's step loop they would be 2 nn KS2 register pairs the ordinary
        // steps cannot spare -- spilled, and fetched back one by one with a wait each)
        uint32_t itq = itc, gq = (uint32_t)g;
        asm volatile("" : "+v"(itq), "+v"(gq));
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int kk = k < nn ? k : 0;                // (a valid address also for the eigenvectors that do not exist)
#pragma unroll
            for (int s = 0; s < KS2; ++s) {
                const bool in = 4 * s + (int)gq < 2 * M;
                const uint32_t ant = in ? 2u * (uint32_t)s + (gq >> 1) : 0u;
                const double v0 = rf.Gs[(size_t)(((uint32_t)kk * M + ant) * 2u + (gq & 1u)) * qstride + itq];
                const double v1 = rf.Gs[(size_t)(((uint32_t)kk * M + ant) * 2u + ((gq & 1u) ^ 1u)) * qstride + itq];
                x0[k][s] = in ? v0 : 0.0;
                x1[k][s] = in ? sgn1 * v1 : 0.0;
            }
        }
        // only the bin tiles that hold such a value (a null is a few adjacent bins: with bins 4 c + t in tile t, often not all four)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            bool want = false;
#pragma unroll
            for (int r = 0; r < 4; ++r) want |= (fabs(acc[t][r]) <= rf.below) && (bin + t < res);
            if (!__any(want)) continue;                   // wave-uniform
            double tbv[KS2];
#pragma unroll
            for (int s = 0; s < KS2; ++s) tbv[s] = tb1[(size_t)(t >> 1) * 128 + (t & 1) + (size_t)s * 256];
            v4f64 d = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                if (k < nn) {                              // wave-uniform
                    v4f64 p = {0, 0, 0, 0};
#pragma unroll
                    for (int s = 0; s < KS2; ++s) p = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[k][s], tbv[s], p, 0, 0, 0);
                    d += p * p;
                    p = (v4f64){0, 0, 0, 0};
#pragma unroll
                    for (int s = 0; s < KS2; ++s) p = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[k][s], tbv[s], p, 0, 0, 0);
                    d += p * p;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool redo = (fabs(acc[t][r]) <= rf.below) && (bin + t < res);
                acc[t][r] = redo ? d[r] : acc[t][r];
                cnt += (redo && row_ok[r]) ? 1u : 0u;
            }
        }
        return cnt;
    }
    // 5 antennas and more: one bin tile at a time (the scheduling barriers keep the four chains apart): this rare path must
    // not raise the register count -- hence the occupancy -- of the ordinary steps around it.
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        v4f64 d = {0, 0, 0, 0};
        for (int k = 0; k < nn; ++k) {            // noise eigenvector k (.cc:93,116-119)
#pragma unroll
            for (int part = 0; part < 2; ++part) {   // Re / Im of c_k = sum_i conj(G_ik) a_i
                v4f64 p = {0, 0, 0, 0};
                // Real coordinate e = 4 s + g: antenna e/2 = 2 s + (g >> 1), re (g even) or im (g odd) part of a.
                //   Re c: +gr*ar +gi*ai      Im c: -gi*ar +gr*ai
                // A rolled loop over s for m >= 5: unrolled, its 2 x KS2 loop-invariant operand addresses get hoisted
                // out of the STEP loop of the scan and cost the ordinary steps up to 32 VGPRs (m = 16: spills).
                const int comp = part ? ((g & 1) ^ 1) : (g & 1);
                const double sgn = (part && !(g & 1)) ? -1.0 : 1.0;
                const double* __restrict__ ga = rf.Gs + (size_t)((k * M + (g >> 1)) * 2 + comp) * qstride + itc;
                const double* __restrict__ tb = tb1 + (size_t)(t >> 1) * 128 + (t & 1);
                if constexpr (KS2 <= 2) {
#pragma unroll
                    for (int s = 0; s < KS2; ++s) {
                        const double a = (4 * s + g < 2 * M) ? sgn * ga[(size_t)s * 4 * qstride] : 0.0;
                        p = __builtin_amdgcn_mfma_f64_16x16x4f64(a, tb[(size_t)s * 256], p, 0, 0, 0);
                    }
                } else {
#pragma nounroll
                    for (int s = 0; s < KS2; ++s) {
                        const double a = (4 * s + g < 2 * M) ? sgn * ga[(size_t)s * 4 * qstride] : 0.0;
                        p = __builtin_amdgcn_mfma_f64_16x16x4f64(a, tb[(size_t)s * 256], p, 0, 0, 0);
                    }
                }
                d += p * p;
            }
        }
        // Per VALUE: only a d at or below the threshold is replaced, so what an (item, bin) pair gets never depends on
        // which items share its wave, on its row class or on how a batch was cut.  Bins outside the table keep their
        // (huge) projector value: never stored, never selected.
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool redo = (fabs(acc[t][r]) <= rf.below) && (bin + t < res);
            acc[t][r] = redo ? d[r] : acc[t][r];
            cnt += (redo && row_ok[r]) ? 1u : 0u;       // (rows beyond the batch repeat the last item: not counted)
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return cnt;
}

// Occupancy floors: m <= 4: 4 waves/SIMD = 128 registers; m >= 9 (q alone is m^2/2 registers): 2 waves/SIMD = 256
// registers INCLUDING accumulation registers (without the floor the m = 16 kernel took 256 + 28 and ran one wave per
// SIMD: scan 0.56 -> 0.78 ms on config 5).  What this displaces lives in the rare literal / statistic paths.
//
// SIG (m >= 9, n = 2): the short form of the same quantity.  With S = the two signal eigenvectors,
//     a^H (I - S S^H) a = ||a||^2 - |s_0^H a|^2 - |s_1^H a|^2,
// i.e. FOUR real inner products of length 2m per (item, bin) -- Re/Im of s_c^H a against the raw table -- instead of one
// of length m^2: half the matrix-core work at m = 16.  Rows of an MFMA tile = (4 items) x (4 outputs), row = item + 4 out,
// so that the 4 outputs of one (item, bin) land in the 4 accumulator registers of ONE lane; a wave still owns 16 items,
// as four groups of 4 (group r = items g + 4r ... see below), and from `acc[t][r]` on the kernel is the same code: gate,
// literal refinement of near-null tiles (the difference loses ~m eps ||a||^2 like the projector form), stores, top-n.
// Qs then carries the coefficient vectors ((out*2M + k) * qstride + item), FB the raw-table image (build_TB).
// SIG = 2: n = 2 as described; SIG = 1: n = 1, two outputs per item, so a tile holds 8 items (row = item + 4 (2 half + out),
// registers (0, 1) = item g, (2, 3) = item g + 4) and a wave's 16 items are two groups: a quarter of the work at m = 16.
template <int M, int NMAX, bool SPEC, bool VEC4, int ABL = 0, int AUX = (1 | 2 | 16), int SIG = 0>
__global__ __launch_bounds__(256, (M <= 4 && NMAX <= 2) ? ((ABL & 2048) ? 3 : 4) : (M >= 9 ? 2 : 1)) void scan_mfma_kernel(const double* __restrict__ Qs,
                                                         const double2* __restrict__ FB,
                                                         float* __restrict__ spec,
                                                         double* __restrict__ cand,
                                                         uint32_t batch, uint32_t res, uint32_t qstride,
                                                         uint32_t nsplit, uint32_t nclass, uint32_t rows_per_class,
                                                         uint32_t keep_mask, uint32_t n, ScanRefine rf, uint32_t seq_walk = 0)
{
    constexpr int MM = M * M;
    constexpr int KS = SIG ? (2 * M + 3) / 4 : (MM + 3) / 4;   // MFMA k-steps per bin step
    static_assert(!SIG || (((SIG == 1 && M >= 6) || M >= 9) && KS <= 8), "the short form: one phase per step, no row classes (m >= 6)");
    constexpr int NG = (SIG == 1) ? 2 : 4;            // SIG: item groups per wave (8 or 4 items per tile)
    constexpr int SCH = (KS <= 8) ? KS : 8;           // k-steps per phase
    constexpr int PPS = (KS + SCH - 1) / SCH;         // phases per bin step
    constexpr int CPP = 2 * SCH;                      // 1-KiB chunks (64 x double2) per full phase
    constexpr int SPW = (CPP + 3) / 4;                // chunks one wave stages per phase
    __shared__ v2f64 stage[2][CPP * 64];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;

    // wave task = (16 rows of one class, range of 64-bin steps); the 4 waves of a block take 4 consecutive row groups
    // of the SAME class (rows_per_class is a multiple of 64) and the same step range.
    const uint32_t split = blockIdx.x % nsplit;
    // ORDER of the blocks: class by class -- the first quarter of a launch writes every fourth row of the WHOLE spectrum, the next quarter the rows
    // between them.  Round 5 tried the compact order (consecutive blocks = the nclass classes of the same 64 x nclass items; ABL & 4096, lab) on
    // the evidence of scripts/ubench_write_order.hip (a plain fill written in a compact moving window runs at 6.8 - 7.0 TB/s, the same bytes
    // grid-strided at 4.9 - 5.9): THIS kernel's stores alone take 0.566 ms class by class and 0.68 ms in the compact order, the whole scan
    // 0.706 against 0.712 (profiles/r05_write_order.txt).  Class by class stays.
    const uint32_t blk = blockIdx.x / nsplit;
    uint32_t p0;                                                     // first row of the wave in class-major numbering
    if constexpr ((ABL & 4096) == 0) p0 = (blk * 4 + wave) * 16;
    else p0 = (blk % nclass) * rows_per_class + ((blk / nclass) * 4 + wave) * 16;
    const uint32_t cls = __builtin_amdgcn_readfirstlane(p0 / rows_per_class);
    const uint32_t j0 = p0 - cls * rows_per_class;
    const uint32_t sh = ((res & 63u) * cls) & 63u;                   // bins: row start of the class inside its 256-B window
    const int shc = (int)(sh >> 2);                                  // ... in 4-bin columns (sh % 4 == 0 when nclass > 1)
    const uint32_t nsteps = (res + sh + 63u) >> 6;
    const uint32_t st_begin = (uint32_t)(((uint64_t)nsteps * split) / nsplit);
    const uint32_t st_end = (uint32_t)(((uint64_t)nsteps * (split + 1)) / nsplit);
    const uint32_t item0 = nclass * j0 + cls;                        // row x of the wave is item item0 + nclass*x
    // (rows beyond the batch, in the last groups of a class, still stage and sync but store nothing)

    // A operand: q[item of row c][e = 4 s + g]   (zero for the K padding e >= MM)
    double qa[SIG ? 1 : KS];
    [[maybe_unused]] double sa[SIG ? NG : 1][SIG ? KS : 1];    // SIG: per item group, row c of the tile = (item, output)
    const uint32_t it_c = item0 + nclass * (uint32_t)c;
    const uint32_t itc = (it_c < batch) ? it_c : (batch - 1);
    if constexpr (SIG) {
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            // D rows are item rows g' + 4r (r = accumulator register).  SIG = 2: group q must produce the d of rows g' + 4q,
            // so its tile row (item ig, output o) = ig + 4o carries wave item ig + 4q.  SIG = 1: tile row ig + 4 (2 half + o)
            // carries wave item ig + 4 (2q + half), whose d comes out of registers (2 half, 2 half + 1).
            const int sub = c >> 2;
            const int wrow = (SIG == 1) ? (c & 3) + 4 * (2 * q + (sub >> 1)) : (c & 3) + 4 * q;
            const int outp = (SIG == 1) ? (sub & 1) : sub;
            const uint32_t it_q = item0 + nclass * (uint32_t)wrow;
            const uint32_t itq = (it_q < batch) ? it_q : (batch - 1);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int e = 4 * s + g;
                sa[q][s] = (e < 2 * M) ? Qs[(size_t)(outp * 2 * M + e) * qstride + itq] : 0.0;
            }
        }
        qa[0] = 0.0;
    } else {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int e = 4 * s + g;
            qa[s] = (e < MM) ? Qs[(size_t)e * qstride + itc] : 0.0;
        }
    }

    double key[4][NMAX];                    // per item row r = 0..3 (row g + 4r)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < NMAX; ++i) key[r][i] = key_empty();
    // upper bounds of key[r][NMAX-1] for the top-n gate (see the epilogue): empty lists accept every finite value
    [[maybe_unused]] float gate_f[4];
    [[maybe_unused]] double gate_d[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        gate_f[r] = __builtin_inff();
        gate_d[r] = __builtin_bit_cast(double, (uint64_t)BAZ_KEY_EMPTY_BITS | 0xFFFFFull);
    }
    const bool refine_on = rf.Gs != nullptr;                         // wave-uniform
    uint32_t refined_wave = 0;                                       // (wave-uniform: values this wave recomputed in literal form)
    [[maybe_unused]] const float below_f = refine_on ? (float)rf.below : -1.0f;
    [[maybe_unused]] const double below_d = refine_on ? rf.below : -1.0;

    // FB is one flat array of chunks: step st, k-step s, tile pair h -> chunk (st*KS + s)*2 + h (st = -1 and
    // st = steps exist as padding).  Phase (st, p) covers k-steps [p*SCH, min(KS, (p+1)*SCH)).
    // This lane's element of a chunk: column c - shc of the step, or column c - shc + 16 of the step before.
    // Staging registers are four named clang vectors, not an array: an array written under `if (more)` and
    // read under a later `if (more)` was left in scratch by the compiler (with a full wait after every load).
    const int cc = c - shc;
    const v2f64* __restrict__ fb = reinterpret_cast<const v2f64*>(FB) + (g * 16 + (cc < 0 ? cc + 16 : cc)) -
                                   (cc < 0 ? KS * 2 * 64 : 0);
    const int wave4 = wave & 3;             // 256-thread blocks: lets the compiler fold `chunk < chunks-per-phase`
    // ROTATING LOADER (round 5, lab only: ABL & 1024).  Every wave fetches a quarter of the next phase into registers and waits for it -- and
    // vmcnt retires in issue order, so that wait is also a wait for the spectrum stores the wave issued before: every wave drains its own
    // stores once per step.  The lab form lets ONE wave per phase (they take turns) move the whole phase L2 -> LDS by LDS-DMA and be the only
    // one that waits.  Measured on one box, 262,144 items: 0.728 - 0.737 ms against 0.724 - 0.725 ms coherent, 0.854 against 0.830 incoherent
    // (profiles/r05_loader_ab.txt): the drain is not what bounds the walk.  Kept out of the product.
    constexpr bool ROT = (ABL & 1024) != 0;
    uint32_t turn = 0;                      // phases staged so far: wave (turn & 3) stages the next one
    v2f64 sreg0 = {0, 0}, sreg1 = {0, 0}, sreg2 = {0, 0}, sreg3 = {0, 0};
    static_assert(SPW <= 4, "a wave stages at most 4 chunks per phase");
    // (lab: ABL & 128 skips the table loads altogether, ABL & 256 issues them non-temporal)
#define BAZ_FB_LD(IDX) ((ABL & 256) ? __builtin_nontemporal_load(fb + (IDX)) : fb[(IDX)])
#define BAZ_STAGE_DMA(ST, P, BUF)                                                                \
    do {                                                                                         \
        if (wave4 == (int)(turn & 3u)) {                                                         \
            const int nch__ = 2 * ((KS - (P) * SCH < SCH) ? (KS - (P) * SCH) : SCH);             \
            const size_t ch0__ = ((size_t)(ST) * KS + (size_t)(P) * SCH) * 2;                    \
            _Pragma("unroll")                                                                    \
            for (int j__ = 0; j__ < CPP; ++j__)                                                  \
                if (j__ < nch__)                                                                 \
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(fb + (ch0__ + j__) * 64), \
                                                     (__attribute__((address_space(3))) void*)(&stage[(BUF)][j__ * 64]), 16, 0, 0); \
        }                                                                                        \
    } while (0)
#define BAZ_STAGE_LOAD(ST, P)                                                                    \
    do {   /* unconditional loads, index clamped into the phase */                               \
        if constexpr (!(ABL & 128)) {                                                            \
        const int nch__ = 2 * ((KS - (P) * SCH < SCH) ? (KS - (P) * SCH) : SCH);                 \
        const size_t ch0__ = ((size_t)(ST) * KS + (size_t)(P) * SCH) * 2;                        \
        sreg0 = BAZ_FB_LD((ch0__ + (wave4 < nch__ ? wave4 : nch__ - 1)) * 64);                   \
        if constexpr (SPW > 1) sreg1 = BAZ_FB_LD((ch0__ + (wave4 + 4 < nch__ ? wave4 + 4 : nch__ - 1)) * 64);   \
        if constexpr (SPW > 2) sreg2 = BAZ_FB_LD((ch0__ + (wave4 + 8 < nch__ ? wave4 + 8 : nch__ - 1)) * 64);   \
        if constexpr (SPW > 3) sreg3 = BAZ_FB_LD((ch0__ + (wave4 + 12 < nch__ ? wave4 + 12 : nch__ - 1)) * 64); \
        }                                                                                        \
    } while (0)
#define BAZ_STAGE_STORE(BUF, P)                                                                  \
    do {                                                                                         \
        const int nch__ = 2 * ((KS - (P) * SCH < SCH) ? (KS - (P) * SCH) : SCH);                 \
        if (wave4 < nch__) stage[(BUF)][wave4 * 64 + lane] = sreg0;                              \
        if constexpr (SPW > 1) { if (wave4 + 4 < nch__) stage[(BUF)][(wave4 + 4) * 64 + lane] = sreg1; }   \
        if constexpr (SPW > 2) { if (wave4 + 8 < nch__) stage[(BUF)][(wave4 + 8) * 64 + lane] = sreg2; }   \
        if constexpr (SPW > 3) { if (wave4 + 12 < nch__) stage[(BUF)][(wave4 + 12) * 64 + lane] = sreg3; } \
    } while (0)

    if (st_begin < st_end) {
        if constexpr (ROT) {
            BAZ_STAGE_DMA(st_begin, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ++turn;
        } else {
            BAZ_STAGE_LOAD(st_begin, 0);
            BAZ_STAGE_STORE(0, 0);
        }
    }
    __syncthreads();

    int buf = 0;
    v4f32 sv[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};   // spectrum values of item row r
    // spectrum addressing: uniform base = the wave's first row + per-lane 32-bit byte offsets of row r (rows are
    // nclass items apart), minus the class shift; the step offset goes into the SGPR soffset
    float* __restrict__ spec_base = SPEC ? spec + (size_t)item0 * res - sh : nullptr;   // (sh > 0 only for classes k >= 1: item0 >= 1)
    uint32_t soff0 = 0;                                              // row g's byte offset; rows g + 4 r add r * row4 through the scalar offset
    const uint32_t row4 = 4u * nclass * res * 4u;
    // raw buffer over this wave's rows: offsets stay < 16*nclass*res*4 <= 64 MiB (nclass <= 16, res <= 65536)
    [[maybe_unused]] __amdgpu_buffer_rsrc_t spec_rsrc = __builtin_amdgcn_make_buffer_rsrc(spec_base, 0, 0x7FFFFFFF, 0x00020000);
    bool row_ok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        // BYTE offset, from the shifted base, of bin (64*st - sh + 4c) at st = 0 (lanes with negative bins never store)
        if (r == 0) soff0 = ((uint32_t)g * nclass * res + 4u * (uint32_t)c) * 4u;
        row_ok[r] = (item0 + nclass * (uint32_t)(g + 4 * r)) < batch;
    }
    // STRIDED WALK (round 5; scan_i8_kernels.hip.h has the measurement): the range's steps in SW interleaved sweeps instead of left to
    // right.  Every step stages its own operands and stores its own 256-B pieces and the lists order their keys by (d, bin), so the order
    // changes no result -- but after one coarse sweep the lists hold values from near every null, and the top-n gate then fires only where a
    // later sweep passes a null's bottom instead of on every tile of the walk down to the first one.  seq_walk != 0 (lab): left to right.
    const uint32_t nst = st_end - st_begin;
    const uint32_t SW = seq_walk ? 1u : (nst >= 64u ? 8u : (nst >= 16u ? 4u : (nst >= 6u ? 2u : 1u)));
    uint32_t st = st_begin, sweep = 0;
    for (uint32_t it = 0; it < nst; ++it) {
        uint32_t st_next = st + SW, sweep_next = sweep;          // the step after this one in walk order (wave-uniform)
        if (st_next >= st_end) { sweep_next = sweep + 1; st_next = st_begin + sweep_next; }
        const bool has_next = it + 1 < nst;
        v4f64 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = (v4f64){0, 0, 0, 0};
        const uint32_t bin = st * 64 + 4 * c - sh;   // this lane's first bin of the step (wraps above res when negative)

#pragma unroll
        for (int p = 0; p < PPS; ++p) {
            // 1. fetch this wave's quarter of the NEXT phase from L2 (lands in registers while the MFMAs run)
            const bool last_p = (p == PPS - 1);
            const bool more = !last_p || has_next;                          // wave-uniform
            if constexpr (ROT) { if (more) BAZ_STAGE_DMA(last_p ? st_next : st, last_p ? 0 : p + 1, buf ^ 1); }
            else { if (more) BAZ_STAGE_LOAD(last_p ? st_next : st, last_p ? 0 : p + 1); }

            // 2. this phase: B operands from LDS, MFMAs
            if constexpr (SIG) {
                // ||a||^2 of this lane's 4 bins (huge outside the table), then per group of 4 items the four outputs
                const v4f64 a2v = *reinterpret_cast<const v4f64*>(rf.A2 + (size_t)st * 64 + 4 * c);
#pragma unroll
                for (int q = 0; q < NG; ++q) {
                    v4f64 tmp[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) tmp[t] = (v4f64){0, 0, 0, 0};
#pragma unroll
                    for (int sl = 0; sl < KS; ++sl) {
                        const v2f64 f01 = stage[buf][(2 * sl) * 64 + lane];
                        const v2f64 f23 = stage[buf][(2 * sl + 1) * 64 + lane];
                        tmp[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[q][sl], f01.x, tmp[0], 0, 0, 0);
                        tmp[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[q][sl], f01.y, tmp[1], 0, 0, 0);
                        tmp[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[q][sl], f23.x, tmp[2], 0, 0, 0);
                        tmp[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(sa[q][sl], f23.y, tmp[3], 0, 0, 0);
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if constexpr (SIG == 1) {
                            acc[t][2 * q] = a2v[t] - (tmp[t][0] * tmp[t][0] + tmp[t][1] * tmp[t][1]);
                            acc[t][2 * q + 1] = a2v[t] - (tmp[t][2] * tmp[t][2] + tmp[t][3] * tmp[t][3]);
                        } else {
                            acc[t][q] = a2v[t] - ((tmp[t][0] * tmp[t][0] + tmp[t][1] * tmp[t][1]) +
                                                  (tmp[t][2] * tmp[t][2] + tmp[t][3] * tmp[t][3]));
                        }
                    }
                }
            } else {
            if constexpr (ABL & 8192) __builtin_amdgcn_s_setprio(3);        // lab (round 6): the matrix phase of a wave ahead of its mates' vector work
            if constexpr (ABL & 16384) __builtin_amdgcn_s_setprio(0);       // lab: ... or behind it
#pragma unroll
            for (int sl = 0; sl < SCH; ++sl) {
                const int s = p * SCH + sl;
                if (s < KS) {
                    const v2f64 f01 = stage[buf][(2 * sl) * 64 + lane];       // tiles (0,1) of k-step s
                    const v2f64 f23 = stage[buf][(2 * sl + 1) * 64 + lane];   // tiles (2,3)
                    if constexpr (ABL & 8) {   // lab only
                        acc[0][s & 3] += qa[s] * f01.x; acc[1][s & 3] += qa[s] * f01.y;
                        acc[2][s & 3] += qa[s] * f23.x; acc[3][s & 3] += qa[s] * f23.y;
                    } else {
                        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[s], f01.x, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[s], f01.y, acc[1], 0, 0, 0);
                        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[s], f23.x, acc[2], 0, 0, 0);
                        acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[s], f23.y, acc[3], 0, 0, 0);
                    }
                }
            }
            if constexpr (ABL & 8192) __builtin_amdgcn_s_setprio(0);
            if constexpr (ABL & 16384) __builtin_amdgcn_s_setprio(3);
            }

            // 3. after the last phase of the step: epilogue arithmetic (no memory traffic yet).
            //    sv (the store data) lives across iterations: the empty asm keeps the PREVIOUS step's values
            //    allocated until here, so the accumulators above can never be given the registers that still
            //    feed in-flight stores (the compiler would otherwise guard that WAR hazard with s_waitcnt
            //    vmcnt(..) in the middle of the MFMA sequence, i.e. wait for fresh stores every step).
            if (last_p) {
#pragma unroll
                for (int r = 0; r < 4; ++r) asm volatile("" ::"v"(sv[r]));
                // Top-n gate.  The key network below (3 fp64 min/max + the key packing per value: ~80 of the ~145
                // VALU instructions of a round-1 step) only changes a list when some value beats the list's LAST
                // entry.  gate[r] bounds that entry from above in the precision the step has at hand anyway --
                // (float)|d| with the spectrum port, |d| itself without -- and the network runs only in steps where
                // SOME lane of the wave sees a value at or below its bound (wave-uniform branch, no divergence).
                // Conservative by monotonicity of the f64->f32 rounding: a key k < list[last] has
                // |d| <= (list[last] | low bits), hence (float)|d| <= (float)(list[last] | low bits) = gate, so no
                // insertion is ever skipped and the lists are bit-identical to running the network on every value;
                // NaN never passes (like .cc:131).  The bound never drops below `refine_below`, so the same vote
                // also catches the near-null tiles.  Items of one stream see the same scene, so the 16 items of a
                // wave rise together and most steps skip (profiles/r02_scan_gate.txt).
                bool hit = false, low = false;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (SPEC && !(ABL & 4)) {
                        float fd[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) fd[t] = (float)acc[t][r];   // sign (rounding noise around 0) dropped by |.| below
                        if constexpr (!(ABL & 2)) {
                            float mn;
                            asm("v_min3_f32 %0, |%1|, |%2|, |%3|" : "=v"(mn) : "v"(fd[0]), "v"(fd[1]), "v"(fd[2]));
                            asm("v_min_f32 %0, %1, |%2|" : "=v"(mn) : "v"(mn), "v"(fd[3]));
                            hit |= (mn <= gate_f[r]);
                            low |= (mn <= below_f);
                        }
                        // the spectrum values right away (strength_f32(|d|); ||G^H a||^2 >= 0 in the reference): the
                        // converted values then die here instead of living across the branch below, which matters for
                        // the register budget at m >= 9; a refined tile redoes its 16 values
#pragma unroll
                        for (int t = 0; t < 4; ++t) sv[r][t] = __builtin_amdgcn_rcpf(fabsf(fd[t]));
                    } else {
                        if constexpr (!(ABL & 2)) {
                            double m01, m23, mn;
                            asm("v_min_f64 %0, |%1|, |%2|" : "=v"(m01) : "v"(acc[0][r]), "v"(acc[1][r]));
                            asm("v_min_f64 %0, |%1|, |%2|" : "=v"(m23) : "v"(acc[2][r]), "v"(acc[3][r]));
                            mn = vmin64(m01, m23);
                            hit |= (mn <= gate_d[r]);
                            low |= (mn <= below_d);
                        }
                    }
                }
                if constexpr (!(ABL & 2)) {
                    if ((ABL & 64) || __any(hit)) {     // (ABL & 64: lab, the ungated network of round 1)
                        if (refine_on && __any(low)) {  // near-null tile: redo it in the reference's literal form
                            const int cr = c - shc;
                            const v2f64* __restrict__ tbl = reinterpret_cast<const v2f64*>(rf.TB) +
                                (g * 16 + (cr < 0 ? cr + 16 : cr)) - (cr < 0 ? ((2 * M + 3) / 4) * 2 * 64 : 0);
                            uint32_t cnt = literal_tile<M>(acc, rf, tbl, st, itc, g, qstride, (int)M - (int)n, bin, res, row_ok);
                            // Round 5: literal_tile() is inlined and its loads land in registers that later hold the store data.  Its values
                            // are all consumed in there -- but hipcc's wait-count pass merges control flow conservatively, and on the COMMON path
                            // (no literal tile) it put `s_waitcnt vmcnt(0)` in front of the step's LAST spectrum store: every wave drained its
                            // first three stores of EVERY step before issuing the fourth (seen in the ISA of rounds 2 - 4).  An explicit wait at the
                            // end of the rare path tells the pass that nothing is pending when the paths join: the four stores issue back to back
                            // and drain behind the next step's arithmetic.  (0x0F70: vmcnt(0), gfx9 encoding.)
                            __builtin_amdgcn_s_waitcnt(0x0F70);
                            if constexpr (SPEC && !(ABL & 4)) {
#pragma unroll
                                for (int r = 0; r < 4; ++r)
#pragma unroll
                                    for (int t = 0; t < 4; ++t) sv[r][t] = strength_f32(fabs(acc[t][r]));
                            }
                            if (rf.count) {         // statistic (baz_music_refined_items): values recomputed, summed in a
                                                    // scalar register; ONE atomic per wave at the end (round 4: an atomic per
                                                    // refined step -- 290,000 per launch of an incoherent 60-dB batch, all on
                                                    // one address -- cost 0.46 of that scan's 2.24 ms)
#pragma unroll
                                for (int msk = 1; msk < 64; msk <<= 1) cnt += __shfl_xor(cnt, msk, 64);
                                refined_wave += (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt);
                            }
                        }
                        const uint32_t nobin = ~keep_mask;          // bin field of a key that must never be selected
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
#pragma unroll
                            for (int t = 0; t < 4; ++t)
                                key_insert_new<NMAX>(key[r], make_key(acc[t][r], (bin + t < res) ? bin + t : nobin, keep_mask));
                            const uint64_t kb = __builtin_bit_cast(uint64_t, key[r][NMAX - 1]) | (uint64_t)(~keep_mask);
                            gate_d[r] = fmax(__builtin_bit_cast(double, kb), below_d);
                            gate_f[r] = (float)gate_d[r];
                        }
                    }
                }
                if constexpr (ABL & 4) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int t = 0; t < 4; ++t) sv[r][t] = __builtin_bit_cast(float, (uint32_t)__builtin_bit_cast(uint64_t, acc[t][r]));
                }
            }

            // 4. publish the next phase (waits only for the stage loads and the previous step's stores) ...
            if constexpr (ROT) {
                // the loader of this phase waits for its LDS-DMA (and, in passing, for its own earlier stores); nobody else waits
                if (more && wave4 == (int)(turn & 3u)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (more) ++turn;
            } else {
                if (more) BAZ_STAGE_STORE(buf ^ 1, last_p ? 0 : p + 1);
            }

            // 5. ... then this step's spectrum stores, then the barrier
            if (last_p) {
                if constexpr (ABL & 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) asm volatile("" ::"v"(sv[r]));
                }
                if constexpr (SPEC && !(ABL & 1)) {
                    // Buffer stores: wave-uniform resource (this wave's rows) + loop-invariant 32-bit lane offsets +
                    // the step offset as SGPR soffset: the store operands are never recomputed, so no register they
                    // occupy is recycled while a store is in flight.  Default cache policy sc0|sc1|nt: the spectrum
                    // is written once and not read by this pipeline (the merge fetches n floats per item), so it
                    // streams to HBM instead of parking ~290 MB of dirty lines in L2 / Infinity Cache that the NEXT
                    // kernel (covariance of the following batch) then pays to drain (profiles/r01g_store_policy.txt).
                    const int step_off = (int)(st * 256u);
                    if constexpr (VEC4) {   // res % 4 == 0: a lane's 4 bins are all in or all out
                        if ((st > 0 || sh == 0) && st * 64 + 64 - sh <= res) { // wave-uniform: whole step inside the row
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (row_ok[r]) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, sv[r]), spec_rsrc, (int)soff0, step_off + (int)(row4 * (uint32_t)r), AUX);
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (row_ok[r] && bin < res) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, sv[r]), spec_rsrc, (int)soff0, step_off + (int)(row4 * (uint32_t)r), AUX);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const v4u32 u = __builtin_bit_cast(v4u32, sv[r]);
#pragma unroll
                            for (int t = 0; t < 4; ++t)
                                if (row_ok[r] && bin + t < res)
                                    __builtin_amdgcn_raw_buffer_store_b32(u[t], spec_rsrc, (int)(soff0 + 4u * t), step_off + (int)(row4 * (uint32_t)r), AUX);
                        }
                    }
                }
            }
            if constexpr (ROT) {
                // (a raw barrier: __syncthreads() would put a vmcnt(0) in front of it -- the LDS-DMA writes are LDS writes to the compiler --
                // and make every wave wait for the stores it has just issued; this phase's ds_reads have been consumed by its MFMAs)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            } else {
                __syncthreads();
            }
            buf ^= 1;
        }
        st = st_next;
        sweep = sweep_next;
    }

    if (rf.count && lane == 0 && refined_wave) atomicAdd(rf.count, (unsigned long long)refined_wave);
    // merge the 16 lanes c = 0..15 that share an item row, then emit this range's candidates.
    // (Letting the wave that has walked a row's WHOLE bin range write ang / lvl itself -- lvl read back from its own
    // spectrum stores after s_waitcnt vmcnt(0) -- saves the merge launch and loses more than it saves: scan 0.721 ->
    // 0.805 ms per 262,144 cfg2 items against a 0.015 ms merge; every wave ends on a wait for its write-through stores.)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        key_merge_xor<NMAX>(key[r], 1);
        key_merge_xor<NMAX>(key[r], 2);
        key_merge_xor<NMAX>(key[r], 4);
        key_merge_xor<NMAX>(key[r], 8);
        const uint32_t it = item0 + nclass * (uint32_t)(g + 4 * r);
        if (c == 0 && it < batch) {
#pragma unroll
            for (int i = 0; i < NMAX; ++i) cand[((size_t)it * nsplit + split) * NMAX + i] = key[r][i];
        }
    }
}

#undef BAZ_STAGE_LOAD
#undef BAZ_STAGE_DMA
#undef BAZ_FB_LD
#undef BAZ_STAGE_STORE

// Final top-n over the per-range candidate keys (one thread per item), ang / lvl outputs
// (lib/baz_music_doa.cc:129-155).  lvl[i] is read back from the spectrum this launch sequence just
// wrote when port 2 is wired, so that lvl[i] == spectrum[bin_i] holds bit for bit as in the reference.
template <int NMAX>
__global__ __launch_bounds__(256) void topn_merge_kernel(const double* __restrict__ cand,
                                                          const float* __restrict__ spec,
                                                          float* __restrict__ ang, float* __restrict__ lvl,
                                                          uint32_t batch, uint32_t res, uint32_t n, uint32_t nsplit,
                                                          uint32_t keep_mask, unsigned long long* __restrict__ next_stat,
                                                          unsigned long long* __restrict__ fire_dev = nullptr,
                                                          unsigned long long* __restrict__ fire_host = nullptr,
                                                          unsigned long long fire_tag = 0ull)
{
    const uint32_t it = blockIdx.x * 256 + threadIdx.x;
    // the refinement statistic is double-buffered by call: this launch clears the counter the NEXT call's scan adds to
    // (a hipMemsetAsync per call was a 5-us kernel of its own between two launches)
    if (it == 0 && next_stat) *next_stat = 0ull;
    // the gated scan's fire statistic of THIS call (scan_coarse_kernel's fstat) goes to a page-locked host word the context's sorting
    // policy reads at a later call without synchronising: counts first, then the tag (sorted or not, call number), a fence between
    if (it == 0 && fire_dev && fire_host) {
        fire_host[0] = fire_dev[0];
        fire_host[1] = fire_dev[1];
        fire_dev[0] = 0ull;
        fire_dev[1] = 0ull;
        __threadfence_system();
        fire_host[2] = fire_tag;
    }
    if (it >= batch) return;
    double key[NMAX];
#pragma unroll
    for (int i = 0; i < NMAX; ++i) key[i] = key_empty();
    const size_t base = (size_t)it * nsplit * NMAX;
    for (uint32_t k = 0; k < nsplit * NMAX; ++k) key_insert<NMAX>(key, cand[base + k]);
#pragma unroll
    for (int i = 0; i < NMAX; ++i)
        if (i < (int)n) {
            const uint64_t b = __builtin_bit_cast(uint64_t, key[i]);
            const uint32_t bin = (uint32_t)b & ~keep_mask;
            const bool used = (b < (uint64_t)BAZ_KEY_EMPTY_BITS) && (bin < res);
            float a = 0.0f, l = 0.0f;                                          // (0, 0): .cc:95
            if (used) {
                a = (float)((double)bin * 360.0 / (double)res);                // .cc:134,152
                if (spec) l = spec[(size_t)it * res + bin];                    // .cc:153 (== spectrum[bin])
                else l = strength_f32(__builtin_bit_cast(double, b & ~(uint64_t)(~keep_mask)));
            }
            ang[(size_t)it * n + i] = a;
            if (lvl) lvl[(size_t)it * n + i] = l;
        }
}

// -------------------------------------------------------------------------------------
// 6. OPT-IN extension (not reference behaviour): the n strongest LOCAL MAXIMA of the
//    pseudo-spectrum instead of the n strongest bins (the reference's top-n, .cc:129-141, typically returns adjacent
//    bins of one lobe).  Bin b is a peak when s[b] > s[b-1] and s[b] >= s[b+1] on the circle (a plateau counts once,
//    at its first bin; NaN never qualifies).  Output format, ordering (descending strength, earlier bin first on
//    ties) and the (0, 0) fill for missing peaks are those of the reference path.  One wave per item, reads the
//    spectrum row the scan has just written.
// -------------------------------------------------------------------------------------
template <int NMAX>
__global__ __launch_bounds__(256) void peak_pick_kernel(const float* __restrict__ spec, float* __restrict__ ang,
                                                         float* __restrict__ lvl, uint32_t batch, uint32_t res,
                                                         uint32_t n)
{
    const int lane = threadIdx.x & 63;
    const uint32_t it = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (it >= batch) return;
    const float* __restrict__ s = spec + (size_t)it * res;
    double key[NMAX];
#pragma unroll
    for (int i = 0; i < NMAX; ++i) key[i] = key_empty();
    for (uint32_t b = lane; b < res; b += 64) {
        const float v = s[b];
        const float vp = s[b == 0 ? res - 1 : b - 1];
        const float vn = s[b + 1 == res ? 0 : b + 1];
        if (v > vp && v >= vn && v > 0.0f) {
            // smaller key = stronger peak, then earlier bin: high word 0x7F800000 - bits(v) (v > 0: bits <= 0x7F800000)
            const uint64_t k = ((uint64_t)(0x7F800000u - __float_as_uint(v)) << 32) | (uint64_t)b;
            key_insert<NMAX>(key, __builtin_bit_cast(double, k));
        }
    }
#pragma unroll
    for (int mask = 1; mask < 64; mask <<= 1) key_merge_xor<NMAX>(key, mask);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NMAX; ++i)
            if (i < (int)n) {
                const uint64_t k = __builtin_bit_cast(uint64_t, key[i]);
                const bool used = k < (uint64_t)BAZ_KEY_EMPTY_BITS;
                const uint32_t bin = (uint32_t)k;
                ang[(size_t)it * n + i] = used ? (float)((double)bin * 360.0 / (double)res) : 0.0f;
                if (lvl) lvl[(size_t)it * n + i] = used ? __uint_as_float(0x7F800000u - (uint32_t)(k >> 32)) : 0.0f;
            }
    }
}

}  // namespace bazmusic
