// scan_i8p_kernels.hip.h -- the pseudo-spectrum scan (lib/baz_music_doa.cc:101-121) for 2 .. 4 antennas on the INT8 matrix core
// with LEVEL-PACKED operands (round 5).  gfx950 only.
//
// Why.  At the headline shape (m = 4) d = a^H Q a = sum_e q_e F_e has only MM = 16 real terms.  scan_mfma_kernel spends four
// v_mfma_f64_16x16x4 per 16-item x 16-bin tile on it (~256 cycles: the fp64 matrix peak, 0.40 ms per 262,144 items), the int8
// form of scan_i8_kernels.hip.h would leave three quarters of K = 64 empty.  But digit pairs (s, t) of the same LEVEL l = s + t
// share an accumulator -- so they can be laid side by side along K:
//     B  = [ F_0 | F_1 | F_2 | F_3 ]             slot g (K = 16 g .. 16 g + 15) = digit g of the 16 terms       (one 1-KiB operand per tile)
//     A_l = [ q_l | q_(l-1) | q_(l-2) | q_(l-3) ]  slot g = digit l - g of q (zero where l - g < 0 or > 6)        (loop invariant)
//     level l of the four leading digits = ONE v_mfma_i32_16x16x64_i8(A_l, B):   4 MFMAs (~64 cycles) per tile for levels 0 .. 3.
// The digits, the fixed-point scales, the three per-value forms and every error bound are scan_i8_kernels.hip.h's with MM = 16
// (same I8Params): four digits where (float)|d4| > T4, else five (level 4: A_4 on B plus [q_0|0|0|0] on B' = [F_4|F_5|F_6|0]),
// else seven (levels 5, 6: A_5, A_6 on B, A_1, A_2 on B'), literal form near nulls, fp64 form for rows that are not projectors.
// Both table operands (4 + 4 KiB per 64-bin step) go L2 -> LDS by LDS-DMA, double-buffered, shared by the 4 waves of a workgroup.
//
// What differs from scan_i8_kernel:
//   * ROW CLASSES (music_kernels.hip.h 4.): m <= 5 is bound by its spectrum stores, so rows are walked by class (rows whose byte
//     offset agrees mod 256) and every 256-B store piece is aligned; the shifted table window is assembled by the staging loads
//     (lane (g, c) fetches column c - sh/4 of the step, or column c - sh/4 + 16 of the step before).
//   * the four-digit value is combined in FLOAT32: hw = A_0 256 + A_1, lw = A_2 256 + A_3 (int32), dv = fma((float) hw, 65536, (float) lw)
//     -- two full-rate conversions and one f32 FMA instead of two fp64 conversions, an fp64 FMA and an fp64 -> f32 conversion.
//     |dv - V| <= 2^-23 |V| + 8 (the two conversions and the FMA), V >= T4 / wt[3] ~ 2^32 for a value that keeps the form: a relative
//     error of 1.3e-7 on top of the 7.5e-7 the form promises for d, inside the same float that is stored.  The thresholds compared
//     with that float carry the slack (T4 (1 + 2^-21), gate (1 + 2^-21)), so the DECISIONS are those of the exact value: a value at or
//     below its row's top-n gate or T4 is always seen by the exact (fp64) path below.
//   * a step is walked in two passes: a straight-line pass over its 4 tiles (1 LDS read, 4 MFMAs, the per-value float work, one
//     vote per tile kept in a scalar mask), then -- rarely -- the flagged tiles again from scratch in the exact path (their 4 MFMAs
//     are cheaper than keeping 16 accumulators alive across the vote).
// A value's bits depend on that value alone (which form it takes is decided by its own four- / five-digit value; the float
// of a four-digit value is the same expression in both passes).
#pragma once

#include "scan_i8_kernels.hip.h"

namespace bazmusic {

// image of the packed operands: [64-bin step, one padded step in front and one behind][tile t][lane] x 16 B, B then B'
constexpr size_t I8P_STEP_UNITS = 4 * 64;                                                    // uint4 per step and operand
__host__ __device__ inline size_t i8p_operand_units(uint32_t steps) { return ((size_t)steps + 2) * I8P_STEP_UNITS; }
__host__ __device__ inline size_t i8p_image_bytes(uint32_t steps) { return 2 * i8p_operand_units(steps) * 16; }

// scan_mfma_kernel's projector GEMM for one tile through a lane pointer that already carries the row-class shift
template <int M>
__device__ __noinline__ v4f64 exact16s(const double* __restrict__ Qs, const double2* __restrict__ FB, const int cc, const uint32_t itn,
                                       const int g, const uint32_t qstride, const uint32_t st, const int t)
{
    constexpr int MM = M * M, KS = (MM + 3) / 4;
    const double2* __restrict__ fbs = FB + (g * 16 + (cc < 0 ? cc + 16 : cc)) - (cc < 0 ? KS * 2 * 64 : 0);
    const double* __restrict__ qp = Qs + itn + (size_t)g * qstride;                                   // e = 4 s + g
    const double* __restrict__ fb = reinterpret_cast<const double*>(fbs + ((size_t)st * KS * 2 + (size_t)(t >> 1)) * 64) + (t & 1);
    v4f64 acc = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const double a = (4 * s + g < MM) ? qp[(size_t)(4 * s) * qstride] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, fb[(size_t)s * 256], acc, 0, 0, 0);
    }
    return acc;
}

// VAL: validation build (baz_music_debug_i8_margin): every tile runs every form and the fp64 form; margin[0 .. 2] = worst
// |d5 - d| / E5, |d7 - d| / allowance, |d4 - d| / E4 over the rows that take the integer forms; outputs are the fp64 form's.
// ABL (lab builds only; timing, results are wrong): 1 no spectrum stores, 2 no tile arithmetic (staging, barriers, stores of a constant),
// 4 the first pass alone (no tile is ever flagged).
template <int M, int NMAX, bool SPEC, bool VEC4, bool VAL = false, int ABL = 0>
__global__ __launch_bounds__(256, (NMAX <= 2 && !VAL) ? 4 : 2) void scan_i8p_kernel(
    const double* __restrict__ Qs, const uint4* __restrict__ P1, const uint4* __restrict__ P2, const double2* __restrict__ FB,
    float* __restrict__ spec, double* __restrict__ cand, uint32_t batch, uint32_t res, uint32_t qstride, uint32_t nsplit,
    uint32_t nclass, uint32_t rows_per_class, uint32_t keep_mask, uint32_t n, ScanRefine rf, I8Params ip,
    unsigned long long* __restrict__ stat, unsigned long long* __restrict__ margin)
{
    constexpr int MM = M * M;
    constexpr int NS = I8_NS, ND = I8_ND;
    static_assert(M >= 2 && M <= 4, "level-packed operands: m^2 <= 16 terms per slot");
    static_assert(I8_NS == 5 && I8_ND == 7, "the level lists below are written out for five + two digits");
    // ONE __shared__ object (see scan_i8_kernel): [2 buffers][B: 4 tiles, B': 4 tiles][64 lanes], then the q operands of levels 4 .. 6
    __shared__ uint4 lds_all[2 * 512 + 4 * 3 * 64];
    uint4 (*stage)[512] = reinterpret_cast<uint4 (*)[512]>(&lds_all[0]);
    v4i32 (*a456)[3][64] = reinterpret_cast<v4i32 (*)[3][64]>(&lds_all[2 * 512]);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;

    // wave task = (16 rows of one class, range of 64-bin steps): scan_mfma_kernel's geometry
    const uint32_t split = blockIdx.x % nsplit;
    const uint32_t p0 = ((blockIdx.x / nsplit) * 4 + wave) * 16;
    const uint32_t cls = __builtin_amdgcn_readfirstlane(p0 / rows_per_class);
    const uint32_t j0 = p0 - cls * rows_per_class;
    const uint32_t sh = ((res & 63u) * cls) & 63u;                   // bins: row start of the class inside its 256-B window
    const int shc = (int)(sh >> 2);
    const uint32_t nsteps = (res + sh + 63u) >> 6;
    const uint32_t st_begin = (uint32_t)(((uint64_t)nsteps * split) / nsplit);
    const uint32_t st_end = (uint32_t)(((uint64_t)nsteps * (split + 1)) / nsplit);
    const uint32_t item0 = nclass * j0 + cls;                        // row x of the wave is item item0 + nclass x

    const uint32_t it_n = item0 + nclass * (uint32_t)c;              // natural row c (fp64 forms)
    const uint32_t itn = (it_n < batch) ? it_n : (batch - 1);

    // this lane's element of a 1-KiB operand chunk: column c - shc of the step, or column c - shc + 16 of the step before.
    // As ONE non-negative byte offset from the chunk of the step BEFORE (the image carries a padded step in front), so that the
    // staging loads are a wave-uniform base (scalar registers) plus this one lane register for both operands.
    const int cc = c - shc;
    const uint32_t lane_off = (uint32_t)((cc < 0 ? 0 : (int)I8P_STEP_UNITS) + g * 16 + (cc < 0 ? cc + 16 : cc)) * 16u;

    // ---- int8 A operands: A[l] slot g = digit l - g of q(item pi(c)), terms j = 0 .. 15 in the slot's 16 bytes -----------------
    v4i32 A[4];
    bool sane_r[4];
    bool any_insane;
    {
        const uint32_t it_p = item0 + nclass * (uint32_t)((c >> 2) + 4 * (c & 3));    // permuted row c: register r of lane (g, c) = row g + 4 r
        const uint32_t itp = (it_p < batch) ? it_p : (batch - 1);
        v4i32 D[ND];
#pragma unroll
        for (int s = 0; s < ND; ++s) D[s] = (v4i32){0, 0, 0, 0};
        int ok = 1;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            double qv = Qs[(size_t)(j < MM ? j : 0) * qstride + itp];
            qv = (j < MM) ? qv : 0.0;
            const bool fine = fabs(qv) <= I8_QMAX;                       // false for NaN
            ok &= fine ? 1 : 0;
            qv = fine ? qv : 0.0;
            double r = __builtin_rint(qv * ip.sq);                       // |r| <= 2^54 (1 + 2^-10): every step below is exact
#pragma unroll
            for (int s = ND - 1; s >= 1; --s) {
                const double h = __builtin_floor(__builtin_fma(r, 0x1p-8, 0.5));     // floor((r + 128) / 256)
                const int dg = (int)__builtin_fma(-256.0, h, r);                     // in [-128, 127]
                D[s][j >> 2] |= (int)((unsigned)(dg & 255) << (8 * (j & 3)));
                r = h;
            }
            D[0][j >> 2] |= (int)((unsigned)((int)r & 255) << (8 * (j & 3)));
        }
        ok &= __shfl_xor(ok, 16, 64);                  // the 4 lanes (g = 0 .. 3) that hold the row
        ok &= __shfl_xor(ok, 32, 64);
        // slot g of level l holds digit l - g: select per lane (g is a lane property), zero outside 0 .. 6 and for rows without digits
        auto pick = [&](const int l) {
            v4i32 v = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < ND; ++s) {
                const bool hit = ok && (l - g == s);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = hit ? D[s][q] : v[q];
            }
            return v;
        };
#pragma unroll
        for (int l = 0; l < 4; ++l) A[l] = pick(l);
#pragma unroll
        for (int l = 4; l < 7; ++l) a456[wave][l - 4][lane] = pick(l);
#pragma unroll
        for (int r = 0; r < 4; ++r) sane_r[r] = __shfl(ok, 4 * g + r, 64) != 0;     // item g + 4 r = permuted row 4 g + r
        any_insane = __any(!ok);
    }

    double key[4][NMAX];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < NMAX; ++i) key[r][i] = key_empty();
    const bool refine_on = rf.Gs != nullptr;
    const double below_d = refine_on ? rf.below : -1.0;
    const double tacc_d = ip.t_acc;

    // ---- table staging: L2 -> LDS directly; wave w fetches tile w of B and of B' (1 KiB each) ------------------------------
    auto stage_load = [&](const uint32_t st, const int b) {
        // (P1 / P2 point at step 0; step st - 1 exists for st = 0: the padded step)
        const char* base1 = reinterpret_cast<const char*>(P1 + ((ptrdiff_t)st * 4 + wave - 4) * 64);      // wave-uniform
        const char* base2 = reinterpret_cast<const char*>(P2 + ((ptrdiff_t)st * 4 + wave - 4) * 64);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base1 + lane_off),
                                         (__attribute__((address_space(3))) void*)(&stage[b][wave * 64]), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base2 + lane_off),
                                         (__attribute__((address_space(3))) void*)(&stage[b][256 + wave * 64]), 16, 0, 0);
    };

    int buf = 0;
    v4f32 sv[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float* __restrict__ spec_base = SPEC ? spec + (size_t)item0 * res - sh : nullptr;   // (sh > 0 only for classes k >= 1: item0 >= 1)
    [[maybe_unused]] __amdgpu_buffer_rsrc_t spec_rsrc = __builtin_amdgcn_make_buffer_rsrc(spec_base, 0, 0x7FFFFFFF, 0x00020000);
    // store addressing: one lane register (row g, column 4c) + scalar offsets (the step, and 4 rows further per register r)
    const uint32_t soff0 = ((uint32_t)g * nclass * res + 4u * (uint32_t)c) * 4u;
    const uint32_t row4 = 4u * nclass * res * 4u;                       // bytes between the rows of registers r and r + 1 (wave-uniform)
    bool row_ok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) row_ok[r] = (item0 + nclass * (uint32_t)(g + 4 * r)) < batch;
    uint32_t refined = 0, fell = 0, nflag = 0;
    [[maybe_unused]] float worst5 = 0.0f, worst7 = 0.0f, worst4 = 0.0f;

    if (st_begin < st_end) stage_load(st_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const uint32_t nobin = ~keep_mask;
    const int nn = (int)M - (int)n;
    const double ws_d = ip.wt[NS - 2];                 // weight of the integer V of both bulk forms (level 3)
    const float ws_f = ip.ws_f;
    // the float thresholds carry the slack of the float32 combination (see the header): T4 and the gate times (1 + 2^-21)
    constexpr float SLACK = 1.0f + 0x1p-21f;
    const float t4_f = ip.t4_f * SLACK;
    float thr4[4];                                     // the row's threshold of the ONE comparison per value: max(top-n gate, T4), +inf while the list is empty
#pragma unroll
    for (int r = 0; r < 4; ++r) thr4[r] = __builtin_inff();

    const v4i32 Z = {0, 0, 0, 0};
    for (uint32_t st = st_begin; st < st_end; ++st) {
        const uint32_t bin = st * 64 + 4 * (uint32_t)c - sh;     // this lane's first bin of the step (tile t: bin + t); wraps above res when negative
        const bool more = st + 1 < st_end;                       // wave-uniform
        if (more) stage_load(st + 1, buf ^ 1);
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" ::"v"(sv[r]));          // (see scan_mfma_kernel: the store data stays put)

        const v4i32* __restrict__ Bp = reinterpret_cast<const v4i32*>(&stage[buf][0]) + lane;
        // ---- pass 1: four tiles, straight line ------------------------------------------------------------------------------------
        uint32_t flagged = VAL ? 15u : 0u;
        if constexpr (!(ABL & 2)) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const v4i32 b = Bp[t * 64];
                const v4i32 L0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[0], b, Z, 0, 0, 0);
                const v4i32 L1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[1], b, Z, 0, 0, 0);
                const v4i32 L2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[2], b, Z, 0, 0, 0);
                const v4i32 L3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[3], b, Z, 0, 0, 0);
                unsigned long long under = 0ull;                           // lanes with a value at or below its row's threshold
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // (the sign stays until the |.| source modifiers of the comparison and the reciprocal)
                    const float sdv = __builtin_fmaf((float)(L0[r] * 256 + L1[r]), 65536.0f, (float)(L2[r] * 256 + L3[r])) * ws_f;
                    under |= __builtin_amdgcn_ballot_w64(fabsf(sdv) <= thr4[r]);
                    if constexpr (SPEC) sv[r][t] = __builtin_amdgcn_rcpf(fabsf(sdv));
                }
                flagged |= under ? (1u << t) : 0u;
            }
            if constexpr ((ABL & 4) != 0) flagged = 0u;                    // lab: the first pass alone
            nflag += (uint32_t)__builtin_popcount(flagged);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) sv[r] = (v4f32){1.0f, 2.0f, 3.0f, (float)st};
        }
        // ---- pass 2 (rare): the flagged tiles again, every form that a value of theirs needs (ONE copy of this code: a rolled loop,
        // its results blended into the store registers at the end) ------------------------------------------------------------------
        if (flagged) {
#pragma nounroll
            for (int t = 0; t < 4; ++t) {
                if (!((flagged >> t) & 1u)) continue;                       // wave-uniform
                const v4i32 b = Bp[t * 64], b2 = Bp[256 + t * 64];
                v4f64 vd;
                float fdv[4];
                bool form4[4], need5 = false;
                {
                    v4i32 L[4];
#pragma unroll
                    for (int l = 0; l < 4; ++l) L[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[l], b, Z, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int hw = L[0][r] * 256 + L[1][r], lw = L[2][r] * 256 + L[3][r];
                        vd[r] = __builtin_fma((double)hw, 65536.0, (double)lw);                     // V, exactly
                        fdv[r] = fabsf(__builtin_fmaf((float)hw, 65536.0f, (float)lw) * ws_f);      // pass 1's float, the same expression
                        form4[r] = !VAL && (fdv[r] > t4_f);                   // per VALUE: its own four-digit value decides
                        need5 |= !form4[r];
                    }
                }
                [[maybe_unused]] const v4f64 d4 = {vd[0] * ws_d, vd[1] * ws_d, vd[2] * ws_d, vd[3] * ws_d};
                v4i32 L4 = Z;
                if (VAL || __any(need5)) {              // second tier: level 4 = A_4 on B plus [q_0|0|0|0] on B'
                    const v4i32 a4 = a456[wave][0][lane];
                    L4 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b, Z, 0, 0, 0);
                    L4 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[0], b2, L4, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double v5 = vd[r] + (double)(L4[r] >> 8);
                        vd[r] = form4[r] ? vd[r] : v5;
                        fdv[r] = form4[r] ? fdv[r] : fabsf((float)v5) * ws_f;
                    }
                }
                v4f64 d;
#pragma unroll
                for (int r = 0; r < 4; ++r) d[r] = vd[r] * ws_d;                // d4 or d5, exactly
                [[maybe_unused]] const v4f64 d5 = d;
                const bool in_table = bin + (uint32_t)t < res;
                // third tier: five-digit values at or below T take levels 5 and 6 of all seven digits and the low byte of level 4
                bool lowt = false;
#pragma unroll
                for (int r = 0; r < 4; ++r) lowt |= (VAL || (sane_r[r] && in_table)) && !form4[r] && !(fabs(d[r]) > tacc_d);
                if (VAL || __any(lowt)) {
                    ++fell;
                    const v4i32 a5 = a456[wave][1][lane], a6 = a456[wave][2][lane];
                    v4i32 L5 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a5, b, Z, 0, 0, 0);
                    v4i32 L6 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a6, b, Z, 0, 0, 0);
                    L5 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[1], b2, L5, 0, 0, 0);
                    L6 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[2], b2, L6, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double low = __builtin_fma((double)(L4[r] & 255), ip.wt[4],
                                                         __builtin_fma((double)L5[r], ip.wt[5], (double)L6[r] * ip.wt[6]));
                        const double d7 = d[r] + low;
                        const bool take = VAL || (!form4[r] && !(fabs(d[r]) > tacc_d));
                        d[r] = take ? d7 : d[r];
                        fdv[r] = take ? fabsf((float)d7) : fdv[r];
                    }
                }
                // rows whose coefficients are not a projector's: scan_mfma_kernel's fp64 form, bit for bit
                if (VAL || any_insane) {
                    const v4f64 ex = exact16s<M>(Qs, FB, cc, itn, g, qstride, st, t);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (VAL) {
                            const float r5 = (float)(fabs(d5[r] - ex[r]) / ip.e_bound);
                            const float r7 = (float)(fabs(d[r] - ex[r]) / (ip.e_refined + 0x1p-50 * fabs(ex[r])));
                            const float r4 = (float)(fabs(d4[r] - ex[r]) / ip.e4_bound);
                            const bool counts = sane_r[r] && row_ok[r] && in_table;        // (NaN never counts)
                            if (counts && r5 > worst5) worst5 = r5;
                            if (counts && r7 > worst7) worst7 = r7;
                            if (counts && r4 > worst4) worst4 = r4;
                        }
                        const bool takex = VAL || !sane_r[r];
                        d[r] = takex ? ex[r] : d[r];
                        fdv[r] = takex ? fabsf((float)ex[r]) : fdv[r];
                    }
                }
                // bins outside the table (first / last step of a row): zero digits gave d = 0; never selected, never stored
                if (!in_table) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { d[r] = 1e300; fdv[r] = __builtin_inff(); }
                }
                // top-n gate and near-null vote in fp64 (exact: the float comparison of pass 1 only decided that this path runs);
                // the gate = the list's last key with its bin field filled, never below `refine_below`
                bool hit = false, low = false;
                float outv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint64_t kb = __builtin_bit_cast(uint64_t, key[r][NMAX - 1]) | (uint64_t)(~keep_mask);
                    hit |= (fabs(d[r]) <= fmax(__builtin_bit_cast(double, kb), below_d));
                    low |= (fabs(d[r]) <= below_d);
                    outv[r] = __builtin_amdgcn_rcpf(fdv[r]);
                }
                if (__any(hit)) {
                    if (refine_on && __any(low)) {          // near-null values: the reference's literal form, per value
                        const v4f64 lit = literal16<M>(rf.Gs, rf.TB, itn, g, qstride, nn, in_table ? bin + (uint32_t)t : 0u);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool redo = (fabs(d[r]) <= rf.below) && in_table;
                            d[r] = redo ? lit[r] : d[r];
                            refined += (redo && row_ok[r]) ? 1u : 0u;
                            outv[r] = redo ? strength_f32(fabs(d[r])) : outv[r];
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        key_insert_new<NMAX>(key[r], make_key(d[r], in_table ? bin + (uint32_t)t : nobin, keep_mask));
                        const uint64_t kb = __builtin_bit_cast(uint64_t, key[r][NMAX - 1]) | (uint64_t)(~keep_mask);
                        const double gd = fmax(__builtin_bit_cast(double, kb), below_d);
                        // pass 1's threshold: at least the gate, rounded UP to a float and widened by the float combination's slack
                        float gu = (float)gd;
                        gu = ((double)gu < gd) ? __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, gu) + 1u) : gu;   // (gu >= 0, finite here)
                        thr4[r] = fmaxf(gu * SLACK, t4_f);
                    }
                }
                if constexpr (SPEC) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int tt = 0; tt < 4; ++tt) sv[r][tt] = (tt == t) ? outv[r] : sv[r][tt];
                }
            }
        }

        // the next step's operands have landed (and the PREVIOUS step's stores are done) ...
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ... then this step's spectrum stores, then the barrier
        if constexpr (SPEC && !(ABL & 1)) {
            const int step_off = (int)(st * 256u);
            if constexpr (VEC4) {   // res % 4 == 0: a lane's 4 bins are all in or all out
                if ((st > 0 || sh == 0) && st * 64 + 64 - sh <= res) { // wave-uniform: whole step inside the row
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (row_ok[r]) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, sv[r]), spec_rsrc, (int)soff0, step_off + (int)(row4 * (uint32_t)r), (1 | 2 | 16));
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (row_ok[r] && bin < res) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, sv[r]), spec_rsrc, (int)soff0, step_off + (int)(row4 * (uint32_t)r), (1 | 2 | 16));
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const v4u32 u = __builtin_bit_cast(v4u32, sv[r]);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (row_ok[r] && bin + (uint32_t)t < res)
                            __builtin_amdgcn_raw_buffer_store_b32(u[t], spec_rsrc, (int)(soff0 + 4u * t), step_off + (int)(row4 * (uint32_t)r), (1 | 2 | 16));
                }
            }
        } else if constexpr (SPEC) {
#pragma unroll
            for (int r = 0; r < 4; ++r) asm volatile("" ::"v"(sv[r]));
        }
        // (a raw barrier: __syncthreads() would put a vmcnt(0) in front of it and wait for the stores just issued)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        buf ^= 1;
    }

    if (rf.count) {
#pragma unroll
        for (int msk = 1; msk < 64; msk <<= 1) refined += __shfl_xor(refined, msk, 64);
        if (lane == 0 && refined) atomicAdd(rf.count, (unsigned long long)refined);
    }
    if (stat && lane == 0) {
        if (fell) atomicAdd(stat, (unsigned long long)fell);
        atomicAdd(stat + 1, (unsigned long long)(st_end - st_begin) * 4ull);
        if (nflag) atomicAdd(stat + 5, (unsigned long long)nflag);      // (lab read-out: tiles the second pass walked)
    }
    if constexpr (VAL) {
        unsigned int w5 = __builtin_bit_cast(unsigned int, worst5), w7 = __builtin_bit_cast(unsigned int, worst7);
        unsigned int w4 = __builtin_bit_cast(unsigned int, worst4);
#pragma unroll
        for (int msk = 1; msk < 64; msk <<= 1) {
            const unsigned int o5 = __shfl_xor(w5, msk, 64), o7 = __shfl_xor(w7, msk, 64), o4 = __shfl_xor(w4, msk, 64);
            w5 = o5 > w5 ? o5 : w5;
            w7 = o7 > w7 ? o7 : w7;
            w4 = o4 > w4 ? o4 : w4;
        }
        if (lane == 0 && margin) {                                 // ratios >= 0: their bits order like the values
            atomicMax(margin, (unsigned long long)w5);
            atomicMax(margin + 1, (unsigned long long)w7);
            atomicMax(margin + 2, (unsigned long long)w4);
        }
    }
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

}  // namespace bazmusic
