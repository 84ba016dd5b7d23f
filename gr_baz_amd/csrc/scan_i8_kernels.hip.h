// scan_i8_kernels.hip.h -- the pseudo-spectrum scan (lib/baz_music_doa.cc:101-121) for 6 .. 16 antennas with the bulk of
// the (item, bin) values on the INT8 matrix core, exactly accumulated, and the fp64 matrix core only where an a-priori
// error bound says so.  gfx950 only.
//
// Why.  d(item, bin) = a^H Q a = sum_e q_e F_e over MM = m^2 real terms (music_kernels.hip.h 4.) cancels, so it cannot be
// evaluated in float32, and on v_mfma_f64_16x16x4 it is bound by the fp64 matrix rate: config 3 (m = 8, 36,000 bins) ran its
// scan at 77 % of the 78.6 TFLOP/s peak and 26 % of HBM (round 3).  v_mfma_i32_16x16x64_i8 multiplies 8-bit integers ~50 x
// faster and ACCUMULATES IN INT32 WITHOUT ROUNDING -- the only error of an integer form is the one made when the operands
// are cut into digits, and that has a bound that needs no statistics (the Ozaki scheme, with integer slices).
//
// Integer form.  Both operands in fixed point with NS = 5 balanced base-256 digits (int8):
//     Qi_e = rint(q_e 2^(8 NS - 2))           |q_e| <= 1 for a projector (|Q_ij| <= 1/2 off the diagonal, q = 2 Re / -2 Im)
//     Fi_e = rint(F_e 2^(8 NS - 2) / Fscale)  Fscale = the power of two with max|F| / Fscale in (1/2 (1 + 2^-10), 1 + 2^-10]
//     X = sum_s x_s 256^(NS - 1 - s),  x_s in [-128, 127] (s >= 1),  |x_0| <= 65
// The product sum_e Qi_e Fi_e is the sum over digit pairs (s, t) of 256^(2 NS - 2 - s - t) A_st, A_st = sum_e q_es F_et: one
// K = 64 int8 MFMA per pair, 16-item x 16-bin tile and block of 64 terms.  Pairs are accumulated per LEVEL l = s + t (one
// int32 accumulator per level: |A| <= (l + 1) MM 2^14 < 2^31) and the levels l >= NS are dropped: NS (NS + 1) / 2 = 15 MFMAs
// per tile and block where the fp64 form issues 16 fp64 MFMAs of 4 x the cycles each.  Levels are combined per value in
// fp64 (every partial sum is an integer below 2^53 times a power of two: exact), d_int = H * unit.
//
// Error bound (a priori, in units of d):
//     digits cut off    |q_e - Qi_e / Sq| <= 2^(1 - 8 NS),  the same for F / Fscale       ->  <= MM Fscale 2^(2 - 8 NS) 1.001
//     levels dropped    pairs (s, t), s + t >= NS: <= MM 128^2 (NS - 1) 256^(NS - 2) 1.005 of the integer product
//                                                                                          ->  <= MM Fscale (NS - 1) 2^(2 - 8 NS) 1.005
//     E = MM Fscale NS 1.01 2^(2 - 8 NS)        (m = 8: 1.2e-9 Fscale; m = 16: 4.7e-9 Fscale;  ||a||^2 = m for the helper's tables)
// tests/lab/i8_split_study.py restates the scheme in numpy (worst observed error 0.04 - 0.09 E), tests/test_i8_scan.py pins
// the host-side image against it, and the VAL instantiation (baz_music_debug_i8_margin) evaluates both forms on EVERY
// (item, bin) of a batch on the hardware and returns the worst |d_int - d| / E.
//
// Where the integer value is used.  A value is within eps = 7.5e-7 of the fp64 one when d_int > T = E (1 + 1 / eps).  A
// 16-item x 64-bin step in which some |d_int| <= T (the bottom of the nulls: 0.5 % of the values of a 20-dB scene; bins
// outside the table and items whose coefficients are not a projector's have zero digits and always land here) is recomputed
// by scan_mfma_kernel's own fp64 instruction sequence (exact16: same operands, same k order, same bits), including its
// literal-form refinement of near-null values (T >= refine_below by orders of magnitude), and the VALUES at or below T take
// the fp64 result -- per value, so an (item, bin) pair's bits do not depend on its wave-mates.  Everything downstream of d --
// (float) conversion, v_rcp_f32, the store pattern, the gated top-n network, the candidate lists -- is scan_mfma_kernel's.
// The top-n keys of the other steps are built from d_int: two bins can change places against the fp64 scan only where their
// strengths agree to 2 eps = 1.5e-6 (the parity rule allows 2e-5), and lvl[i] == spectrum[bin_i] holds bit for
// bit as before (the merge reads lvl back from the spectrum it indexes).
//
// Layouts.  int8 / f16 MFMA C/D: col = lane & 15, row = 4 (lane >> 4) + reg; f64 MFMA: row = (lane >> 4) + 4 reg.  The int8 A
// operand therefore carries item pi(i) = (i >> 2) + 4 (i & 3) in row i, so that register r of lane (g, c) is item g + 4 r in
// BOTH forms (the trick of scan_coarse_kernels.hip.h).  A / B operand of the K = 64 form: lane (g, c) holds row / column c,
// k = 16 g + j in byte j of its 16 bytes (the same map on both sides: any consistent one gives the same dot product).
// Table image IB (built at set_table, build_i8_image): [64-bin step][tile t][block kb][digit s][lane] x 16 B, tile t of a
// step carrying the bins 64 st + 4 c + t in its columns like FB (a lane ends up with 4 consecutive bins: one 16-B store).
// A phase = TPP tiles (20 KiB at m <= 8 and at m >= 12) goes L2 -> LDS by global_load_lds_dwordx4, double-buffered, shared by
// the 4 waves of a workgroup; a wave owns 16 items and a range of steps, like scan_mfma_kernel (no row classes: m >= 6).
#pragma once

#include "music_kernels.hip.h"

namespace bazmusic {

typedef int v4i32 __attribute__((ext_vector_type(4)));

constexpr int I8_NS = 5;                                           // digits per operand = levels kept
constexpr int i8_nkb(int m) { return (m * m + 63) / 64; }          // blocks of 64 terms
constexpr int i8_tpp(int m) { return i8_nkb(m) == 1 ? 4 : (i8_nkb(m) == 2 ? 2 : 1); }     // tiles per staged phase
constexpr int i8_tile_units(int m) { return i8_nkb(m) * I8_NS * 64; }                      // 16-B units per 16-bin tile
constexpr double I8_QMAX = 1.0009765625;                           // |q_e| <= 1 + 2^-10, else the item never takes the integer form
constexpr double I8_EPS = 7.5e-7;                                  // relative accuracy promised for values that keep the integer form

struct I8Params {
    double wt[I8_NS];     // wt[l] = unit 256^(NS - 1 - l): weight of the level-l sum (unit = Fscale 2^(-8 NS - 4))
    double sq;            // 2^(8 NS - 2)
    double t_acc;         // T = E (1 + 1 / eps): at or below it a value takes the fp64 form
    float t_acc_f;        // (float) T rounded up
    double e_bound;       // E (VAL only)
};

// two adjacent levels fit one int32 (a_l 256 + a_(l+1)) while (l + 1) MM 2^22 + (l + 2) MM 2^14 < 2^31
constexpr bool i8_pair_ok(int mm, int l) { return (long long)(l + 1) * mm * 4194304ll + (long long)(l + 2) * mm * 16384ll < 2147483648ll; }

template <int MM, int L>
struct I8Comb {
    static __device__ __forceinline__ double run(const int (&a)[I8_NS], const double (&wt)[I8_NS])
    {
        if constexpr (L >= I8_NS) return 0.0;
        else if constexpr (L + 1 < I8_NS && i8_pair_ok(MM, L))
            return __builtin_fma((double)(a[L] * 256 + a[L + 1]), wt[L + 1], I8Comb<MM, L + 2>::run(a, wt));
        else return __builtin_fma((double)a[L], wt[L], I8Comb<MM, L + 1>::run(a, wt));
    }
};

// scan_mfma_kernel's projector GEMM for ONE 16-item x 16-bin tile (tile t of step st): same operands, same k order, hence
// the same bits.  Both operands come from L2 (q of the natural row c, FB where the image has it).  Not inlined: the ordinary
// steps do not pay its registers.
template <int M>
__device__ __noinline__ v4f64 exact16(const double* __restrict__ Qs, const double2* __restrict__ FB, const uint32_t itn,
                                      const int g, const int lane, const uint32_t qstride, const uint32_t st, const int t)
{
    constexpr int MM = M * M, KS = (MM + 3) / 4;
    const double* __restrict__ qp = Qs + itn + (size_t)g * qstride;                                   // e = 4 s + g
    const double* __restrict__ fb = reinterpret_cast<const double*>(FB + ((size_t)st * KS * 2 + (size_t)(t >> 1)) * 64 + lane) + (t & 1);
    v4f64 acc = {0, 0, 0, 0};
#pragma unroll 4
    for (int s = 0; s < KS; ++s) {
        const double a = (4 * s + g < MM) ? qp[(size_t)(4 * s) * qstride] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, fb[(size_t)s * 256], acc, 0, 0, 0);
    }
    return acc;
}

// VAL: validation build (baz_music_debug_i8_margin): every step runs both forms; the worst |d_int - d| / E over all (item,
// bin) whose item took the integer form is left in *margin (float bits, atomicMax); outputs are the fp64 form's.
// stat (may be nullptr): [0] += wave steps recomputed in the fp64 form, [1] += wave steps walked.
template <int M, int NMAX, bool SPEC, bool VEC4, bool VAL = false>
__global__ __launch_bounds__(256, 2) void scan_i8_kernel(const double* __restrict__ Qs, const uint4* __restrict__ IB,
                                                         const double2* __restrict__ FB, float* __restrict__ spec,
                                                         double* __restrict__ cand, uint32_t batch, uint32_t res,
                                                         uint32_t qstride, uint32_t nsplit, uint32_t keep_mask, uint32_t n,
                                                         ScanRefine rf, I8Params ip, unsigned long long* __restrict__ stat,
                                                         unsigned long long* __restrict__ margin)
{
    constexpr int MM = M * M;
    constexpr int NS = I8_NS;
    static_assert(I8_NS == 5, "the level lists below are written out for five digits");
    constexpr int NKB = i8_nkb(M);
    constexpr int TPP = i8_tpp(M);
    constexpr int PPS = 4 / TPP;                       // phases per 64-bin step
    constexpr int TU = i8_tile_units(M);               // 16-B units per tile
    constexpr int CH = TPP * NKB * NS;                 // 1-KiB chunks per phase
    static_assert(M >= 6 && M <= 16, "6 <= m <= 16 (row classes below, run-time-m kernels above)");
    __shared__ uint4 stage[2][CH * 64];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;

    const uint32_t split = blockIdx.x % nsplit;
    const uint32_t item0 = ((blockIdx.x / nsplit) * 4 + wave) * 16;          // first item of the wave
    const uint32_t nsteps = (res + 63u) >> 6;
    const uint32_t st_begin = (uint32_t)(((uint64_t)nsteps * split) / nsplit);
    const uint32_t st_end = (uint32_t)(((uint64_t)nsteps * (split + 1)) / nsplit);

    const uint32_t it_n = item0 + (uint32_t)c;                               // natural row c (fp64 form)
    const uint32_t itn = (it_n < batch) ? it_n : (batch - 1);

    // ---- int8 A operand: the digits of q(item pi(c)), k = 64 kb + 16 g + j ------------------------------------------
    v4i32 A[NKB][NS];
    bool sane_r[4];
    {
        const uint32_t it_p = item0 + (uint32_t)((c >> 2) + 4 * (c & 3));    // permuted row c
        const uint32_t itp = (it_p < batch) ? it_p : (batch - 1);
        int ok = 1;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int s = 0; s < NS; ++s) A[kb][s] = (v4i32){0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int e = 64 * kb + 16 * g + j;
                double qv = Qs[(size_t)(e < MM ? e : 0) * qstride + itp];    // (unconditional load, clamped: no branch per value)
                qv = (e < MM) ? qv : 0.0;
                const bool fine = fabs(qv) <= I8_QMAX;                       // false for NaN
                ok &= fine ? 1 : 0;
                qv = fine ? qv : 0.0;
                double r = __builtin_rint(qv * ip.sq);
#pragma unroll
                for (int s = NS - 1; s >= 1; --s) {
                    const double h = __builtin_floor(__builtin_fma(r, 0x1p-8, 0.5));     // floor((r + 128) / 256)
                    const int dg = (int)__builtin_fma(-256.0, h, r);                     // in [-128, 127]
                    A[kb][s][j >> 2] |= (int)((unsigned)(dg & 255) << (8 * (j & 3)));
                    r = h;
                }
                A[kb][0][j >> 2] |= (int)((unsigned)((int)r & 255) << (8 * (j & 3)));
                // (one word of digits at a time: without the fence the scheduler hoists all 16 NKB loads of the prologue to
                // its top and the 12 .. 15-antenna instantiations spill 150 .. 490 registers)
                if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        ok &= __shfl_xor(ok, 16, 64);                  // the 4 lanes (g = 0 .. 3) that hold the row
        ok &= __shfl_xor(ok, 32, 64);
        if (!ok) {                                     // not a projector's coefficients: zero digits, every step takes the fp64 form
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int s = 0; s < NS; ++s) A[kb][s] = (v4i32){0, 0, 0, 0};
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sane_r[r] = __shfl(ok, 4 * g + r, 64) != 0;     // item g + 4 r = permuted row 4 g + r
    }

    double key[4][NMAX];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < NMAX; ++i) key[r][i] = key_empty();
    [[maybe_unused]] float gate_f[4];
    [[maybe_unused]] double gate_d[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        gate_f[r] = __builtin_inff();
        gate_d[r] = __builtin_bit_cast(double, (uint64_t)BAZ_KEY_EMPTY_BITS | 0xFFFFFull);
    }
    const bool refine_on = rf.Gs != nullptr;
    const double lit_below = refine_on ? rf.below : -1.0;
    const float tacc_f = ip.t_acc_f;
    const double tacc_d = ip.t_acc;

    // ---- table staging: L2 -> LDS directly, 1 KiB per wave instruction ----------------------------------------------
    auto stage_load = [&](const uint32_t st, const int p, const int b) {
        const uint4* __restrict__ src = IB + ((size_t)st * 4 + (size_t)p * TPP) * TU + lane;
#pragma unroll
        for (int i = 0; i < (CH + 3) / 4; ++i) {
            const int j = i * 4 + wave;                                    // wave-uniform: chunk j of the phase
            if (j < CH)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 64),
                                                 (__attribute__((address_space(3))) void*)(&stage[b][j * 64]), 16, 0, 0);
        }
    };

    int buf = 0;
    v4f32 sv[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float* __restrict__ spec_base = SPEC ? spec + (size_t)item0 * res : nullptr;
    [[maybe_unused]] __amdgpu_buffer_rsrc_t spec_rsrc = __builtin_amdgcn_make_buffer_rsrc(spec_base, 0, 0x7FFFFFFF, 0x00020000);
    uint32_t soff[4];
    bool row_ok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        soff[r] = ((uint32_t)(g + 4 * r) * res + 4u * (uint32_t)c) * 4u;
        row_ok[r] = (item0 + (uint32_t)(g + 4 * r)) < batch;
    }
    uint32_t refined = 0, fell = 0;
    [[maybe_unused]] float worst = 0.0f;

    if (st_begin < st_end) stage_load(st_begin, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (uint32_t st = st_begin; st < st_end; ++st) {
        v4f64 acc[4];
        const uint32_t bin = st * 64 + 4 * (uint32_t)c;          // this lane's first bin of the step
#pragma unroll
        for (int p = 0; p < PPS; ++p) {
            const bool last_p = (p == PPS - 1);
            const bool more = !last_p || (st + 1 < st_end);      // wave-uniform
            if (more) stage_load(last_p ? st + 1 : st, last_p ? 0 : p + 1, buf ^ 1);

            // the phase's tiles: NKB NS (NS + 1) / 2 int8 MFMAs each, then the levels of every value combined in fp64
            const v4i32* __restrict__ Bp = reinterpret_cast<const v4i32*>(&stage[buf][0]) + lane;
#pragma unroll
            for (int tl = 0; tl < TPP; ++tl) {
                const int t = p * TPP + tl;
                v4i32 L[NS];
#pragma unroll
                for (int l = 0; l < NS; ++l) L[l] = (v4i32){0, 0, 0, 0};
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    v4i32 b[NS];
#pragma unroll
                    for (int s = 0; s < NS; ++s) b[s] = Bp[((tl * NKB + kb) * NS + s) * 64];
                    // level by level across the digits of q: consecutive MFMAs go to different accumulators
#pragma unroll
                    for (int s = 0; s < NS; ++s)
#pragma unroll
                        for (int l = s; l < NS; ++l)
                            L[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[kb][s], b[l - s], L[l], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int a[NS] = {L[0][r], L[1][r], L[2][r], L[3][r], L[4][r]};
                    acc[t][r] = I8Comb<MM, 0>::run(a, ip.wt);
                }
            }

            if (last_p) {
#pragma unroll
                for (int r = 0; r < 4; ++r) asm volatile("" ::"v"(sv[r]));      // (see scan_mfma_kernel: the store data stays put)
                bool hit = false, low = false;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (SPEC) {
                        float fd[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) fd[t] = (float)acc[t][r];
                        float mn;
                        asm("v_min3_f32 %0, |%1|, |%2|, |%3|" : "=v"(mn) : "v"(fd[0]), "v"(fd[1]), "v"(fd[2]));
                        asm("v_min_f32 %0, %1, |%2|" : "=v"(mn) : "v"(mn), "v"(fd[3]));
                        hit |= (mn <= gate_f[r]);
                        low |= (mn <= tacc_f);
#pragma unroll
                        for (int t = 0; t < 4; ++t) sv[r][t] = __builtin_amdgcn_rcpf(fabsf(fd[t]));
                    } else {
                        double m01, m23, mn;
                        asm("v_min_f64 %0, |%1|, |%2|" : "=v"(m01) : "v"(acc[0][r]), "v"(acc[1][r]));
                        asm("v_min_f64 %0, |%1|, |%2|" : "=v"(m23) : "v"(acc[2][r]), "v"(acc[3][r]));
                        mn = vmin64(m01, m23);
                        hit |= (mn <= gate_d[r]);
                        low |= (mn <= tacc_d);
                    }
                }
                hit |= low;
                if (VAL || __any(hit)) {
                    if (VAL || __any(low)) {
                        // some value of the step is too small for the integer form's bound: the whole step in the fp64 form
                        ++fell;
                        bool lit = false;
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const v4f64 ex = exact16<M>(Qs, FB, itn, g, lane, qstride, st, t);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                if constexpr (VAL) {
                                    const float ratio = (float)(fabs(acc[t][r] - ex[r]) / ip.e_bound);
                                    if (sane_r[r] && row_ok[r] && bin + t < res && ratio > worst) worst = ratio;   // (NaN never counts)
                                }
                                // Per VALUE: only a d_int at or below T is replaced, so what an (item, bin) pair gets never
                                // depends on which items share its wave or on how a batch was cut (the rule of literal_tile()).
                                const bool take = VAL || !(fabs(acc[t][r]) > tacc_d);
                                acc[t][r] = take ? ex[r] : acc[t][r];
                                lit |= take && (fabs(ex[r]) <= lit_below);
                            }
                        }
                        if (refine_on && __any(lit)) {          // near-null values: the reference's literal form (scan_mfma_kernel)
                            const v2f64* __restrict__ tbl = reinterpret_cast<const v2f64*>(rf.TB) + lane;
                            refined += literal_tile<M>(acc, rf, tbl, st, itn, g, qstride, (int)M - (int)n, bin, res, row_ok);
                        }
                        if constexpr (SPEC) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
#pragma unroll
                                for (int t = 0; t < 4; ++t) sv[r][t] = strength_f32(fabs(acc[t][r]));
                        }
                    }
                    const uint32_t nobin = ~keep_mask;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            key_insert_new<NMAX>(key[r], make_key(acc[t][r], (bin + t < res) ? bin + t : nobin, keep_mask));
                        const uint64_t kb = __builtin_bit_cast(uint64_t, key[r][NMAX - 1]) | (uint64_t)(~keep_mask);
                        gate_d[r] = fmax(__builtin_bit_cast(double, kb), tacc_d);
                        gate_f[r] = (float)gate_d[r];
                    }
                }
            }

            // the next phase's operands have landed (and the PREVIOUS step's stores are done) ...
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // ... then this step's spectrum stores, then the barrier
            if (last_p) {
                if constexpr (SPEC) {
                    const int step_off = (int)(st * 256u);
                    if constexpr (VEC4) {
                        if (st * 64 + 64 <= res) {              // wave-uniform: whole step inside the row
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (row_ok[r]) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, sv[r]), spec_rsrc, (int)soff[r], step_off, (1 | 2 | 16));
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (row_ok[r] && bin < res) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, sv[r]), spec_rsrc, (int)soff[r], step_off, (1 | 2 | 16));
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const v4u32 u = __builtin_bit_cast(v4u32, sv[r]);
#pragma unroll
                            for (int t = 0; t < 4; ++t)
                                if (row_ok[r] && bin + t < res)
                                    __builtin_amdgcn_raw_buffer_store_b32(u[t], spec_rsrc, (int)(soff[r] + 4u * t), step_off, (1 | 2 | 16));
                        }
                    }
                }
            }
            __syncthreads();
            buf ^= 1;
        }
    }

    if (rf.count) {
#pragma unroll
        for (int msk = 1; msk < 64; msk <<= 1) refined += __shfl_xor(refined, msk, 64);
        if (lane == 0 && refined) atomicAdd(rf.count, (unsigned long long)refined);
    }
    if (stat && lane == 0) {
        if (fell) atomicAdd(stat, (unsigned long long)fell);
        atomicAdd(stat + 1, (unsigned long long)(st_end - st_begin));
    }
    if constexpr (VAL) {
        unsigned int wb = __builtin_bit_cast(unsigned int, worst);
#pragma unroll
        for (int msk = 1; msk < 64; msk <<= 1) {
            const unsigned int o = __shfl_xor(wb, msk, 64);
            wb = o > wb ? o : wb;
        }
        if (lane == 0 && margin) atomicMax(margin, (unsigned long long)wb);     // ratio >= 0: its bits order like the value
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        key_merge_xor<NMAX>(key[r], 1);
        key_merge_xor<NMAX>(key[r], 2);
        key_merge_xor<NMAX>(key[r], 4);
        key_merge_xor<NMAX>(key[r], 8);
        const uint32_t it = item0 + (uint32_t)(g + 4 * r);
        if (c == 0 && it < batch) {
#pragma unroll
            for (int i = 0; i < NMAX; ++i) cand[((size_t)it * nsplit + split) * NMAX + i] = key[r][i];
        }
    }
}

}  // namespace bazmusic
