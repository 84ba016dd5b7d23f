// scan_i8_kernels.hip.h -- the pseudo-spectrum scan (lib/baz_music_doa.cc:101-121) for 6 .. 16 antennas on the INT8 matrix
// core: digit products accumulated exactly in int32, five digits per operand for the bulk of the (item, bin) values, seven
// where an a-priori error bound says so (and the reference's literal form on the fp64 matrix core near nulls, as before).
// gfx950 only.
//
// Why.  d(item, bin) = a^H Q a = sum_e q_e F_e over MM = m^2 real terms (music_kernels.hip.h 4.) cancels, so it cannot be
// evaluated in float32, and on v_mfma_f64_16x16x4 it is bound by the fp64 matrix rate: config 3 (m = 8, 36,000 bins) ran its
// scan at 77 % of the 78.6 TFLOP/s peak and 26 % of HBM (round 3).  v_mfma_i32_16x16x64_i8 multiplies 8-bit integers ~50 x
// faster and ACCUMULATES IN INT32 WITHOUT ROUNDING -- the only error of an integer form is the one made when the operands
// are cut into digits, and that has a bound that needs no statistics (the Ozaki scheme, with integer slices).
//
// Integer form.  Both operands in fixed point, cut into ND = 7 balanced base-256 digits (int8), most significant first:
//     Qi_e = rint(q_e 2^54)            |q_e| <= 1 for a projector (|Q_ij| <= 1/2 off the diagonal, q = 2 Re / -2 Im)
//     Fi_e = rint(F_e 2^54 / Fscale)   Fscale = the power of two with max|F| / Fscale in (1/2 (1 + 2^-10), 1 + 2^-10]
//     X = sum_s x_s 256^(6 - s),  x_s in [-128, 127] (s >= 1),  |x_0| <= 65
// The product sum_e Qi_e Fi_e is the sum over digit pairs (s, t) of 256^(12 - s - t) A_st, A_st = sum_e q_es F_et: one K = 64
// int8 MFMA per pair, 16-item x 16-bin tile and block of 64 terms, accumulated per LEVEL l = s + t in one int32 accumulator
// (|A| <= (l + 1) MM 2^14 < 2^31).  The BULK form keeps the levels l < NS = 5 -- 15 MFMAs per tile and block, where the fp64
// form issues 16 fp64 MFMAs of 4 x the cycles each --, i.e. it multiplies the operands' five leading digits.  Levels are
// combined per value in fp64 (every partial sum is an integer below 2^53 times a power of two: exact), d5 = H * 2^-12 Fscale.
//
// Error bound of the bulk form (a priori, in units of d):
//     digits cut off    five leading BALANCED digits of seven: |q_e - Q5_e / 2^38| <= 0.502 2^-38, the same for F / Fscale
//                                                                                          ->  <= MM Fscale 2^-38 1.006
//     levels dropped    pairs (s, t) of the five digits with s + t >= 5: <= MM 128^2 4 256^3 1.005 of the integer product
//                                                                                          ->  <= MM Fscale 4 2^-38 1.005
//     level 4 cut       the bulk form keeps floor(A_4 / 256) of the level-4 sum, folded into the level-3 accumulator (below):
//                                                                                          ->  <= Fscale 2^-36
//     E5 = MM Fscale 5 1.01 2^-38 + Fscale 2^-36   (m = 8: 1.2e-9 Fscale; m = 16: 4.7e-9 Fscale;  ||a||^2 = m for the helper's tables)
// A value is within eps = 7.5e-7 of the true one when d5 > T = E5 (1 + 1 / eps).
//
// Two tiers (round 4, fourth form).  The matrix core and the vector unit do not overlap on this kernel (their times add:
// profiles/r04_i8_scan_ablations.txt) and an int8 MFMA costs what ~5 vector instructions cost, so the 15 MFMAs were the
// largest share of a tile.  FIRST the 10 pairs of the four leading digits with s + t < 4: d4 = (A_0 256 + A_1) 65536 + A_2 256 +
// A_3 in units of wt[3], error <= E4 = MM Fscale 4 1.01 2^-30 (m = 8: 2.4e-7 Fscale), good to eps where |d4| > T4 = E4 (1 + 1 / eps)
// = 0.32 Fscale: 95 % of the values of a two-emitter scene.  A tile in which some value is at or below max(T4, its row's top-n
// gate) adds the 5 level-4 pairs and gives the values at or below T4 -- per value, as always -- the five-digit form d5 below.
//
// Bulk form per value (third form: the vector unit, not the matrix core, bounds this kernel -- 109 vector
// instructions per tile against 15 MFMAs in the second form, profiles/r04_i8_scan_v2_pmc.txt):  V = (A_0 256 + A_1) 65536 +
// (A_2 256 + A_3 + floor(A_4 / 256)), an integer below 2^48 built from two int32 words, two conversions and one fp64 FMA;
// d5 = V wt[3] exactly; (float) V, one multiplication by the power of two (float) wt[3], ONE comparison against the row's
// threshold max(top-n gate, T), one reciprocal.  Everything else -- the seven-digit form (which adds the cut-off low byte of
// A_4 back: d7 = d5 + (A_4 mod 256) wt[4] + A_5 wt[5] + A_6 wt[6]), bins outside the table (zero digits: V = 0 is under every
// threshold), items that are not projectors (zero digits), the literal form near nulls, the key network -- happens behind
// that one wave-uniform branch.
//
// Refined form.  A 16 x 16 tile in which some |d5| <= T (the bottom of the nulls: 0.5 % of the values of a 20-dB scene) adds
// the levels 5 and 6 of ALL SEVEN digits -- 13 more MFMAs on top of the five accumulators it still holds; digits 5, 6 of q
// from the lane's LDS slot, of F from a second image in L2 -- and the VALUES at or below T take d7 = d5 + (level 5, 6 sums):
// per value, so an (item, bin) pair's bits do not depend on its wave-mates.  |d7 - d| <= MM Fscale 7.07 2^-54 + 2^-53 |d|
// (digits; one rounding of the final sum): m = 8: 2.5e-14, the accuracy class of the fp64 form itself (m^2 eps ||a||^2 in
// the worst case) -- so the refined form REPLACES it: at `refine_below` = m 1e-8 max||a||^2, where scan_mfma_kernel's
// literal-form refinement of near-null values takes over (T exceeds it by orders of magnitude), d7 is good to 4e-8 relative.
// Bins outside the table get a huge d (never selected, never stored); items whose coefficients are not a projector's
// (non-finite or garbage covariance) have zero digits and take scan_mfma_kernel's own fp64 instruction sequence for their
// rows (exact16: same operands, same k order, same bits).  Everything downstream of d -- (float) conversion, v_rcp_f32, the
// store pattern, the gated top-n network, literal_tile(), the candidate lists -- is scan_mfma_kernel's.  Two bins can change
// places in the top-n list against the fp64 scan only where their strengths agree to 2 eps = 1.5e-6 (the parity rule allows
// 2e-5), and lvl[i] == spectrum[bin_i] holds bit for bit (the merge reads lvl back from the spectrum).
// tests/lab/i8_split_study.py restates the scheme in numpy (worst observed error 0.04 - 0.09 E5), tests/test_i8_scan.py pins
// the host-side images against it, and the VAL instantiation (baz_music_debug_i8_margin) evaluates all three forms on EVERY
// (item, bin) of a batch on the hardware: worst |d5 - d| / E5 and worst |d7 - d| / allowance.
//
// Layouts.  int8 / f16 MFMA C/D: col = lane & 15, row = 4 (lane >> 4) + reg; f64 MFMA: row = (lane >> 4) + 4 reg.  The int8 A
// operand therefore carries item pi(i) = (i >> 2) + 4 (i & 3) in row i, so that register r of lane (g, c) is item g + 4 r in
// BOTH forms (the trick of scan_coarse_kernels.hip.h).  A / B operand of the K = 64 form: lane (g, c) holds row / column c,
// k = 16 g + j in byte j of its 16 bytes (the same map on both sides: any consistent one gives the same dot product).
// Table images (built at set_table, build_i8_image): IB = [64-bin step][tile t][block kb][digit 0 .. 4][lane] x 16 B, IB2 the
// same with [digit 5, 6]; tile t of a step carries the bins 64 st + 4 c + t in its columns like FB (a lane ends up with 4
// consecutive bins: one 16-B store).  A phase = TPP tiles of IB (20 KiB at m <= 8 and at m >= 12) goes L2 -> LDS by
// global_load_lds_dwordx4, double-buffered, shared by the 4 waves of a workgroup; a wave owns 16 items and a range of steps,
// like scan_mfma_kernel (no row classes: m >= 6).
#pragma once

#include "scan_coarse_kernels.hip.h"      // (music_kernels.hip.h, literal16)

namespace bazmusic {

typedef int v4i32 __attribute__((ext_vector_type(4)));

constexpr int I8_NS = 5;                                           // digits of the bulk form = levels it keeps
constexpr int I8_ND = 7;                                           // digits of the refined form (operands are cut into these)
constexpr int i8_nkb(int m) { return (m * m + 63) / 64; }          // blocks of 64 terms
constexpr int i8_tpp(int m) { return i8_nkb(m) == 1 ? 4 : (i8_nkb(m) == 2 ? 2 : 1); }     // tiles per staged phase
constexpr int i8_tile_units(int m) { return i8_nkb(m) * I8_NS * 64; }                      // 16-B units per 16-bin tile (digits 0 .. 4)
constexpr int i8_tile_units2(int m) { return i8_nkb(m) * (I8_ND - I8_NS) * 64; }           // ... of the refinement digits (5, 6)
constexpr double I8_QMAX = 1.0009765625;                           // |q_e| <= 1 + 2^-10, else the item takes the fp64 form throughout
constexpr double I8_EPS = 7.5e-7;                                  // relative accuracy promised for values that keep the bulk form

struct I8Params {
    double wt[I8_ND];     // wt[l] = Fscale 2^(-12 - 8 l): weight of the level-l sum of digit products
    double sq;            // 2^(8 ND - 2): the fixed-point scale of q
    double t_acc;         // T = E5 (1 + 1 / eps): at or below it a value takes the refined form
    double e_bound;       // E5 (VAL only)
    double e_refined;     // allowance of the refined form against the fp64 form (VAL only)
    float ws_f;           // (float) wt[3], a power of two: |d5| as float = |(float) V| ws_f, exactly
    float t_acc_f;        // the smallest float >= T: |d5| <= T  =>  (float)|d5| <= t_acc_f
    float t4_f;           // the smallest float >= T4 = E4 (1 + 1 / eps): a value keeps the four-digit form iff (float)|d4| > t4_f
    double e4_bound;      // E4 (VAL only)
};

// two adjacent levels fit one int32 (a_l 256 + a_(l+1)) while (l + 1) MM 2^22 + (l + 2) MM 2^14 < 2^31
constexpr bool i8_pair_ok(int mm, int l) { return (long long)(l + 1) * mm * 4194304ll + (long long)(l + 2) * mm * 16384ll < 2147483648ll; }

template <int MM, int L>
struct I8Comb {
    static __device__ __forceinline__ double run(const int (&a)[I8_NS], const double (&wt)[I8_ND])
    {
        if constexpr (L >= I8_NS) return 0.0;
        else if constexpr (L + 1 < I8_NS && i8_pair_ok(MM, L))
            return __builtin_fma((double)(a[L] * 256 + a[L + 1]), wt[L + 1], I8Comb<MM, L + 2>::run(a, wt));
        else return __builtin_fma((double)a[L], wt[L], I8Comb<MM, L + 1>::run(a, wt));
    }
};

// scan_mfma_kernel's projector GEMM for ONE 16-item x 16-bin tile (tile t of step st): same operands, same k order, hence
// the same bits.  Both operands come from L2 (q of the natural row c, FB where the image has it).  Not inlined: the ordinary
// steps do not pay its registers.  Used for items whose coefficients are not a projector's, and by the validation build.
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

// The float32 combinations of the level sums (signed, in units of d; round 5): hw = A_0 256 + A_1 and lw = A_2 256 + A_3 (+ A_4 / 256) as int32
// words, two full-rate conversions and one f32 FMA -- where the fourth form spent two fp64 conversions, an fp64 FMA and an fp64 -> f32
// conversion per value.  |result - V wt[3]| <= (2^-23 |V| + 128) wt[3] and V >= T4 / wt[3] ~ 2^33 (five digits: T / wt[3] ~ 2^27, |lw| < 2^28: + 8)
// for a value that keeps the form: 1.3e-7 relative on top of the 7.5e-7 the form promises for d.  PAIR23: lw fits an int32 (m <= 13).
template <bool PAIR23>
__device__ __forceinline__ float i8_comb4(const int l0, const int l1, const int l2, const int l3, const float ws)
{
    const float hw = (float)(l0 * 256 + l1);
    float lw;
    if constexpr (PAIR23) lw = (float)(l2 * 256 + l3);
    else lw = __builtin_fmaf((float)l2, 256.0f, (float)l3);
    return __builtin_fmaf(hw, 65536.0f, lw) * ws;
}
template <bool PAIR23>
__device__ __forceinline__ float i8_comb5(const int l0, const int l1, const int l2, const int l3, const int l4, const float ws)
{
    const float hw = (float)(l0 * 256 + l1);
    float lw;
    if constexpr (PAIR23) lw = (float)(l2 * 256 + l3 + (l4 >> 8));
    else lw = __builtin_fmaf((float)l2, 256.0f, (float)l3) + (float)(l4 >> 8);
    return __builtin_fmaf(hw, 65536.0f, lw) * ws;
}

// Order of the first tier's 10 digit pairs (s = digit of q, l = level = s + digit of F): the levels go round so that two
// MFMAs on one accumulator are apart (a dependent MFMA waits for its predecessor's passes).  The second tier is the 5 pairs
// of level 4, s = 0 .. 4.
constexpr int i8_t1_s(int i) { constexpr int t[10] = {0, 0, 0, 0, 1, 1, 1, 2, 2, 3}; return t[i]; }
constexpr int i8_t1_l(int i) { constexpr int t[10] = {0, 1, 2, 3, 1, 2, 3, 2, 3, 3}; return t[i]; }

// VAL: validation build (baz_music_debug_i8_margin): every tile runs the bulk AND the refined form, every step the fp64 form;
// margin[0] / margin[1] / margin[2] = the worst |d5 - d| / E5, |d7 - d| / allowance and |d4 - d| / E4 over all (item, bin) of items
// that take the integer forms (float bits, atomicMax); outputs are the fp64 form's.
// stat (may be nullptr): [0] += wave tiles (16 items x 16 bins) that ran the refined form, [1] += wave tiles walked.
// ABL (lab builds only; timing, results are wrong): 1 no spectrum stores, 4 no MFMAs, 8 only the first phase is staged (no
// further staging loads, waits or barriers: every phase reads the first one's operands),
// 32 no tile arithmetic at all (staging, barriers and stores of a constant), 64 plain stores (no nt / sc bits), 128 no per-value
// work (the accumulators are only kept alive), 256 no MFMAs and nothing in their place (accumulators = the staged operands),
// 512 the 10 MFMAs of levels 0 .. 3 only.
template <int M, int NMAX, bool SPEC, bool VEC4, bool VAL = false, int ABL = 0>
__global__ __launch_bounds__(256, 2) void scan_i8_kernel(const double* __restrict__ Qs, const uint4* __restrict__ IB,
                                                         const uint4* __restrict__ IB2, const double2* __restrict__ FB,
                                                         float* __restrict__ spec, double* __restrict__ cand, uint32_t batch,
                                                         uint32_t res, uint32_t qstride, uint32_t nsplit, uint32_t keep_mask,
                                                         uint32_t n, ScanRefine rf, I8Params ip,
                                                         unsigned long long* __restrict__ stat,
                                                         unsigned long long* __restrict__ margin)
{
    constexpr int MM = M * M;
    constexpr int NS = I8_NS, ND = I8_ND;
    static_assert(I8_NS == 5 && I8_ND == 7, "the level lists below are written out for five + two digits");
    constexpr int NKB = i8_nkb(M);
    constexpr int TPP = i8_tpp(M);
    constexpr int PPS = 4 / TPP;                       // phases per 64-bin step
    constexpr int TU = i8_tile_units(M);               // 16-B units per tile
    constexpr int TU2 = i8_tile_units2(M);
    constexpr int CH = TPP * NKB * NS;                 // 1-KiB chunks per phase
    static_assert(M >= 6 && M <= 16, "6 <= m <= 16 (row classes below, run-time-m kernels above)");
    // ONE __shared__ object: with a second one hipcc (ROCm 7.2) can no longer tell the LDS-DMA writes of the NEXT phase from
    // the ds_reads of this one and drains vmcnt(0) in front of every phase's first ds_read -- the staging loads it has just
    // issued AND the spectrum stores issued before the barrier (seen in the second form's .s; it cost 0.2 ms of config 3's 0.8)
    __shared__ uint4 lds_all[2 * CH * 64 + 4 * NKB * (ND - NS) * 64];
    uint4 (*stage)[CH * 64] = reinterpret_cast<uint4 (*)[CH * 64]>(&lds_all[0]);
    // digits 5, 6 of q, per wave and LANE (each lane reads back its own)
    v4i32 (*a56)[NKB][ND - NS][64] = reinterpret_cast<v4i32 (*)[NKB][ND - NS][64]>(&lds_all[2 * CH * 64]);
    // Spectrum stores and the tiles' arithmetic (profiles/r04_i8_scan_ablations.txt): the stores alone run at 5.3 TB/s (0.44 ms for
    // config 3), the arithmetic alone takes 0.53 ms, together 0.73 -- a wave blocks at a store until the write path accepts it,
    // and three waves per SIMD cover that as far as a closed queue of three customers does.  What did NOT change it, each built
    // and measured: the stores in flight across the next step's wait (vmcnt(4) behind the staging loads; vmcnt retires in issue
    // order -- hipcc's own counted waits rely on it), one store behind every tile of the next step from a second register set,
    // workgroups started a fraction of a step apart, 256-B aligned rows, plain instead of nt / sc stores.

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
    bool any_insane;
    {
        const uint32_t it_p = item0 + (uint32_t)((c >> 2) + 4 * (c & 3));    // permuted row c
        const uint32_t itp = (it_p < batch) ? it_p : (batch - 1);
        int ok = 1;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            v4i32 lo[ND - NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) A[kb][s] = (v4i32){0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < ND - NS; ++s) lo[s] = (v4i32){0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int e = 64 * kb + 16 * g + j;
                double qv = Qs[(size_t)(e < MM ? e : 0) * qstride + itp];    // (unconditional load, clamped: no branch per value)
                qv = (e < MM) ? qv : 0.0;
                const bool fine = fabs(qv) <= I8_QMAX;                       // false for NaN
                ok &= fine ? 1 : 0;
                qv = fine ? qv : 0.0;
                double r = __builtin_rint(qv * ip.sq);                       // |r| <= 2^54 (1 + 2^-10): every step below is exact
#pragma unroll
                for (int s = ND - 1; s >= 1; --s) {
                    const double h = __builtin_floor(__builtin_fma(r, 0x1p-8, 0.5));     // floor((r + 128) / 256)
                    const int dg = (int)__builtin_fma(-256.0, h, r);                     // in [-128, 127]
                    const int w = (int)((unsigned)(dg & 255) << (8 * (j & 3)));
                    if (s >= NS) lo[s - NS][j >> 2] |= w;
                    else A[kb][s][j >> 2] |= w;
                    r = h;
                }
                A[kb][0][j >> 2] |= (int)((unsigned)((int)r & 255) << (8 * (j & 3)));
                // (one word of digits at a time: without the fence the scheduler hoists all 16 NKB loads of the prologue to
                // its top and the 12 .. 15-antenna instantiations spill 150 .. 490 registers)
                if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s = 0; s < ND - NS; ++s) a56[wave][kb][s][lane] = lo[s];
        }
        ok &= __shfl_xor(ok, 16, 64);                  // the 4 lanes (g = 0 .. 3) that hold the row
        ok &= __shfl_xor(ok, 32, 64);
        if (!ok) {                                     // not a projector's coefficients: zero digits; the row takes the fp64 form
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
                for (int s = 0; s < NS; ++s) A[kb][s] = (v4i32){0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < ND - NS; ++s) a56[wave][kb][s][lane] = (v4i32){0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sane_r[r] = __shfl(ok, 4 * g + r, 64) != 0;     // item g + 4 r = permuted row 4 g + r
        any_insane = __any(!ok);
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
    [[maybe_unused]] const float below_f = refine_on ? (float)rf.below : -1.0f;
    [[maybe_unused]] const double below_d = refine_on ? rf.below : -1.0;
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
    float* __restrict__ spec_base = SPEC ? spec + (size_t)item0 * ((ABL & 2048) ? (res & ~63u) : res) : nullptr;
    [[maybe_unused]] __amdgpu_buffer_rsrc_t spec_rsrc = __builtin_amdgcn_make_buffer_rsrc(spec_base, 0, 0x7FFFFFFF, 0x00020000);
    uint32_t soff[4];
    bool row_ok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        soff[r] = ((uint32_t)(g + 4 * r) * ((ABL & 2048) ? (res & ~63u) : res) + 4u * (uint32_t)c) * 4u;    // (2048, lab: rows 256-B aligned)
        row_ok[r] = (item0 + (uint32_t)(g + 4 * r)) < batch;
    }
    uint32_t refined = 0, fell = 0;
    [[maybe_unused]] float worst5 = 0.0f, worst7 = 0.0f, worst4 = 0.0f;

    if constexpr ((ABL & 1024) != 0) {      // lab: workgroups start up to ~one step apart (their spectrum stores no longer arrive in one burst)
        const uint32_t frac = ((blockIdx.x * 0x9E3779B1u) >> 28) & 15u;          // 0 .. 15 sixteenths of ~8,000 cycles
        for (uint32_t i = 0; i < frac; ++i) __builtin_amdgcn_s_sleep(8);        // 8 x 64 cycles
    }
    if (st_begin < st_end) stage_load(st_begin, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const uint32_t nobin = ~keep_mask;
    const int nn = (int)M - (int)n;
    constexpr bool PAIR23 = i8_pair_ok(MM, 2);         // A_2 256 + A_3 (+ A_4 / 256) fits an int32 (m <= 13)
    const double ws_d = ip.wt[NS - 2];                 // weight of the integer V of both bulk forms (level 3)
    const float ws_f = ip.ws_f;
    const float t4_f = ip.t4_f;
    // the row's thresholds of the ONE comparison per value and tier: at or under thr4 a four-digit value needs the second tier
    // (T4) or may enter the row's list (top-n gate), or lies outside the table / belongs to an item without digits (V = 0);
    // at or under thr5 a five-digit value needs the seven-digit form (T) or may enter the list
    // (round 5) the floats compared with them are float32 COMBINATIONS, good to 1.3e-7 of the exact value: the thresholds carry that slack
    // (the gate and T times 1 + 2^-21), so no value the exact path would have wanted is ever kept from it; the exact path decides in fp64
    constexpr float SLACK = 1.0f + 0x1p-21f;
    float thr4[4], thr5[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) thr4[r] = thr5[r] = __builtin_inff();
    // this wave's 16 rows x 64 bins of step `sst`: a 16-B store per lane and row, 256 B contiguous per row
    auto store_rows = [&](const uint32_t sst) __attribute__((always_inline)) {
        const uint32_t sbin = sst * 64 + 4 * (uint32_t)c;
        const bool stail = sst * 64 + 64 > res;
        if constexpr (SPEC && !(ABL & 1)) {
            const int step_off = (ABL & 4096) ? 0 : (int)(sst * 256u);       // (4096, lab: every step overwrites the row's first piece)
            if constexpr (VEC4) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (!stail) {                       // wave-uniform: whole step inside the row
                        if (row_ok[r]) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, sv[r]), spec_rsrc, (int)soff[r], step_off, (ABL & 64) ? 0 : ((ABL & 16384) ? 16 : ((ABL & 32768) ? (1 | 16) : ((ABL & 65536) ? 2 : (1 | 2 | 16)))));
                    } else {
                        if (row_ok[r] && sbin < res) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, sv[r]), spec_rsrc, (int)soff[r], step_off, (1 | 2 | 16));
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const v4u32 u = __builtin_bit_cast(v4u32, sv[r]);
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        if (row_ok[r] && sbin + t < res)
                            __builtin_amdgcn_raw_buffer_store_b32(u[t], spec_rsrc, (int)(soff[r] + 4u * t), step_off, (1 | 2 | 16));
                }
            }
        } else if constexpr (SPEC) {
#pragma unroll
            for (int r = 0; r < 4; ++r) asm volatile("" ::"v"(sv[r]));
        }
    };
    [[maybe_unused]] long long t_wait = 0, t_store = 0, t_bar = 0;       // (8192, lab: s_memtime around the step end's three parts)
    [[maybe_unused]] const long long t_loop0 = ((ABL & 8192) != 0) ? (long long)__builtin_readcyclecounter() : 0;
    // STRIDED WALK (round 5).  The steps of the range are visited in SW interleaved sweeps (st_begin + k, + SW, + 2 SW, ...; k = 0 .. SW - 1)
    // instead of left to right.  Nothing depends on the order -- every step stages its own operands and stores its own 256-B pieces, and
    // the lists order their keys by (d, bin) whatever order they arrive in -- but the top-n gate does: walking left to right, EVERY tile on
    // the way down to the range's first null holds a new minimum and takes the exact path; after one coarse sweep the lists already hold
    // values from near every null, and the later sweeps take the exact path only where they pass a null's bottom.
    const uint32_t nst = st_end - st_begin;
    const uint32_t SW = (ABL & 256) ? 1u : (nst >= 64u ? 8u : (nst >= 16u ? 4u : (nst >= 6u ? 2u : 1u)));
    uint32_t st = st_begin, sweep = 0;
    for (uint32_t it = 0; it < nst; ++it) {
        uint32_t st_next = st + SW, sweep_next = sweep;          // the step after this one in walk order (wave-uniform)
        if (st_next >= st_end) { sweep_next = sweep + 1; st_next = st_begin + sweep_next; }
        const bool has_next = it + 1 < nst;
        const uint32_t bin = st * 64 + 4 * (uint32_t)c;          // this lane's first bin of the step (tile t: bin + t)
        const bool tail_step = st * 64 + 64 > res;               // wave-uniform: the step reaches beyond the table
#pragma unroll
        for (int p = 0; p < PPS; ++p) {
            const bool last_p = (p == PPS - 1);
            const bool more = !last_p || has_next;               // wave-uniform
            if constexpr (!(ABL & 8)) {
                if (more) stage_load(last_p ? st_next : st, last_p ? 0 : p + 1, buf ^ 1);
            }
            if constexpr (ABL & 32) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sv[r] = (v4f32){1.0f, 2.0f, 3.0f, (float)st};
            }
            if (p == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) asm volatile("" ::"v"(sv[r]));      // (see scan_mfma_kernel: the store data stays put)
            }

            const v4i32* __restrict__ Bp = reinterpret_cast<const v4i32*>(&stage[(ABL & 8) ? 0 : buf][0]) + lane;
#pragma unroll
            for (int tl = 0; tl < ((ABL & 32) ? 0 : TPP); ++tl) {
                const int t = p * TPP + tl;
                // ---- first tier: the 10 NKB MFMAs of the four leading digits, levels 0 .. 3 -------------------------------
                v4i32 L[NS];
#pragma unroll
                for (int l = 0; l < NS; ++l) L[l] = (v4i32){0, 0, 0, 0};
                v4i32 b0[NS - 1];                       // (one block of terms: the second tier reuses these registers)
#pragma unroll
                for (int kb = 0; kb < NKB; ++kb) {
                    v4i32 b[NS - 1];
#pragma unroll
                    for (int s = 0; s < NS - 1; ++s) b[s] = Bp[((tl * NKB + kb) * NS + s) * 64];
#pragma unroll
                    for (int i = 0; i < 10; ++i) {
                        const int sq_ = i8_t1_s(i), l = i8_t1_l(i);
                        if constexpr (ABL & 4) L[l] += b[l - sq_] ^ A[kb][sq_];
                        else L[l] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[kb][sq_], b[l - sq_], L[l], 0, 0, 0);
                    }
                    if constexpr (NKB == 1) {
#pragma unroll
                        for (int s = 0; s < NS - 1; ++s) b0[s] = b[s];
                    }
                }
                // ---- four-digit form: the float32 combination of two int32 words, one comparison, one reciprocal per value ------------
                float fdv[4];
                unsigned long long under = 0ull;                           // lanes with a value at or below its row's threshold
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    fdv[r] = fabsf(i8_comb4<PAIR23>(L[0][r], L[1][r], L[2][r], L[3][r], ws_f));
                    under |= __builtin_amdgcn_ballot_w64(fdv[r] <= thr4[r]);
                    if constexpr (SPEC) sv[r][t] = __builtin_amdgcn_rcpf(fdv[r]);
                }
                if (VAL || under) {
                    // ======== second tier (wave-uniform branch): the 5 NKB MFMAs of level 4; the values at or below T4 take the
                    // five-digit form d5 = d4 + floor(A_4 / 256) wt[3], per value ============================================
                    bool form4[4];
                    unsigned long long need5 = 0ull;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        form4[r] = !VAL && (fdv[r] > t4_f);                  // per VALUE: its own four-digit value decides
                        need5 |= __builtin_amdgcn_ballot_w64(!form4[r]);
                    }
                    if (VAL || need5) {                 // (a tile that only holds candidates of the top-n lists skips the MFMAs)
#pragma unroll
                        for (int kb = 0; kb < NKB; ++kb) {
                            v4i32 b[NS];
#pragma unroll
                            for (int s = 0; s < NS; ++s) {
                                if (NKB == 1 && s < NS - 1) b[s] = b0[s];
                                else b[s] = Bp[((tl * NKB + kb) * NS + s) * 64];
                            }
#pragma unroll
                            for (int sq_ = 0; sq_ < NS; ++sq_) L[4] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[kb][sq_], b[4 - sq_], L[4], 0, 0, 0);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float f5 = fabsf(i8_comb5<PAIR23>(L[0][r], L[1][r], L[2][r], L[3][r], L[4][r], ws_f));
                            fdv[r] = form4[r] ? fdv[r] : f5;
                            if constexpr (SPEC) sv[r][t] = __builtin_amdgcn_rcpf(fdv[r]);
                        }
                    }
                    under = 0ull;
#pragma unroll
                    for (int r = 0; r < 4; ++r) under |= __builtin_amdgcn_ballot_w64(fdv[r] <= thr5[r]);
                if (VAL || under) {
                    // ======== everything that is not the bulk of the values (wave-uniform branch): from here on in fp64, exactly ========
                    v4f64 d;
                    [[maybe_unused]] v4f64 d4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int hw = L[0][r] * 256 + L[1][r];
                        double v4;
                        if constexpr (PAIR23) v4 = __builtin_fma((double)hw, 65536.0, (double)(L[2][r] * 256 + L[3][r]));
                        else v4 = __builtin_fma(__builtin_fma((double)hw, 256.0, (double)L[2][r]), 256.0, (double)L[3][r]);
                        d4[r] = v4 * ws_d;
                        d[r] = (form4[r] ? v4 : v4 + (double)(L[4][r] >> 8)) * ws_d;        // d4 or d5, exactly
                    }
                    [[maybe_unused]] const v4f64 d5 = d;
                    bool exact_r[4] = {false, false, false, false};         // values whose float is (float)|d| of an fp64 form below
                    // Five-digit values at or below T: two more digits of both operands (levels 5 and 6 on top of the accumulated
                    // ones: 13 NKB MFMAs) and the low byte of the level-4 sum; digits 5, 6 of q from this lane's LDS slot, of F
                    // from the image in L2.
                    // (rows without digits and bins outside the table have V = 0: they are replaced below -- exact16 / a huge d -- and
                    // must not send the wave through the 13 MFMAs, nor count as refined tiles: ADVICE r4)
                    bool lowt = false;
                    const bool in_table = bin + t < res;
#pragma unroll
                    for (int r = 0; r < 4; ++r) lowt |= (VAL || (sane_r[r] && in_table)) && !form4[r] && !(fabs(d[r]) > tacc_d);
                    if (VAL || __any(lowt)) {
                        ++fell;
                        v4i32 L5 = {0, 0, 0, 0}, L6 = {0, 0, 0, 0};
                        const v4i32* __restrict__ B2 = reinterpret_cast<const v4i32*>(IB2) + ((size_t)st * 4 + (size_t)t) * TU2 + lane;
#pragma unroll
                        for (int kb = 0; kb < NKB; ++kb) {
                            const v4i32 b5 = B2[(kb * 2 + 0) * 64], b6 = B2[(kb * 2 + 1) * 64];
                            const v4i32 a5 = a56[wave][kb][0][lane], a6 = a56[wave][kb][1][lane];
                            v4i32 b[NS];
#pragma unroll
                            for (int s = 0; s < NS; ++s) b[s] = Bp[((tl * NKB + kb) * NS + s) * 64];
                            L5 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[kb][0], b5, L5, 0, 0, 0);
                            L6 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[kb][0], b6, L6, 0, 0, 0);
                            L5 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[kb][1], b[4], L5, 0, 0, 0);
                            L6 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[kb][1], b5, L6, 0, 0, 0);
                            L5 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[kb][2], b[3], L5, 0, 0, 0);
                            L6 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[kb][2], b[4], L6, 0, 0, 0);
                            L5 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[kb][3], b[2], L5, 0, 0, 0);
                            L6 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[kb][3], b[3], L6, 0, 0, 0);
                            L5 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[kb][4], b[1], L5, 0, 0, 0);
                            L6 = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[kb][4], b[2], L6, 0, 0, 0);
                            L5 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a5, b[0], L5, 0, 0, 0);
                            L6 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a5, b[1], L6, 0, 0, 0);
                            L6 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a6, b[0], L6, 0, 0, 0);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            // the level sums are exact (small integers times powers of two); ONE rounding, of the final sum
                            const double low = __builtin_fma((double)(L[4][r] & 255), ip.wt[4],
                                                             __builtin_fma((double)L5[r], ip.wt[5], (double)L6[r] * ip.wt[6]));
                            const double d7 = d[r] + low;
                            // Per VALUE: only a five-digit d at or below T is replaced, so what an (item, bin) pair gets never depends
                            // on which items share its wave or on how a batch was cut (the rule of literal_tile()).
                            const bool take = VAL || (!form4[r] && !(fabs(d[r]) > tacc_d));
                            d[r] = take ? d7 : d[r];
                            exact_r[r] = take;
                        }
                    }
                    // items whose coefficients are not a projector's (non-finite or garbage covariance): scan_mfma_kernel's fp64
                    // form for their rows, bit for bit (a wave-uniform branch, taken by waves that hold such an item)
                    if (VAL || any_insane) {
                        const v4f64 ex = exact16<M>(Qs, FB, itn, g, lane, qstride, st, t);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if constexpr (VAL) {
                                const float r5 = (float)(fabs(d5[r] - ex[r]) / ip.e_bound);
                                const float r7 = (float)(fabs(d[r] - ex[r]) / (ip.e_refined + 0x1p-50 * fabs(ex[r])));
                                const float r4 = (float)(fabs(d4[r] - ex[r]) / ip.e4_bound);
                                const bool counts = sane_r[r] && row_ok[r] && bin + t < res;        // (NaN never counts)
                                if (counts && r5 > worst5) worst5 = r5;
                                if (counts && r7 > worst7) worst7 = r7;
                                if (counts && r4 > worst4) worst4 = r4;
                            }
                            d[r] = (VAL || !sane_r[r]) ? ex[r] : d[r];
                            exact_r[r] = exact_r[r] || VAL || !sane_r[r];
                        }
                    }
                    // bins outside the table (last step of a row): zero digits gave d = 0; they must never be selected
                    const bool inside = in_table;
                    if (tail_step) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            d[r] = inside ? d[r] : 1e300;
                            exact_r[r] = exact_r[r] || !inside;
                        }
                    }
                    // Top-n gate and near-null vote of scan_mfma_kernel, per tile -- in fp64 (exact: the float comparisons above only
                    // decided that this path runs).  A value that kept an integer form keeps its float32 combination as its float.
                    bool hit = false, low = false;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        hit |= (fabs(d[r]) <= gate_d[r]);
                        low |= (fabs(d[r]) <= below_d);
                        if constexpr (SPEC) sv[r][t] = __builtin_amdgcn_rcpf(exact_r[r] ? fabsf((float)d[r]) : fdv[r]);
                    }
                    if (__any(hit)) {
                        if (refine_on && __any(low)) {          // near-null values: the reference's literal form, per value
                            const v4f64 lit = literal16<M>(rf.Gs, rf.TB, itn, g, qstride, nn, bin + t);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const bool redo = (fabs(d[r]) <= rf.below) && inside;
                                d[r] = redo ? lit[r] : d[r];
                                refined += (redo && row_ok[r]) ? 1u : 0u;
                                if constexpr (SPEC) sv[r][t] = redo ? strength_f32(fabs(d[r])) : sv[r][t];   // (per VALUE: the others keep their float)
                            }
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            key_insert_new<NMAX>(key[r], make_key(d[r], inside ? bin + t : nobin, keep_mask));
                            const uint64_t kb = __builtin_bit_cast(uint64_t, key[r][NMAX - 1]) | (uint64_t)(~keep_mask);
                            gate_d[r] = fmax(__builtin_bit_cast(double, kb), below_d);
                            // the bulk paths' thresholds: at least the gate (rounded UP to a float, times the combination's slack), at least T4 / T
                            float gu = (float)gate_d[r];
                            gu = ((double)gu < gate_d[r]) ? __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, gu) + 1u) : gu;   // (gu >= 0, finite here)
                            gu *= SLACK;
                            thr4[r] = fmaxf(gu, t4_f);
                            thr5[r] = fmaxf(gu, ip.t_acc_f * SLACK);
                        }
                    }
                }
                }
            }

            [[maybe_unused]] long long tm0 = 0, tm1 = 0, tm2 = 0, tm3 = 0;
            if constexpr ((ABL & 8192) != 0) tm0 = __builtin_readcyclecounter();
            if constexpr (!(ABL & 8)) {
                // the next phase's operands have landed (and the PREVIOUS step's stores are done) ...
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if constexpr ((ABL & 8192) != 0) tm1 = __builtin_readcyclecounter();
            // ... then this step's spectrum stores, then the barrier
            if (last_p) store_rows(st);                                  // ... then this step's spectrum stores, then the barrier
            if constexpr ((ABL & 8192) != 0) tm2 = __builtin_readcyclecounter();
            // (a raw barrier: __syncthreads() would put a vmcnt(0) in front of it -- the LDS-DMA writes are LDS writes to the
            // compiler -- and wait for the stores just issued; this phase's ds_reads have been consumed by its MFMAs)
            if constexpr (!(ABL & 8)) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");           // (the next phase's ds_reads stay behind the barrier for the compiler too)
            }
            if constexpr ((ABL & 8192) != 0) {
                tm3 = __builtin_readcyclecounter();
                t_wait += tm1 - tm0; t_store += tm2 - tm1; t_bar += tm3 - tm2;
            }
            buf ^= 1;
        }
        st = st_next;
        sweep = sweep_next;
    }
    if constexpr ((ABL & 8192) != 0) {
        if (stat && lane == 0) {
            atomicAdd(stat + 4, (unsigned long long)t_wait);
            atomicAdd(stat + 5, (unsigned long long)t_store);
            atomicAdd(stat + 6, (unsigned long long)t_bar);
            atomicAdd(stat + 7, (unsigned long long)((long long)__builtin_readcyclecounter() - t_loop0));
        }
    }
    if (rf.count) {
#pragma unroll
        for (int msk = 1; msk < 64; msk <<= 1) refined += __shfl_xor(refined, msk, 64);
        if (lane == 0 && refined) atomicAdd(rf.count, (unsigned long long)refined);
    }
    if (stat && lane == 0) {
        if (fell) atomicAdd(stat, (unsigned long long)fell);
        atomicAdd(stat + 1, (unsigned long long)(st_end - st_begin) * 4ull);
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
        const uint32_t it = item0 + (uint32_t)(g + 4 * r);
        if (c == 0 && it < batch) {
#pragma unroll
            for (int i = 0; i < NMAX; ++i) cand[((size_t)it * nsplit + split) * NMAX + i] = key[r][i];
        }
    }
}

}  // namespace bazmusic
