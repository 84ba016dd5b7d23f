// scan_coarse_kernels.hip.h -- the pseudo-spectrum scan when ONLY the n strongest bins are wanted
// (lib/baz_music_doa.cc:97-99,120-121: the spectrum port is not wired -- music_doa_helper's default
// output_spectrum=False, python/music_doa_helper.py:49,61-64), m <= 8.  gfx950 only.
//
// The reference evaluates 1/||G^H a||^2 for every bin (.cc:103-121) and keeps the n largest (.cc:129-141).  Without the
// spectrum port the only observable is that top-n list, i.e. the n SMALLEST d(bin) = a^H Q a.  scan_mfma_kernel
// computes all res values of d on the fp64 matrix core (57,600 FMAs per cfg2 item) to throw all but n away; at 0.50 ms
// per 262,144 items it is bound by fp64 matrix issue (77 % of 78.6 TF), and the step (0.88 ms) sits at 30 % of the
// HBM-read roofline (review r2, weak 4).  Here every 16-item x 16-bin tile is first evaluated by a COARSE form on the
// f16 matrix core (16x the fp64 rate) whose error against the fp64 value is bounded rigorously, and only tiles that
// can still hold a top-n member run the exact fp64 form -- the same instructions on the same operands as
// scan_mfma_kernel, so the keys that reach the lists, hence ang / lvl, are BIT-IDENTICAL to the full scan
// (tests/test_gpu_parity.py::test_coarse_gated_scan_equals_the_full_scan).
//
// Coarse form.  d = sum_e q_e F_e over the MM = m^2 <= 16 real terms of the Hermitian form (music_kernels.hip.h 4.).
// Both operands are split into two f16 pieces of scaled values,
//     qs = q 2^10 = qh + ql + rq,      Fs = F FS = Fh + Fl + rF,     FS = the power of two with max|Fs| in [2^13, 2^14)
// and  c = sum_e (qh + ql) (Fh + Fl)  is TWO K = 32 f16 MFMAs with f32 accumulation on the same A operand:
// A = [qh | ql], B = [Fh | Fh], then B = [Fl | Fl].  f16 x f16 products are exact in f32.  Error budget, in units of d
// (divide by SC = 2^10 FS), with S = max|F| sum_e |q_e|:
//     representation   |rq| <= 2^-22 |qs| + 2^-14 (the second term covers a flushed subnormal ql),  same for rF
//                                                                                          ->  <= 1.5 2^-20 S
//     accumulation     <= 66 additions, each <= 2^-23 (truncation) of a partial sum <= S + thr   ->  <= 2^-17 (S + thr)
// so |c / SC - d| <= E := 2^-16 (S + D) with a factor ~2 to spare (D = the threshold d is compared with).  The bound
// is checked on hardware over every (item, bin) of random batches by the VAL instantiation
// (baz_music_debug_coarse_margin: worst observed |c/SC - d| / E; measured 0.007-0.009: the matrix core accumulates far
// more accurately than the worst case assumed).
// (A first version formed the third product with the legacy v_mfma_f32_16x16x16_f16.  Wherever hipcc -- ROCm 7.2,
// -amdgpu-mfma-vgpr-form -- gave that instruction a destination different from its C operand, registers 0 and 1 of its
// result were wrong on MI355X, with 2 wait states or with three other MFMAs in between; the K = 32 form has shown no
// such behaviour in either role.  tests/lab/coarse_dump.py, tests/lab/coarse_diff.py; the K = 32 pair is also faster:
// scripts/ubench_f16mfma.hip, 13.8 against 17.0 clocks per instruction at 2 waves per SIMD.)
//
// Gate.  A lane's list of item i holds n keys; its last entry K_n bounds the item's final n-th smallest key from
// above, and so does any other lane's.  A bin can enter the final list only if |d| <= D := (K_n | low bits), so only
// if  c <= thr := (D + E) SC.  thr enters the coarse MFMA as C = -thr (f32, rounded up): the instruction itself
// delivers c - thr, three v_min and one compare per 64 values decide the tile.  After an exact tile the new
// thresholds are shared over the 16 lanes of an item row (4 DPP row rotations on f32).  D never drops below
// `refine_below`, so near-null values (literal-form refinement, music_kernels.hip.h) are always candidates.
// NaN q (non-finite covariance): the row's allowance es is +inf (`sane` below), so every tile of the row's wave fires and the
// exact form decides: NaN never enters a list, the lists stay empty = (0, 0) pairs like the full scan.
//
// Layouts.  f16 MFMA C/D: col = lane & 15, row = 4 (lane >> 4) + reg;  f64 MFMA: row = (lane >> 4) + 4 reg.  The
// coarse A operand therefore carries item pi(i) = (i >> 2) + 4 (i & 3) in row i: register r of lane (g, c) is item
// g + 4 r and bin 16 tile + c in BOTH forms.  A wave owns RG row groups of 16 items and walks the bin tiles of its
// range; the 4 waves of a workgroup share the table images through a double-buffered LDS stage of TPP tiles.
// Table images, per tile: C (1,024 B) = Fh then Fl, each 32 x 8 f16: entry (gb, c), j = piece[bin 16 tile + c][e = 8 gb + j]
// (lane (g, c) reads entry (g & 1, c): the K = 32 B operand [F | F] holds every piece twice);  X (2,048 B) = the fp64
// operand, k-steps (0,1) as 64 x double2, then (2,3).  Two arrays, because the first pass stages C only.
#pragma once

#include "music_kernels.hip.h"

namespace bazmusic {

typedef _Float16 v4f16 __attribute__((ext_vector_type(4)));
typedef _Float16 v8f16 __attribute__((ext_vector_type(8)));

constexpr int CS_C_UNITS = 64;            // 16-B units of a tile's coarse operands per GROUP of 16 terms: 512 B (Fh) + 512 B (Fl)
constexpr int CS_X_UNITS = 128;           // ... of its fp64 operand per group: 2048 B (4 k-steps)
// m^2 <= 16 terms are one group.  5 <= m <= 8 (25 .. 64 terms): the exact form takes 4 cs_groups(m) fp64 k-steps, and the
// coarse form works in NG = cs_groups32(m) groups of 32 terms with the two pieces of q as SEPARATE A operands:
//     per group  qh Fh, ql Fh (one B operand [Fh of 32 terms], read from LDS once) and qh Fl   -- 3 NG MFMAs per tile,
// all on one accumulator; ql Fl (<= 2^-22 S) is left to the budget.  Per tile 2 NG KiB of LDS reads against 4 NG KiB for
// the [F | F] form, which made the LDS, not the matrix core, the bound (measured at m = 8: 0.52 ms against 0.26 ms of MFMA
// issue).  Budget: representation <= 1.75 2^-20 S, accumulation <= (96 NG + 3 NG) 2^-23 (S + thr)  ->  E = NG 2^-16 (S + D).
constexpr int CS_C32_UNITS = 128;         // 16-B units of a tile's coarse operands per group of 32 terms: 1 KiB (Fh) + 1 KiB (Fl)
constexpr int cs_groups(int m) { return (m * m + 15) / 16; }
constexpr int cs_groups32(int m) { return (m * m + 31) / 32; }
constexpr int cs_c_units(int m) { return m <= 4 ? CS_C_UNITS : cs_groups32(m) * CS_C32_UNITS; }    // per tile
constexpr int cs_ng(int m) { return m <= 4 ? 1 : cs_groups32(m); }                                  // the NG of the allowance

struct CoarseParams {
    float sc_up;        // SC (1 + 2^-16) rounded up: threshold scale, d units -> coarse units
    float es_factor;    // 2^-16 max|F| SC, rounded up: es = es_factor * sum_e |q_e|
    double sc;          // SC = 2^10 FS (VAL only)
    double fmax;        // max|F| (VAL only)
    int lazy;           // lab: 0 = recompute and share the thresholds after every exact tile
};

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}

// min over the 16 lanes of a DPP row (= the lanes that hold the same four items), result in every lane
__device__ __forceinline__ float row_allmin(float v)
{
    v = fminf(v, dpp_f32<0x128>(v));   // row_ror:8
    v = fminf(v, dpp_f32<0x124>(v));   // row_ror:4
    v = fminf(v, dpp_f32<0x122>(v));   // row_ror:2
    v = fminf(v, dpp_f32<0x121>(v));   // row_ror:1
    return v;
}

// The reference's literal form ||G^H a||^2 (.cc:110-119) for ONE 16-item x 16-bin tile: literal_tile()'s instruction
// sequence per value (k outer, Re / Im, k-steps inner, d += p p), B gathered from the TB image in its 64-bin-step
// order.  Rare path (near-null tiles, SNR >~ 55 dB): not inlined, so the ordinary tiles do not pay its registers.
template <int M>
__device__ __noinline__ v4f64 literal16(const double* __restrict__ Gs, const double2* __restrict__ TB, const uint32_t itc,
                                        const int g, const uint32_t qstride, const int nn, const uint32_t bin)
{
    constexpr int KS2 = (2 * M + 3) / 4;
    const uint32_t st = bin >> 6, w = bin & 63u, c4 = w >> 2, t = w & 3u;
    const double* __restrict__ tb = reinterpret_cast<const double*>(TB + ((size_t)st * KS2 * 2 + (t >> 1)) * 64 + (g * 16 + c4)) + (t & 1u);
    // Operands in batches (round 4): the table's KS2 values once, the eigenvector's 2 KS2 coefficients per k with UNCONDITIONAL
    // loads (clamped index, the select afterwards).  The form before -- `cond ? sgn * ga[..] : 0` in front of every MFMA -- made
    // hipcc branch around each load and wait for it: 4 nn KS2 dependent L2 round trips per tile, which an incoherent 60-dB batch
    // pays in a third of its steps.  Same MFMAs, same operands, same order: the same bits.
    double tbv[KS2];
#pragma unroll
    for (int s = 0; s < KS2; ++s) tbv[s] = tb[(size_t)s * 256];
    const double sgn1 = (g & 1) ? 1.0 : -1.0;             // Im c: -gi*ar (g even) / +gr*ai (g odd)
    v4f64 d = {0, 0, 0, 0};
    for (int k = 0; k < nn; ++k) {
        double x0[KS2], x1[KS2];
#pragma unroll
        for (int s = 0; s < KS2; ++s) {
            const bool in = 4 * s + g < 2 * M;
            const int ant = in ? 2 * s + (g >> 1) : 0;
            const double v0 = Gs[(size_t)((k * M + ant) * 2 + (g & 1)) * qstride + itc];
            const double v1 = Gs[(size_t)((k * M + ant) * 2 + ((g & 1) ^ 1)) * qstride + itc];
            x0[s] = in ? v0 : 0.0;
            x1[s] = in ? sgn1 * v1 : 0.0;
        }
        v4f64 p = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < KS2; ++s) p = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[s], tbv[s], p, 0, 0, 0);
        d += p * p;
        p = (v4f64){0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < KS2; ++s) p = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[s], tbv[s], p, 0, 0, 0);
        d += p * p;
    }
    return d;
}

// Ordering f32 values through their bit patterns as signed integers (v_min_i32 / v_min3_i32, no canonicalisation pass
// over MFMA results, and -- unlike an inline-asm v_min_f32 -- instructions the compiler pads against the MFMA's result
// hazard, cdna_hip_programming.md 5.7).  For non-NaN floats: bits <= 0 <=> value <= +0, and a negative value's bits are
// below every non-negative value's.  Among two negatives the order is reversed (the one closer to zero wins a min):
// where that matters here the result is only ever used as an UPPER bound.  +NaN patterns are large positive (never win
// a min, never pass "<= 0"); -NaN patterns pass "<= 0", which only means an extra exact tile.
__device__ __forceinline__ int fbits(float v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }

// An upper bound of the k-th smallest of the 16 values a DPP row holds (one per lane), in every lane of the row: k - 1
// times drop the row minimum (all lanes that tie with it: the bound can only get looser), then take the minimum.
__device__ __forceinline__ float row_kth_smallest(float v, const uint32_t k)
{
    float mk = row_allmin(v);
    for (uint32_t i = 1; i < k; ++i) {
        v = (v <= mk) ? __builtin_inff() : v;
        mk = row_allmin(v);
    }
    return mk;
}

// VAL: validation build (baz_music_debug_coarse_margin): every tile runs both forms, nothing is gated, no lists are kept,
// and the worst |c / SC - d| / (2^-16 (S + |d|)) over all (item, bin) is left in *margin (float bits, atomicMax).
//
// Two passes over the wave's bin range.  PASS 1 (coarse only): the smallest coarse value of every (lane, item) -- a lane
// sees the bins = c (mod 16), so the n-th smallest of a row's 16 lane minima bounds the item's n-th smallest coarse value
// c_(n) from above (n different bins at or below it) -- gives the threshold  D <= (c_(n) + E) (1 + 2^-15)  before a
// single exact tile has run.  Without it the thresholds only tighten as the walk happens to pass the minima: on a
// descending slope of the spectrum EVERY tile beats the list and fires (measured: 0.28 ms per 262,144 coherent cfg2 items,
// and 0.93 ms -- slower than the full scan -- when every item of a wave has its own scene).  PASS 2: the gated walk.
// PERM (lab builds only, sort_kernels.hip.h): rows through an index list and the fire statistic; the product's instantiation has neither.
template <int M, int NMAX, int RG, int TPP, bool VAL = false, int LAB = 0, bool PERM = false>
__global__ __launch_bounds__(256, (RG <= 2 && M <= 4) ? 3 : 2) void scan_coarse_kernel(const double* __restrict__ Qs,
                                                                             const uint4* __restrict__ imgC,
                                                                             const uint4* __restrict__ imgX,
                                                                             double* __restrict__ cand, uint32_t batch,
                                                                             uint32_t res, uint32_t qstride, uint32_t nphases,
                                                                             uint32_t nsplit, uint32_t keep_mask, uint32_t n,
                                                                             ScanRefine rf, CoarseParams cp,
                                                                             unsigned long long* __restrict__ margin,
                                                                             float* __restrict__ val_dump = nullptr,
                                                                             const uint32_t* __restrict__ perm = nullptr,
                                                                             unsigned long long* __restrict__ fstat = nullptr)
{
    // perm (round 5, sort_kernels.hip.h): row x of the launch is item perm[x] -- the items in an order in which the 16 rows of a group share
    // their nulls; nullptr: row x is item x.  fstat: [0] += exact (row group, tile) evaluations, [1] += (row group, tile) pairs walked.
    constexpr int MM = M * M;
    constexpr int NGC = cs_groups(M);                  // groups of 16 terms (4 fp64 k-steps each)
    constexpr bool WIDE = MM > 16;                     // 5 <= m <= 8: the coarse operands in groups of 32 terms (see above)
    constexpr int NG2 = WIDE ? cs_groups32(M) : 1;
    static_assert(MM <= 64, "m <= 8");
    constexpr int TC_UNITS = WIDE ? NG2 * CS_C32_UNITS : CS_C_UNITS, TX_UNITS = NGC * CS_X_UNITS;   // per tile
    // One group: the fp64 operands of the phase's tiles are staged through LDS beside the coarse ones.  More (m >= 5): an
    // exact tile reads its 2 NG KiB from the image in L2 when it fires (a few per cent of the tiles; LDS holds C only).
    constexpr bool XLDS = (NGC == 1);
    // 16-B units per staged phase, both multiples of 64: the C operands of TPP + 1 tiles -- the phase's own and the FIRST tile
    // of the next phase, whose MFMAs are issued while this phase's last tile is reduced (the software pipeline below runs
    // across the phase boundary) -- and the X operands of TPP tiles
    constexpr int C_UNITS = (TPP + 1) * TC_UNITS, X_UNITS = XLDS ? TPP * TX_UNITS : 0;
    constexpr int C_CHUNKS = C_UNITS / 64, X_CHUNKS = X_UNITS / 64;         // 1-KiB wave loads per phase
    __shared__ uint4 stage[2][C_UNITS + X_UNITS];                            // per buffer: [Fh | Fl of TPP + 1 tiles][X of TPP tiles]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;

    const uint32_t split = blockIdx.x % nsplit;
    const uint32_t item0 = ((blockIdx.x / nsplit) * 4 + wave) * (16 * RG);      // first ROW of the wave (= its item where perm is nullptr)
    auto item_of = [&](uint32_t row) -> uint32_t {
        row = (row < batch) ? row : (batch - 1);
        if constexpr (PERM) return perm ? perm[row] : row;
        else return row;
    };
    const uint32_t ph_begin = (uint32_t)(((uint64_t)nphases * split) / nsplit);
    const uint32_t ph_end = (uint32_t)(((uint64_t)nphases * (split + 1)) / nsplit);

    // ---- operands -------------------------------------------------------------------------------------------------
    double qa[RG][XLDS ? 4 : 1];   // exact A: q[item of row c][e = 4 s + g]   (natural row order; m >= 5: fetched when a tile fires)
    // coarse A, K = 32, row c = item pi(c), k = 8 g + j.  m <= 4: ONE operand [qh | ql], e = k & 15, piece = k >> 4 (ah[.][0]).
    // m >= 5: per group G of 32 terms the two pieces as separate operands, e = 32 G + k (ah, al).
    v8f16 ah[RG][NG2], al[RG][WIDE ? NG2 : 1];
    v4f32 es[RG], negthr[RG];  // per accumulator register r (item g + 4 r): error allowance and -threshold, coarse units
    double key[VAL ? 1 : RG][4][NMAX];
    bool row_ok[RG][4];
    [[maybe_unused]] uint32_t itn_q[RG];     // PERM: the item of natural row c of row group q (else it is recomputed where a rare path needs it)
    auto row_item = [&](const int q) -> uint32_t {
        if constexpr (PERM) return itn_q[q];
        else {
            const uint32_t it_n = item0 + 16 * (uint32_t)q + (uint32_t)c;
            return (it_n < batch) ? it_n : (batch - 1);
        }
    };
#pragma unroll
    for (int q = 0; q < RG; ++q) {
        const uint32_t itn = item_of(item0 + 16 * q + (uint32_t)c);                  // natural row c
        if constexpr (PERM) itn_q[q] = itn;
#pragma unroll
        for (int s = 0; s < (XLDS ? 4 : 1); ++s) {
            const int e = 4 * s + g;
            qa[q][s] = (XLDS && e < MM) ? Qs[(size_t)e * qstride + itn] : 0.0;
        }
        const uint32_t itp = item_of(item0 + 16 * q + (uint32_t)((c >> 2) + 4 * (c & 3)));  // permuted row c (the MFMA layouts' pi(c))
        // this lane's 8 coefficients per group of the permuted item: m <= 4: e = (8 g + j) & 15 (lanes g = 0, 1 together cover
        // all 16); m >= 5: e = 32 G + 8 g + j (the four g cover a group)
        float qsf[NG2][8], asum = 0.0f;
#pragma unroll
        for (int G = 0; G < NG2; ++G)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = WIDE ? 32 * G + 8 * g + j : ((8 * g + j) & 15);
                const double qv = (e < MM) ? Qs[(size_t)e * qstride + itp] : 0.0;
                qsf[G][j] = (float)(qv * 1024.0);
                asum += fabsf((float)qv);
            }
        asum += __shfl_xor(asum, 16, 64);                     // sum_e |q_e| of the permuted item (m <= 4: rows g = 0,1 / 2,3 agree)
        if constexpr (WIDE) asum += __shfl_xor(asum, 32, 64);
        // a projector's coefficients are <= 2 in magnitude; anything else (non-finite or garbage q) never gates
        const bool sane = asum <= 4.0f * (float)MM;           // false for NaN
#pragma unroll
        for (int G = 0; G < NG2; ++G)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = sane ? qsf[G][j] : 0.0f;
                const _Float16 h = (_Float16)v;
                const _Float16 l = (_Float16)(v - (float)h);  // exact difference (11-bit piece of a 24-bit value)
                if constexpr (WIDE) { ah[q][G][j] = h; al[q][G][j] = l; }
                else ah[q][0][j] = (g < 2) ? h : l;           // k < 16: hi, k >= 16: lo
            }
        // allowance of item g + 4 r = permuted row 4 g + r: held by the lanes with c = 4 g + r
        const float es_row = sane ? asum * cp.es_factor * 1.0001f : __builtin_inff();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            es[q][r] = __shfl(es_row, 4 * g + r, 64);
            negthr[q][r] = -__builtin_inff();                 // (set by pass 1)
            row_ok[q][r] = (item0 + 16 * q + (uint32_t)(g + 4 * r)) < batch;
        }
    }
    if constexpr (!VAL) {
#pragma unroll
        for (int q = 0; q < RG; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < NMAX; ++i) key[q][r][i] = key_empty();
    }
    const bool refine_on = rf.Gs != nullptr;
    const double below_d = refine_on ? rf.below : -1.0;
    const float below_s = refine_on ? (float)(rf.below * cp.sc) * 1.000001f : 0.0f;      // in coarse units, rounded up
    const uint32_t nobin = ~keep_mask;
    [[maybe_unused]] float worst = 0.0f;
    [[maybe_unused]] uint32_t worst_at = 0;       // VAL: bin | (item & 0xFFF) << 20 of the worst value (diagnostics)
    uint32_t refined = 0;
    [[maybe_unused]] uint32_t fired = 0;          // exact (row group, tile) evaluations of this wave (lab statistic)

    // ---- table staging: L2 -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave instruction, no registers) ----------
    auto stage_load = [&](uint32_t ph, int b, const bool with_x, const int tile_stride = 1) {
        const uint4* __restrict__ sc = imgC + (size_t)ph * (TPP * TC_UNITS) + lane;        // (the C array carries one tile of padding)
        constexpr int CPT = TC_UNITS / 64;                     // 1-KiB chunks per tile
#pragma unroll
        for (int i = 0; i < (C_CHUNKS + 3) / 4; ++i) {
            const int j = i * 4 + wave;                        // wave-uniform: chunk j of the phase
            if (j < C_CHUNKS && ((j / CPT) % tile_stride) == 0)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sc + j * 64),
                                                 (__attribute__((address_space(3))) void*)(&stage[b][j * 64]), 16, 0, 0);
        }
        if (XLDS && with_x) {
            const uint4* __restrict__ sx = imgX + (size_t)ph * X_UNITS + lane;
#pragma unroll
            for (int i = 0; i < (X_CHUNKS + 3) / 4; ++i) {
                const int j = i * 4 + wave;
                if (j < X_CHUNKS)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sx + j * 64),
                                                     (__attribute__((address_space(3))) void*)(&stage[b][C_UNITS + j * 64]), 16, 0, 0);
            }
        }
    };

    int buf = 0;
    // Software pipeline of both passes: the MFMAs of tile t + 1 are issued BEFORE the integer reduction of tile t's results,
    // and the scheduler is told to interleave them (one MFMA, two VALU): a wave issues in order, so VALU work placed between
    // two MFMAs runs while the matrix pipe is busy, and VALU work placed behind them waits for them.  Two register sets (A,
    // B) alternate, so nothing is copied.  (Measured before: MFMA pipe 49 % busy, 30 % of the wave cycles in issue stalls;
    // profiles/r03_coarse_scan_pmc_first.txt.)
    struct BOp { v8f16 h[1], l[1]; };
    // One group: the B operands of the NEXT tile are fetched into a second register set a tile ahead (ld_b).  More groups:
    // they are fetched inside issue(), group by group (8 .. 16 KiB of registers would not fit beside two row groups, and
    // with one row group the LDS reads -- 2 NG KiB per wave and tile for 2 NG MFMAs -- bound the kernel, not the matrix core).
    constexpr bool BAHEAD = !WIDE;
    auto ld_b = [&](const char* __restrict__ T0, int t, BOp& b) __attribute__((always_inline)) {
        if constexpr (BAHEAD) {
            const char* __restrict__ T = T0 + t * (TC_UNITS * 16);
            b.h[0] = *reinterpret_cast<const v8f16*>(T);
            b.l[0] = *reinterpret_cast<const v8f16*>(T + 512);
        }
    };
    auto issue = [&](v4f32 (&u)[RG], const BOp& b, const char* __restrict__ T0, int t, const bool with_thr) __attribute__((always_inline)) {
        if constexpr (!WIDE) {
#pragma unroll
            for (int q = 0; q < RG; ++q)
                u[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[q][0], b.h[0], with_thr ? negthr[q] : (v4f32){0, 0, 0, 0}, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < RG; ++q) u[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[q][0], b.l[0], u[q], 0, 0, 0);
        } else {
            // per group: qh Fh, ql Fh on one B operand, then qh Fl (ql Fl is below the budget: <= 2^-22 S)
            const char* __restrict__ T = T0 + t * (TC_UNITS * 16);
#pragma unroll
            for (int G = 0; G < NG2; ++G) {
                const v8f16 bh = *reinterpret_cast<const v8f16*>(T + G * 2048);
#pragma unroll
                for (int q = 0; q < RG; ++q)
                    u[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[q][G], bh, (G > 0) ? u[q] : (with_thr ? negthr[q] : (v4f32){0, 0, 0, 0}), 0, 0, 0);
#pragma unroll
                for (int q = 0; q < RG; ++q) u[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[q][G], bh, u[q], 0, 0, 0);
            }
#pragma unroll
            for (int G = 0; G < NG2; ++G) {
                const v8f16 bl = *reinterpret_cast<const v8f16*>(T + G * 2048 + 1024);
#pragma unroll
                for (int q = 0; q < RG; ++q) u[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[q][G], bl, u[q], 0, 0, 0);
            }
        }
    };
    auto interleave = [&]() __attribute__((always_inline)) {      // the tile's MFMAs, two VALU instructions behind each
#pragma unroll
        for (int i = 0; i < (WIDE ? 3 * NG2 : 2) * RG; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
    };

    // ---- pass 1: thresholds from the coarse form alone -----------------------------------------------------------------
    if constexpr (!VAL) {
        int pm[RG][4];              // running minimum of the coarse values, as bit patterns (see fbits)
#pragma unroll
        for (int q = 0; q < RG; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) pm[q][r] = 0x7F800000;    // +inf
        auto reduce1 = [&](const v4f32 (&u)[RG]) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < RG; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pm[q][r] = imin(pm[q][r], fbits(u[q][r]));
                    // opaque from here: otherwise the compiler folds this tile's and the next tile's reduction into one
                    // v_min3 that waits (s_nop 7) for the NEXT tile's MFMAs -- the opposite of the pipeline
                    asm volatile("" : "+v"(pm[q][r]));
                }
        };
        // EVERY OTHER bin tile only (round 4): the n-th smallest of the minima over a SUBSET of a row's bins is still an upper
        // bound of its n-th smallest d, and 16 bins further on the spectrum has barely moved -- half of this pass's MFMAs,
        // reductions and staged operands, the same share of exact tiles in pass 2 (config 2, 262,144 items: scan 0.220 -> 0.197 ms
        // on coherent streams, 0.488 -> 0.475 on an incoherent batch; every fourth tile: 0.186 / 0.494 with 26 instead of 23 % of
        // the tiles exact there; profiles/r04_coarse_pass1_stride.txt).
        constexpr int PS = (LAB == 3) ? 1 : ((LAB == 4 && TPP >= 8) ? 4 : 2);     // (lab 3 / 4: every tile / every fourth tile)
        static_assert(TPP % (2 * PS) == 0, "pass 1 walks its tiles of a phase in pairs");
        if (ph_begin < ph_end) stage_load(ph_begin, 0, false, PS);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        BOp bA, bB;
        v4f32 uA[RG], uB[RG];
        for (uint32_t ph = ph_begin; ph < ph_end; ++ph) {
            if (ph + 1 < ph_end) stage_load(ph + 1, buf ^ 1, false, PS);
            const char* __restrict__ T0 = reinterpret_cast<const char*>(&stage[buf][0]) + (((WIDE ? g : (g & 1)) * 16 + c) * 16);
            if (ph == ph_begin) {                                              // prologue of the pass: tile 0
                ld_b(T0, 0, bA);
                issue(uA, bA, T0, 0, false);
            }
            ld_b(T0, PS, bB);
#pragma nounroll
            for (int tl = 0; tl < TPP; tl += 2 * PS) {
                issue(uB, bB, T0, tl + PS, false);                             // tile tl + PS ...
                ld_b(T0, tl + 2 * PS, bA);
                reduce1(uA);                                                   // ... while tile tl is reduced
                interleave();
                issue(uA, bA, T0, tl + 2 * PS, false);                         // tile tl + 2 PS (= tile 0 of the next phase at the end) ...
                ld_b(T0, (tl + 3 * PS <= TPP) ? tl + 3 * PS : TPP, bB);
                reduce1(uB);                                                   // ... while tile tl + PS is reduced
                interleave();
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            buf ^= 1;
        }
        // With E = NG2 2^-16 (S + D):  D <= (c_(n) + es) / (1 - NG2 2^-16), never below `refine_below`;
        // thr = D (1 + NG2 2^-16) + es   (all in coarse units).  Both factors follow from NG2 (round 3 had the constants of one
        // group everywhere: for 6 .. 8 antennas, NG2 = 2, the first threshold was 1.4e-5 D tighter than the bound it claims)
        // and are rounded up; one group (m <= 5) keeps round 3's values.
        constexpr float DDF = NG2 <= 1 ? 1.0000306f : (float)((1.0 + 0x1p-20) / (1.0 - NG2 * 0x1p-16));
        constexpr float THF = NG2 <= 1 ? 1.0000164f : (float)((1.0 + NG2 * 0x1p-16) * (1.0 + 0x1p-19));
        static_assert((double)DDF * (1.0 - NG2 * 0x1p-16) >= 1.0 && (double)THF >= 1.0 + NG2 * 0x1p-16, "threshold factors must be upper bounds");
#pragma unroll
        for (int q = 0; q < RG; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float cn = row_kth_smallest(__builtin_bit_cast(float, pm[q][r]), n);
                const float Dd = fmaxf(fmaxf(cn + es[q][r], 0.0f) * DDF, below_s);
                negthr[q][r] = -__builtin_fmaf(Dd, THF, es[q][r]);
            }
    }

    // ---- pass 2: the gated walk ------------------------------------------------------------------------------------------
    // one tile whose vote fired: the exact form for the row groups that asked for it (u = that tile's coarse results)
    auto exact_tile = [&](const v4f32 (&u)[RG], const uint32_t tile, const char* __restrict__ Xt) __attribute__((always_inline)) {
        const uint32_t bin = tile * 16u + (uint32_t)c;
        const double* __restrict__ X = reinterpret_cast<const double*>(Xt);
#pragma unroll
        for (int q = 0; q < RG; ++q) {
            const int mnq = imin(imin(fbits(u[q][0]), fbits(u[q][1])), imin(fbits(u[q][2]), fbits(u[q][3])));
            if (!VAL && !__any(mnq <= 0)) continue;
            if constexpr (!VAL) ++fired;
            // exact form: scan_mfma_kernel's projector GEMM for this 16 x 16 tile (same k order, same operands)
            v4f64 acc = {0, 0, 0, 0};
            if constexpr (XLDS) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const v2f64 x = *reinterpret_cast<const v2f64*>(X + s2 * 128 + lane * 2);       // k-steps 2 s2, 2 s2 + 1
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[q][2 * s2], x.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(qa[q][2 * s2 + 1], x.y, acc, 0, 0, 0);
                }
            } else {
                // both operands from L2: the item's coefficients (natural row c) and the tile's fp64 image
                const double* __restrict__ qp = Qs + row_item(q) + (size_t)g * qstride;
#pragma unroll
                for (int s2 = 0; s2 < 2 * NGC; ++s2) {
                    const v2f64 x = *reinterpret_cast<const v2f64*>(X + s2 * 128 + lane * 2);
                    if (4 * (2 * s2) < MM) {
                        const double a = (4 * (2 * s2) + g < MM) ? qp[(size_t)(4 * (2 * s2)) * qstride] : 0.0;
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, x.x, acc, 0, 0, 0);
                    }
                    if (4 * (2 * s2 + 1) < MM) {
                        const double a = (4 * (2 * s2 + 1) + g < MM) ? qp[(size_t)(4 * (2 * s2 + 1)) * qstride] : 0.0;
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, x.y, acc, 0, 0, 0);
                    }
                }
            }
            if constexpr (VAL) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // S of item g + 4 r in d units: es / (2^-16 SC 1.0001); allowance 2^-16 (S + |d|)
                    const double S = (double)es[q][r] / (1.0001 * (double)cp.es_factor) * cp.fmax;
                    const double err = fabs((double)u[q][r] / cp.sc - acc[r]);
                    const double allow = (double)NG2 * 0x1p-16 * (S + fabs(acc[r]));
                    const bool counts = bin < res && row_ok[q][r] && S < 1e30 && allow > 0.0 && err == err;
                    const float ratio = counts ? (float)(err / allow) : 0.0f;
                    if (val_dump && bin < res && row_ok[q][r])       // lab: every ratio, [item][bin]
                        val_dump[(size_t)(item0 + 16 * q + (uint32_t)(g + 4 * r)) * res + bin] = ratio;
                    if (ratio > worst) {
                        worst = ratio;
                        worst_at = bin | (((item0 + 16 * q + (uint32_t)(g + 4 * r)) & 0xFFFu) << 20);
                    }
                }
            } else {
                bool low = false;
#pragma unroll
                for (int r = 0; r < 4; ++r) low |= (fabs(acc[r]) <= below_d);
                if (refine_on && __any(low)) {                  // near-null tile: the reference's literal form, per value
                    const v4f64 d = literal16<M>(rf.Gs, rf.TB, row_item(q), g, qstride, (int)M - (int)n, bin);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool redo = (fabs(acc[r]) <= rf.below) && (bin < res);
                        acc[r] = redo ? d[r] : acc[r];
                        refined += (redo && row_ok[q][r]) ? 1u : 0u;
                    }
                }
                const uint32_t kbin = (bin < res) ? bin : nobin;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // Only accumulator registers in which some lane's value passed the coarse test can change a list: a
                    // value with c > thr cannot be among the final n (the argument of the gate itself), so the other
                    // registers' values need not be offered at all.  In a batch of unrelated items the tile was usually
                    // drawn by ONE of the 16 items: three of the four insert / threshold blocks are skipped.
                    if (!__any(fbits(u[q][r]) <= 0)) continue;
                    double ko = key[q][r][0];                   // the list's n-th entry (the lists hold NMAX >= n) ...
#pragma unroll
                    for (int i = 1; i < NMAX; ++i) ko = ((uint32_t)i < n) ? key[q][r][i] : ko;
                    key_insert_new<NMAX>(key[q][r], make_key(acc[r], kbin, keep_mask));
                    double kn = key[q][r][0];                   // ... before and after this tile's value
#pragma unroll
                    for (int i = 1; i < NMAX; ++i) kn = ((uint32_t)i < n) ? key[q][r][i] : kn;
                    // the threshold moves only when some lane's n-th entry did: usually one item of the 16 drew the tile
                    if (!cp.lazy || __any(__builtin_bit_cast(uint64_t, kn) != __builtin_bit_cast(uint64_t, ko))) {
                        const uint64_t kb = __builtin_bit_cast(uint64_t, kn) | (uint64_t)(~keep_mask);
                        const double D = fmax(__builtin_bit_cast(double, kb), below_d);
                        // (float) rounds to nearest: sc_up carries the factor that makes the product an upper bound
                        const float thr = row_allmin(__builtin_fmaf((float)D, cp.sc_up, es[q][r]));
                        negthr[q][r] = fmaxf(negthr[q][r], -thr);   // thresholds only ever tighten
                    }
                }
            }
        }
    };
    // the vote of one tile: some c - thr <= 0 among the wave's 16 RG values per lane?
    auto vote = [&](const v4f32 (&u)[RG]) __attribute__((always_inline)) -> int {
        int m = 0x7F800000;
#pragma unroll
        for (int q = 0; q < RG; ++q) m = imin(m, imin(imin(fbits(u[q][0]), fbits(u[q][1])), imin(fbits(u[q][2]), fbits(u[q][3]))));
        return m;
    };
    if (ph_begin < ph_end) stage_load(ph_begin, buf, LAB != 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    BOp bA, bB;
    v4f32 uA[RG], uB[RG];
    for (uint32_t ph = ph_begin; ph < ph_end; ++ph) {
        if (ph + 1 < ph_end) stage_load(ph + 1, buf ^ 1, LAB != 2);   // lands while this phase's tiles run (lab 2: no X operands)
        const char* __restrict__ T0 = reinterpret_cast<const char*>(&stage[buf][0]) + (((WIDE ? g : (g & 1)) * 16 + c) * 16);
        // the fp64 operands of this phase's tiles: staged (one group) or where the image has them (L2)
        const char* __restrict__ X0 = XLDS ? reinterpret_cast<const char*>(&stage[buf][C_UNITS])
                                           : reinterpret_cast<const char*>(imgX + (size_t)ph * (TPP * TX_UNITS));
        if (ph == ph_begin) {                                                  // prologue of the pass: tile 0
            ld_b(T0, 0, bA);
            issue(uA, bA, T0, 0, !VAL);
        }
        ld_b(T0, 1, bB);
#pragma nounroll
        for (int tl = 0; tl < TPP; tl += 2) {
            issue(uB, bB, T0, tl + 1, !VAL);                                   // tile tl + 1 (with the thresholds as they are now)
            ld_b(T0, tl + 2, bA);
            const int mA = vote(uA);
            interleave();
            if constexpr (LAB >= 1) fired += (mA <= 0) ? 1u : 0u;              // lab: the cost of the coarse passes alone (results are wrong)
            else if (VAL || __any(mA <= 0)) exact_tile(uA, ph * TPP + (uint32_t)tl, X0 + tl * (TX_UNITS * 16));
            issue(uA, bA, T0, tl + 2, !VAL);                                   // tile tl + 2 (= tile 0 of the next phase at the end)
            ld_b(T0, (tl + 3 <= TPP) ? tl + 3 : TPP, bB);
            const int mB = vote(uB);
            interleave();
            if constexpr (LAB >= 1) fired += (mB <= 0) ? 1u : 0u;
            else if (VAL || __any(mB <= 0)) exact_tile(uB, ph * TPP + (uint32_t)tl + 1u, X0 + (tl + 1) * (TX_UNITS * 16));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        buf ^= 1;
    }

    if constexpr (VAL) {
        unsigned long long packed = ((unsigned long long)__builtin_bit_cast(unsigned int, worst) << 32) | worst_at;
#pragma unroll
        for (int msk = 1; msk < 64; msk <<= 1) {
            const unsigned long long o = __shfl_xor(packed, msk, 64);
            packed = o > packed ? o : packed;
        }
        if (lane == 0 && margin) atomicMax(margin, packed);      // ratio >= 0: its bit pattern orders like the value
        return;
    } else {
        if (rf.count) {
#pragma unroll
            for (int msk = 1; msk < 64; msk <<= 1) refined += __shfl_xor(refined, msk, 64);
            if (lane == 0 && refined) atomicAdd(rf.count, (unsigned long long)refined);
        }
        if (margin && lane == 0) atomicAdd(margin, (unsigned long long)fired);      // lab (BAZ_MUSIC_COARSE_STATS)
        if (PERM && fstat && lane == 0) {                                            // what the context's sorting policy reads
            atomicAdd(fstat, (unsigned long long)fired);
            atomicAdd(fstat + 1, (unsigned long long)(ph_end - ph_begin) * (unsigned long long)(TPP * RG));
        }
        // merge the 16 lanes of an item row, emit this range's candidates (topn_merge_kernel folds the ranges)
#pragma unroll
        for (int q = 0; q < RG; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                key_merge_xor<NMAX>(key[q][r], 1);
                key_merge_xor<NMAX>(key[q][r], 2);
                key_merge_xor<NMAX>(key[q][r], 4);
                key_merge_xor<NMAX>(key[q][r], 8);
                const uint32_t row = item0 + 16 * q + (uint32_t)(g + 4 * r);
                uint32_t it = row;
                if constexpr (PERM) it = (uint32_t)__shfl((int)itn_q[q], g + 4 * r, 64);    // the item of that row: lane (0, g + 4 r) holds it
                if (c == 0 && row < batch) {
#pragma unroll
                    for (int i = 0; i < NMAX; ++i) cand[((size_t)it * nsplit + split) * NMAX + i] = key[q][r][i];
                }
            }
    }
}

}  // namespace bazmusic
