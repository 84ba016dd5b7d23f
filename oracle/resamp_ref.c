/* ORACLE -- TEST INFRASTRUCTURE ONLY (never linked into or imported by the product path).
 *
 * Plain-C restatement of gr-baz's fractional resampler
 *     /root/reference/lib/baz_fractional_resampler_cc.cc:80-101  (constructor, ratio rules)
 *     /root/reference/lib/baz_fractional_resampler_cc.cc:152-217 (general_work, both branches)
 *     /root/reference/lib/baz_fractional_resampler_cc.cc:220-254 (mu / ratio setters, deferred)
 * with x87 `long double` for the phase accumulator exactly as the reference declares it (.cc:41-49).
 *
 * The arithmetic of one output sample lives in a THIRD-PARTY dependency that is NOT under
 * /root/reference: GNU Radio 3.7's gr::filter::mmse_fir_interpolator_cc (gnuradio-filter, unpinned:
 * found through find_package(Gnuradio) in the reference's CMake tree; call sites .cc:28,41,87,172,206).
 * Its published algorithm, restated here:
 *   - NSTEPS = 128, NTAPS = 8; table taps[0..128][0..7];
 *   - interpolate(input, mu): imu = (int) rint(mu * NSTEPS) with mu a float; the result is the FIR
 *     filters[imu]->filter(input), whose kernel stores the taps reversed: sum_k input[k] * taps[imu][7-k],
 *     accumulated in float;
 *   - the table was produced by GNU Radio's gen_interpolator_taps, which minimises
 *         integral_{-B}^{B} | sum_j h_j e^{-i 2 pi f j} - e^{-i 2 pi f (4 - mu)} |^2 df,   B = 0.25,
 *     numerically (praxis) and prints 6 significant digits.  That problem is linear least squares with
 *     the closed form  A h = b,  A_jl = 2B sinc(2B (j-l)),  b_j = 2B sinc(2B (j - (4 - mu))), used here.
 *     The header holds the generator's printout ("%12.5e") as float literals, so the table here is the closed form
 *     ROUNDED TO SIX SIGNIFICANT DECIMAL DIGITS and then to float.  PARITY UNPINNED: the reference holds no golden
 *     vectors for this path and the header is not available offline; two recollections of its rows 1 and 3 exist and
 *     disagree in the outermost tap by 9.5e-7 / 2.8e-6 (tests/test_resamp.py keeps both as candidates and asserts
 *     neither) -- so the default table may differ from the real one by a few 1e-6 per tap.  A host that has
 *     gnuradio-filter installs ITS table through baz_resamp_set_taps() (include/baz_resamp_hip.h) and is then exact;
 *     scripts/dump_gr_mmse_taps.py turns such a host into the fixture tests/golden/mmse_taps_gr37.npz, against which
 *     the default is then compared bit for bit (test_default_table_equals_gnuradio_filters_bit_for_bit).
 */
#define _POSIX_C_SOURCE 200809L
#include <locale.h>
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define RS_NSTEPS 128
#define RS_NTAPS 8

typedef struct {
    long double mu;              /* d_mu            .cc:41 */
    long double mu_inc;          /* d_mu_inc        .cc:42 */
    int update;                  /* d_update        .cc:44 */
    long double mu_inc_update;   /* d_mu_inc_update .cc:45 */
    int update_mu;               /* d_update_mu     .cc:46 */
    long double mu_update;       /* d_mu_update     .cc:47 */
    int update_mu_adj;           /* d_update_mu_adj .cc:48 */
    long double mu_adj;          /* d_mu_adj        .cc:49 */
    float taps[RS_NSTEPS + 1][RS_NTAPS];
} resamp_ref_t;

static long double sincl_(long double x)
{
    const long double pi = 3.14159265358979323846264338327950288L;
    if (x == 0.0L) return 1.0L;
    return sinl(pi * x) / (pi * x);
}

/* closed-form MMSE table (see header) */
void resamp_ref_taps(float taps[RS_NSTEPS + 1][RS_NTAPS])
{
    const long double B = 0.25L;
    locale_t cloc = newlocale(LC_ALL_MASK, "C", (locale_t)0);      /* "." is the decimal point of the round trip below */
    locale_t prev = cloc ? uselocale(cloc) : (locale_t)0;
    for (int i = 0; i <= RS_NSTEPS; ++i) {
        long double A[RS_NTAPS][RS_NTAPS + 1];
        const long double delay = 4.0L - (long double)i / RS_NSTEPS;
        for (int j = 0; j < RS_NTAPS; ++j) {
            for (int l = 0; l < RS_NTAPS; ++l) A[j][l] = 2 * B * sincl_(2 * B * (long double)(j - l));
            A[j][RS_NTAPS] = 2 * B * sincl_(2 * B * ((long double)j - delay));
        }
        for (int c = 0; c < RS_NTAPS; ++c) {              /* Gauss-Jordan, partial pivoting */
            int p = c;
            for (int r = c + 1; r < RS_NTAPS; ++r) if (fabsl(A[r][c]) > fabsl(A[p][c])) p = r;
            if (p != c) for (int k = 0; k <= RS_NTAPS; ++k) { long double t = A[c][k]; A[c][k] = A[p][k]; A[p][k] = t; }
            for (int r = 0; r < RS_NTAPS; ++r) {
                if (r == c) continue;
                const long double f = A[r][c] / A[c][c];
                for (int k = c; k <= RS_NTAPS; ++k) A[r][k] -= f * A[c][k];
            }
        }
        /* the published table holds the generator's printout ("%12.5e", six significant digits) compiled as float */
        for (int j = 0; j < RS_NTAPS; ++j) {
            char dec[40];
            snprintf(dec, sizeof dec, "%.5Le", A[j][RS_NTAPS] / A[j][j]);
            taps[i][j] = (float)strtod(dec, NULL);
        }
    }
    if (cloc) { uselocale(prev); freelocale(cloc); }
    /* the end rows are pure delays in the published table */
    memset(taps[0], 0, sizeof(taps[0]));
    memset(taps[RS_NSTEPS], 0, sizeof(taps[RS_NSTEPS]));
    taps[0][4] = 1.0f;
    taps[RS_NSTEPS][3] = 1.0f;
}

/* constructor rules, .cc:80-101; returns 0, or -1 for the two std::out_of_range cases (.cc:94-97) */
int resamp_ref_init(resamp_ref_t* s, double phase_shift, double resamp_ratio, unsigned long long num,
                    unsigned long long denom)
{
    long double ratio = (long double)resamp_ratio;
    memset(s, 0, sizeof(*s));
    s->mu = (long double)phase_shift;
    s->mu_inc = ratio;
    if (denom != 0) s->mu_inc = ratio = (long double)num / (long double)denom;   /* .cc:89-92 */
    if (ratio <= 0) return -1;
    if (phase_shift < 0 || phase_shift > 1) return -1;
    resamp_ref_taps(s->taps);
    return 0;
}

void resamp_ref_set_mu(resamp_ref_t* s, double mu) { s->update_mu = 1; s->mu_update = (long double)mu; }            /* .cc:233-238 */
void resamp_ref_set_ratio(resamp_ref_t* s, double r) { s->mu_inc_update = (long double)r; s->update = 1; }           /* .cc:240-245 */
void resamp_ref_set_ratio_rational(resamp_ref_t* s, unsigned long long num, unsigned long long denom)                 /* .cc:247-254 */
{
    if (denom != 0) { s->mu_inc_update = (long double)num / (long double)denom; s->update = 1; }
}
void resamp_ref_adjust(resamp_ref_t* s, double d) { s->mu_adj = (long double)d * s->mu_inc; s->update_mu_adj = 1; }   /* .cc:130-134 */
double resamp_ref_mu(const resamp_ref_t* s) { return (double)s->mu; }
double resamp_ref_ratio(const resamp_ref_t* s) { return (double)s->mu_inc; }

/* mmse_fir_interpolator_cc::interpolate (gnuradio-filter, see header) */
static void interpolate_(const resamp_ref_t* s, const float* in_ri, float mu, float* out_ri)
{
    const int imu = (int)rintf(mu * RS_NSTEPS);
    const float* t = s->taps[imu];
    float re = 0.0f, im = 0.0f;
    for (int k = 0; k < RS_NTAPS; ++k) {
        re += in_ri[2 * k] * t[RS_NTAPS - 1 - k];
        im += in_ri[2 * k + 1] * t[RS_NTAPS - 1 - k];
    }
    out_ri[0] = re;
    out_ri[1] = im;
}

/* forecast, .cc:141-149 */
int resamp_ref_forecast(const resamp_ref_t* s, int noutput_items)
{
    return (int)ceill((noutput_items * s->mu_inc) + RS_NTAPS);
}

/* general_work, one-input branch (.cc:162-203).  Returns noutput_items; *consumed = ii (consume_each). */
int resamp_ref_work(resamp_ref_t* s, const float* in_ri, int noutput_items, float* out_ri, int* consumed)
{
    int ii = 0, oo = 0;
    while (oo < noutput_items) {
        if (s->update_mu) { s->mu = s->mu_update; s->update_mu = 0; }             /* .cc:165-170 */
        interpolate_(s, in_ri + 2 * (size_t)ii, (float)s->mu, out_ri + 2 * (size_t)oo);   /* .cc:172 */
        ++oo;
        if (s->update) { s->mu_inc = s->mu_inc_update; s->update = 0; }            /* .cc:175-181 */
        long double sum = s->mu + s->mu_inc;                                       /* .cc:183 */
        if (s->update_mu_adj) { sum += s->mu_adj; s->update_mu_adj = 0; }          /* .cc:184-189 */
        long double f = floorl(sum);                                               /* .cc:190 */
        int incr = (int)f;
        s->mu = sum - f;
        ii += incr;
    }
    *consumed = ii;
    return noutput_items;
}

/* general_work, two-input branch (.cc:205-217): per-sample ratio input rr */
int resamp_ref_work2(resamp_ref_t* s, const float* in_ri, const float* rr, int noutput_items, float* out_ri,
                     int* consumed)
{
    int ii = 0, oo = 0;
    while (oo < noutput_items) {
        interpolate_(s, in_ri + 2 * (size_t)ii, (float)s->mu, out_ri + 2 * (size_t)oo);
        ++oo;
        s->mu_inc = rr[ii];
        long double sum = s->mu + s->mu_inc;
        long double f = floorl(sum);
        int incr = (int)f;
        s->mu = sum - f;
        ii += incr;
    }
    *consumed = ii;
    return noutput_items;
}

size_t resamp_ref_sizeof(void) { return sizeof(resamp_ref_t); }
const float* resamp_ref_table(const resamp_ref_t* s) { return &s->taps[0][0]; }
