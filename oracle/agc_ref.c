/* CPU ORACLE (test infrastructure, NOT the product) -- plain-C restatement of gr-baz's AGC block
 *     /root/reference/lib/baz_agc_cc.cc:64-102   (the live part of work(); everything after the
 *     unconditional `continue` at :102 is dead code)
 * state: _env (double), _gain (double), _count  (lib/baz_agc_cc.h:52-57), constructor defaults
 * rate = 1e-4, reference = 1.0, gain = 1.0, max_gain = 0.0 (lib/baz_agc_cc.h:41; max_gain is only
 * used by the dead code).  Only tests/ may load this library.
 */
#include <math.h>
#include <stddef.h>

typedef struct {
    float rate;          /* _rate (float)            .h:52 */
    double reference;    /* _reference (double)      .h:53 */
    double gain;         /* _gain                    .h:54 */
    unsigned long long count;   /* _count            .h:56 */
    double env;          /* _env                     .h:57 */
} agc_ref_state;

void agc_ref_init(agc_ref_state* s, float rate, float reference, float gain)
{
    s->rate = rate;
    s->reference = reference;
    s->gain = gain;
    s->count = 0;      /* .cc:58 */
    s->env = 0.0;      /* .cc:59 */
}

/* in/out: n complex64 as interleaved floats; env, mul: n floats or NULL (ports 1, 2; .cc:68-69). */
int agc_ref_work(agc_ref_state* s, const float* in_ri, size_t n, float* out_ri, float* env, float* mul)
{
    size_t i;
    for (i = 0; i < n; i++, s->count++) {                       /* .cc:72 */
        double d0 = in_ri[2 * i], d1 = in_ri[2 * i + 1];        /* .cc:74-75 */
        double mag2 = d0 * d0 + d1 * d1;                        /* .cc:76 */
        double mag = sqrt(mag2);                                /* .cc:77 */
        if (s->count == 0)                                      /* .cc:79-82 */
            s->env = mag;
        else
            s->env = (s->env * (1.0 - s->rate)) + (mag * s->rate);
        if (env) env[i] = (float)s->env;                        /* .cc:84-85 */
        s->gain = s->reference / s->env;                        /* .cc:89 */
        if (mul) mul[i] = (float)s->gain;                       /* .cc:91-92 */
        d0 *= s->gain;                                          /* .cc:97-98 */
        d1 *= s->gain;
        out_ri[2 * i] = (float)d0;                              /* .cc:100 */
        out_ri[2 * i + 1] = (float)d1;
    }
    return (int)n;                                              /* .cc:149 */
}
