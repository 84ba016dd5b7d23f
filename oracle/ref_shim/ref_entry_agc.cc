// ORACLE-ONLY (test infrastructure): C entry points around the reference's OWN baz_agc_cc block
// (compiled from /root/reference/lib/baz_agc_cc.cc, see oracle/Makefile target "ref").
#include <baz_agc_cc.h>
#include <cstddef>

extern "C" {

void* baz_ref_agc_create(float rate, float reference, float gain, float max_gain)
{
    return new baz_agc_cc_sptr(baz_make_agc_cc(rate, reference, gain, max_gain));
}

void baz_ref_agc_destroy(void* h) { delete static_cast<baz_agc_cc_sptr*>(h); }

// n_outputs: 1 (out), 2 (+env) or 3 (+env, mul), like output_items.size() in work()
int baz_ref_agc_work(void* h, const float* in_ri, int n, float* out_ri, float* env, float* mul)
{
    baz_agc_cc_sptr& blk = *static_cast<baz_agc_cc_sptr*>(h);
    gr_vector_const_void_star in(1, in_ri);
    gr_vector_void_star out;
    out.push_back(out_ri);
    if (env) out.push_back(env);
    if (env && mul) out.push_back(mul);
    return blk->work(n, in, out);
}

}  // extern "C"
