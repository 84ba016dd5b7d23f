// ORACLE-ONLY (test infrastructure): C entry points around the reference's OWN fractional resampler block
// (compiled from /root/reference/lib/baz_fractional_resampler_cc.cc, see oracle/Makefile target "ref"; the MMSE
// interpolator it calls is the shim in gnuradio/filter/, because gnuradio-filter is not vendored).
#include <baz_fractional_resampler_cc.h>
#include <stdexcept>
#include <cstddef>

typedef gr::baz::fractional_resampler_cc::sptr rs_sptr;

extern "C" {

void* baz_ref_resamp_create(double phase_shift, double ratio, unsigned long long num, unsigned long long denom)
{
    try {
        return new rs_sptr(gr::baz::fractional_resampler_cc::make(phase_shift, ratio, num, denom));
    } catch (const std::exception&) {
        return nullptr;     // std::out_of_range from the constructor (.cc:94-97)
    }
}
void baz_ref_resamp_destroy(void* h) { delete static_cast<rs_sptr*>(h); }

int baz_ref_resamp_forecast(void* h, int noutput, int ninputs)
{
    gr_vector_int req(ninputs, 0);
    (*static_cast<rs_sptr*>(h))->forecast(noutput, req);
    return req[0];
}

// rr == NULL: one-input branch; else the two-input (per-sample ratio) branch.  Returns produced, *consumed = consume_each().
int baz_ref_resamp_work(void* h, const float* in_ri, const float* rr, int noutput, float* out_ri, int* consumed)
{
    rs_sptr& b = *static_cast<rs_sptr*>(h);
    gr_vector_int nin(rr ? 2 : 1, 1 << 30);
    gr_vector_const_void_star in;
    in.push_back(in_ri);
    if (rr) in.push_back(rr);
    gr_vector_void_star out(1, out_ri);
    int r = b->general_work(noutput, nin, in, out);
    *consumed = b->shim_consumed();
    return r;
}
void baz_ref_resamp_set_mu(void* h, double mu) { (*static_cast<rs_sptr*>(h))->set_mu((long double)mu); }
void baz_ref_resamp_set_ratio(void* h, double r) { (*static_cast<rs_sptr*>(h))->set_resamp_ratio(r); }
void baz_ref_resamp_set_ratio_rational(void* h, unsigned long long n, unsigned long long d) { (*static_cast<rs_sptr*>(h))->set_resamp_ratio(n, d); }
void baz_ref_resamp_post_double(void* h, double d) { (*static_cast<rs_sptr*>(h))->shim_post(pmt::from_double(d)); }      // .cc:127-134
void baz_ref_resamp_post_ppb(void* h, long i, double frac) { (*static_cast<rs_sptr*>(h))->shim_post(pmt::cons(pmt::from_long(i), pmt::from_double(frac))); }  // .cc:116-124
double baz_ref_resamp_mu(void* h) { return (double)(*static_cast<rs_sptr*>(h))->mu(); }
double baz_ref_resamp_ratio(void* h) { return (double)(*static_cast<rs_sptr*>(h))->resamp_ratio(); }

}  // extern "C"
