/* -*- c++ -*- */
/* Fractional resampler, MI355X (gfx950) implementation -- drop-in for gr-baz's gr::baz::fractional_resampler_cc:
 * same class name, make() signature and accessors as /root/reference/lib/baz_fractional_resampler_cc.h:30-60
 * (make(phase_shift, resamp_ratio, resamp_ratio_num = 0, resamp_ratio_denom = 0); mu(), resamp_ratio(), set_mu(),
 * set_resamp_ratio() x3).  The phase state lives in a baz_resamp_ctx (include/baz_resamp_hip.h); no arithmetic here.
 * Both input forms of the reference are served: one input (make2(1, 2, ...), .cc:84), or a second float input that
 * carries the resampling ratio per sample (.cc:205-217; a serial chain on the device, offered for completeness), and
 * the PMT "msg" port (.cc:101-102, handler .cc:109-139). */
#ifndef INCLUDED_BAZ_FRACTIONAL_RESAMPLER_CC_H
#define INCLUDED_BAZ_FRACTIONAL_RESAMPLER_CC_H

#include <gnuradio/block.h>
#include <pmt/pmt.h>

#include <vector>

#ifndef BAZ_API
#define BAZ_API
#endif

namespace gr {
namespace baz {

class BAZ_API fractional_resampler_cc : virtual public block
{
public:
    typedef boost::shared_ptr<fractional_resampler_cc> sptr;

    static sptr make(double phase_shift, double resamp_ratio, unsigned long long resamp_ratio_num = 0,
                     unsigned long long resamp_ratio_denom = 0);

    virtual long double mu() const = 0;
    virtual long double resamp_ratio() const = 0;
    virtual void set_mu(long double mu) = 0;
    virtual void set_resamp_ratio(long double resamp_ratio) = 0;
    virtual void set_resamp_ratio(double resamp_ratio) = 0;
    virtual void set_resamp_ratio(unsigned long long resamp_ratio_num, unsigned long long resamp_ratio_denom) = 0;
    /* the two cases of the reference's "msg" handler (.cc:109-139), and the handler itself */
    virtual void handle_ppb(long whole, double frac) = 0;
    virtual void handle_adjust(double d) = 0;
    virtual void handle_msg(pmt::pmt_t msg) = 0;
};

/* The 129 x 8 tap table of the gnuradio-filter this build sees, recovered through
 * gr::filter::mmse_fir_interpolator_cc::interpolate (empty if its geometry is not 8 taps x 128 steps). */
BAZ_API std::vector<float> recover_mmse_taps();

}  // namespace baz
}  // namespace gr

#endif /* INCLUDED_BAZ_FRACTIONAL_RESAMPLER_CC_H */
