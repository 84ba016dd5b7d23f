/* -*- c++ -*- */
/* Fractional resampler, MI355X (gfx950) implementation -- drop-in for gr-baz's gr::baz::fractional_resampler_cc:
 * same class name, make() signature and accessors as /root/reference/lib/baz_fractional_resampler_cc.h:30-60
 * (make(phase_shift, resamp_ratio, resamp_ratio_num = 0, resamp_ratio_denom = 0); mu(), resamp_ratio(), set_mu(),
 * set_resamp_ratio() x3).  The phase state lives in a baz_resamp_ctx (include/baz_resamp_hip.h); no arithmetic here.
 * Differences, both deliberate: (1) one input only -- the optional per-sample ratio input (io_signature make2(1, 2),
 * .cc:84) is a data-dependent serial chain and is not offered; (2) the PMT "msg" port (.cc:99-100) needs the GNU Radio
 * runtime: on a real host add the two lines back and route them to handle_ppb()/handle_adjust() below. */
#ifndef INCLUDED_BAZ_FRACTIONAL_RESAMPLER_CC_H
#define INCLUDED_BAZ_FRACTIONAL_RESAMPLER_CC_H

#include <gnuradio/block.h>

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
    /* the two cases of the reference's "msg" handler (.cc:109-139) */
    virtual void handle_ppb(long whole, double frac) = 0;
    virtual void handle_adjust(double d) = 0;
};

}  // namespace baz
}  // namespace gr

#endif /* INCLUDED_BAZ_FRACTIONAL_RESAMPLER_CC_H */
