/* -*- c++ -*- */
/* AGC block, MI355X (gfx950) implementation -- drop-in for gr-baz's baz_agc_cc: same factory, defaults and
 * ports as /root/reference/lib/baz_agc_cc.h:36-66 (baz_make_agc_cc(rate = 1e-4, reference = 1.0, gain = 1.0,
 * max_gain = 0.0); in: gr_complex; out: gr_complex, optional float env, optional float gain).  The state
 * (_env, _count) lives in a baz_agc_ctx (include/baz_agc_hip.h); no CPU arithmetic here. */
#ifndef INCLUDED_BAZ_AGC_CC_H
#define INCLUDED_BAZ_AGC_CC_H

#include <gnuradio/sync_block.h>

#ifndef BAZ_API
#define BAZ_API
#endif

struct baz_agc_ctx;

class BAZ_API baz_agc_cc;
typedef boost::shared_ptr<baz_agc_cc> baz_agc_cc_sptr;

BAZ_API baz_agc_cc_sptr baz_make_agc_cc(float rate = 1e-4, float reference = 1.0, float gain = 1.0, float max_gain = 0.0);

class BAZ_API baz_agc_cc : public gr::sync_block
{
    friend BAZ_API baz_agc_cc_sptr baz_make_agc_cc(float rate, float reference, float gain, float max_gain);
    baz_agc_cc(float rate, float reference, float gain, float max_gain);

public:
    ~baz_agc_cc();
    virtual int work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items);

private:
    baz_agc_ctx* d_ctx;
};

#endif /* INCLUDED_BAZ_AGC_CC_H */
