/* -*- c++ -*- */
/* Host side of the MI355X AGC block: buffer marshalling across include/baz_agc_hip.h.  Mirrors
 * /root/reference/lib/baz_agc_cc.cc:44-62 (factory, ports "gr_agc_cc": in 1 x gr_complex; out 1..3 x
 * {gr_complex, float, float}) and :64-102 (work; the arithmetic runs in the HIP kernels). */
#include <baz_agc_cc.h>
#include <baz_agc_hip.h>

#include <gnuradio/io_signature.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

baz_agc_cc_sptr baz_make_agc_cc(float rate, float reference, float gain, float max_gain)
{
    return baz_agc_cc_sptr(new baz_agc_cc(rate, reference, gain, max_gain));
}

baz_agc_cc::baz_agc_cc(float rate, float reference, float gain, float max_gain)
    : gr::sync_block("gr_agc_cc", gr::io_signature::make(1, 1, sizeof(gr_complex)),
                     gr::io_signature::make2(1, 3, sizeof(gr_complex), sizeof(float))),
      d_ctx(NULL)
{
    const int rc = baz_agc_create(&d_ctx, 1, rate, reference, gain, max_gain, -1);
    if (rc != BAZ_AGC_OK)
        throw std::runtime_error(std::string("agc_cc: cannot open the gfx950 engine: ") + baz_agc_strerror(rc));
    /* Scheduler hints, as for the MUSIC block: a call is two copies and three launches whatever its
     * size, and GNU Radio's default 64-KiB buffers would hand work() at most 4,096 samples (gr_shim/gnuradio/
     * flowgraph_model.h) -- less than the launches cost.  The block declares BAZ_AGC_INPUT_LOOKBACK = 16,384 samples of
     * history it never reads (the upstream buffer grows to 2 x that, 256 KiB; calls of up to 16,384 samples form when the
     * block is the bottleneck) and asks for output buffers of twice that.  The output multiple stays 1: every
     * sample of a finite capture is processed -- the reference's AGC processes every sample -- and a short capture or a
     * slow source still gets its output at once.  BAZ_AGC_OUTPUT_MULTIPLE / BAZ_AGC_MIN_OUTPUT_BUFFER override
     * (lookback 0, multiple 1, buffer 0 = the reference's scheduling). */
    long lookback = 16384, multiple = 1, min_buffer = -1;
    if (const char* v = getenv("BAZ_AGC_INPUT_LOOKBACK")) lookback = std::max(0L, std::min(1L << 24, atol(v)));
    if (const char* v = getenv("BAZ_AGC_OUTPUT_MULTIPLE")) multiple = std::max(1L, std::min(1L << 24, atol(v)));
    if (const char* v = getenv("BAZ_AGC_MIN_OUTPUT_BUFFER")) min_buffer = std::max(0L, std::min(1L << 30, atol(v)));
    if (min_buffer < 0) min_buffer = std::max(2 * lookback, multiple > 1 ? 8 * multiple : 0L);
    set_history((unsigned)lookback + 1);
    set_output_multiple((int)multiple);
    if (min_buffer > 0) set_min_output_buffer(min_buffer);
}

baz_agc_cc::~baz_agc_cc()
{
    baz_agc_destroy(d_ctx);
}

int baz_agc_cc::work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items)
{
    if (noutput_items <= 0) return 0;
    const float* in = static_cast<const float*>(input_items[0]) + (size_t)(history() - 1) * 2;   /* past the look-back samples */
    float* out = static_cast<float*>(output_items[0]);
    float* env = (output_items.size() >= 2) ? static_cast<float*>(output_items[1]) : NULL;   /* .cc:68 */
    float* mul = (output_items.size() >= 3) ? static_cast<float*>(output_items[2]) : NULL;   /* .cc:69 */
    const int rc = baz_agc_process(d_ctx, in, (uint64_t)noutput_items, (uint64_t)noutput_items, out, env, mul);
    if (rc < 0) {
        fprintf(stderr, "[%s<%li>] AGC: device error: %s\n", name().c_str(), unique_id(), baz_agc_strerror(rc));
        return -1;
    }
    return noutput_items;
}
