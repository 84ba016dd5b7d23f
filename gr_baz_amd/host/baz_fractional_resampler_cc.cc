/* -*- c++ -*- */
/* Host side of the MI355X fractional resampler: buffer marshalling across include/baz_resamp_hip.h.  Mirrors
 * /root/reference/lib/baz_fractional_resampler_cc.cc:73-101 (make, constructor: block name, ports, ratio rules,
 * banner, relative rate, "msg" port), :109-139 (handle_msg), :141-149 (forecast), :152-217 (general_work, both
 * branches; the arithmetic runs in the HIP kernels) and :220-254 (accessors, deferred setters). */
#include <baz_fractional_resampler_cc.h>
#include <baz_resamp_hip.h>

#include <gnuradio/filter/mmse_fir_interpolator_cc.h>
#include <gnuradio/io_signature.h>

#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

namespace gr {
namespace baz {

/* The tap table of the gnuradio-filter this block is compiled against, read out of the very class the reference
 * computes its samples with (.cc:87 constructs it, .cc:172,203 call interpolate()): phase imu is selected with
 * mu = imu / nsteps (exact in float), a unit impulse at input[k] returns taps[imu][ntaps - 1 - k] exactly (the other
 * seven products are 0 * tap = 0, the sums add zeros).  Returns an empty vector when the library's geometry is not the
 * 8 x 128 the engine is built for (then the engine keeps its closed-form table). */
std::vector<float> recover_mmse_taps()
{
    gr::filter::mmse_fir_interpolator_cc interp;
    const unsigned ntaps = interp.ntaps(), nsteps = interp.nsteps();
    if (ntaps != (unsigned)BAZ_RESAMP_NTAPS || nsteps != (unsigned)BAZ_RESAMP_NSTEPS) return std::vector<float>();
    std::vector<float> taps((size_t)(nsteps + 1) * ntaps);
    std::vector<gr_complex> impulse(ntaps);
    for (unsigned imu = 0; imu <= nsteps; ++imu)
        for (unsigned k = 0; k < ntaps; ++k) {
            for (unsigned j = 0; j < ntaps; ++j) impulse[j] = gr_complex(j == k ? 1.0f : 0.0f, 0.0f);
            taps[(size_t)imu * ntaps + (ntaps - 1 - k)] = interp.interpolate(impulse.data(), (float)imu / (float)nsteps).real();
        }
    return taps;
}

class fractional_resampler_cc_impl : public fractional_resampler_cc
{
    baz_resamp_ctx* d_ctx;

public:
    fractional_resampler_cc_impl(double phase_shift, double resamp_ratio, unsigned long long num, unsigned long long denom)
        : block("fractional_resampler_cc", io_signature::make2(1, 2, sizeof(gr_complex), sizeof(float)),   /* .cc:84 */
                io_signature::make(1, 1, sizeof(gr_complex))),
          d_ctx(NULL)
    {
        const int rc = baz_resamp_create(&d_ctx, 1, phase_shift, resamp_ratio, num, denom, -1);
        if (rc == BAZ_RESAMP_E_INVALID)   /* .cc:94-97 */
            throw std::out_of_range("resampling ratio must be > 0 and phase shift ratio must be >= 0 and <= 1");
        if (rc != BAZ_RESAMP_OK)
            throw std::runtime_error(std::string("fractional_resampler_cc: cannot open the gfx950 engine: ") +
                                     baz_resamp_strerror(rc));
        /* interpolate with the table of the gnuradio-filter on this host (the stand-in's table is the engine's own) */
        const std::vector<float> taps = recover_mmse_taps();
        if (!taps.empty() && baz_resamp_set_taps(d_ctx, taps.data()) != BAZ_RESAMP_OK)
            fprintf(stderr, "[fractional_resampler_cc] the host's MMSE tap table was rejected; using the closed-form table\n");
        set_relative_rate(1.0 / baz_resamp_ratio(d_ctx));   /* .cc:99 */

        message_port_register_in(pmt::mp("msg"));           /* .cc:101-102 */
        set_msg_handler(pmt::mp("msg"), [this](pmt::pmt_t msg) { this->handle_msg(msg); });
    }
    ~fractional_resampler_cc_impl() { baz_resamp_destroy(d_ctx); }

    void forecast(int noutput_items, gr_vector_int& ninput_items_required)   /* .cc:141-149 */
    {
        const int need = (int)baz_resamp_forecast(d_ctx, (uint32_t)(noutput_items > 0 ? noutput_items : 0));
        for (size_t i = 0; i < ninput_items_required.size(); ++i) ninput_items_required[i] = need;
    }

    int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                     gr_vector_void_star& output_items)
    {
        if (noutput_items <= 0) return 0;
        uint64_t consumed = 0;
        int64_t produced;
        if (ninput_items.size() == 1) {           /* .cc:162-203 */
            produced = baz_resamp_process(d_ctx, static_cast<const float*>(input_items[0]),
                                          (uint64_t)ninput_items[0], (uint64_t)ninput_items[0],
                                          static_cast<float*>(output_items[0]), (uint64_t)noutput_items,
                                          (uint32_t)noutput_items, &consumed);
        } else {                                  /* .cc:205-217: the second input carries the ratio, per sample */
            const uint64_t nin = (uint64_t)(ninput_items[0] < ninput_items[1] ? ninput_items[0] : ninput_items[1]);
            produced = baz_resamp_process2(d_ctx, static_cast<const float*>(input_items[0]), nin, nin,
                                           static_cast<const float*>(input_items[1]), static_cast<float*>(output_items[0]),
                                           (uint64_t)noutput_items, (uint32_t)noutput_items, &consumed);
        }
        if (produced < 0) {
            fprintf(stderr, "[%s<%li>] device error: %s\n", name().c_str(), unique_id(), baz_resamp_strerror((int)produced));
            return -1;
        }
        set_relative_rate(1.0 / baz_resamp_ratio(d_ctx));   /* .cc:178 (after a ratio update) */
        consume_each((int)consumed);                         /* .cc:201 */
        return (int)produced;
    }

    long double mu() const { return baz_resamp_mu(d_ctx); }                 /* .cc:220-224 */
    long double resamp_ratio() const { return baz_resamp_ratio(d_ctx); }    /* .cc:226-230 */
    void set_mu(long double mu) { check(baz_resamp_set_mu(d_ctx, (double)mu)); }
    void set_resamp_ratio(long double r) { check(baz_resamp_set_ratio(d_ctx, (double)r)); }
    void set_resamp_ratio(double r) { set_resamp_ratio((long double)r); }
    void set_resamp_ratio(unsigned long long num, unsigned long long denom) { check(baz_resamp_set_ratio_rational(d_ctx, num, denom)); }
    void handle_ppb(long whole, double frac) { check(baz_resamp_set_ratio_ppb(d_ctx, whole, frac)); }
    void handle_adjust(double d) { check(baz_resamp_adjust(d_ctx, d)); }

    /* the "msg" port's handler, .cc:109-139: a pair (long . double) is a ratio in parts per billion, a bare number a
     * one-off phase adjustment in units of the ratio; anything else (or a value the engine rejects) is reported and
     * dropped, like the reference's catch (...) */
    void handle_msg(pmt::pmt_t msg)
    {
        try {
            if (pmt::is_pair(msg)) {
                const long i = pmt::to_long(pmt::car(msg));
                const double frac = pmt::to_double(pmt::cdr(msg));
                handle_ppb(i, frac);
            } else {
                handle_adjust(pmt::to_double(msg));
            }
        } catch (...) {
            fprintf(stderr, "Failed to handle PMT\n");
        }
    }

private:
    static void check(int rc)
    {
        if (rc != BAZ_RESAMP_OK) throw std::out_of_range(std::string("fractional_resampler_cc: ") + baz_resamp_strerror(rc));
    }
};

fractional_resampler_cc::sptr fractional_resampler_cc::make(double phase_shift, double resamp_ratio,
                                                            unsigned long long resamp_ratio_num,
                                                            unsigned long long resamp_ratio_denom)
{
    return gnuradio::get_initial_sptr(new fractional_resampler_cc_impl(phase_shift, resamp_ratio, resamp_ratio_num, resamp_ratio_denom));
}

}  // namespace baz
}  // namespace gr
