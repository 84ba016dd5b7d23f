// ORACLE-ONLY API SHIM for gr::filter::mmse_fir_interpolator_cc (gnuradio-filter 3.7, NOT vendored under
// /root/reference; call sites lib/baz_fractional_resampler_cc.cc:28,41,87,172,206).  Restates its published
// algorithm: 8 taps x 129 phases, imu = rint(mu*128), float dot product with the reversed tap row.  The tap
// table is the closed-form solution of the MMSE criterion GNU Radio's gen_interpolator_taps optimises
// numerically (see oracle/resamp_ref.c, whose resamp_ref_taps() this shim calls) -- PARITY UNPINNED.
#ifndef BAZ_ORACLE_MMSE_INTERP_SHIM
#define BAZ_ORACLE_MMSE_INTERP_SHIM
#include <gnuradio/sync_block.h>
#include <cmath>
extern "C" void resamp_ref_taps(float taps[129][8]);
namespace gr { namespace filter {
class mmse_fir_interpolator_cc {
public:
    mmse_fir_interpolator_cc() { resamp_ref_taps(d_taps); }
    unsigned ntaps() const { return 8; }
    unsigned nsteps() const { return 128; }
    gr_complex interpolate(const gr_complex input[], float mu) const
    {
        const int imu = (int)rintf(mu * 128);
        if (imu < 0 || imu > 128) throw std::runtime_error("mmse_fir_interpolator_cc: imu out of bounds.");
        const float* t = d_taps[imu];
        float re = 0.0f, im = 0.0f;
        for (int k = 0; k < 8; ++k) { re += input[k].real() * t[7 - k]; im += input[k].imag() * t[7 - k]; }
        return gr_complex(re, im);
    }
private:
    float d_taps[129][8];
};
} }
#endif
