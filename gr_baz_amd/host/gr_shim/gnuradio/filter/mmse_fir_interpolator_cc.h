/* Stand-in for gnuradio-filter 3.7's gr::filter::mmse_fir_interpolator_cc, used ONLY where GNU Radio is not installed
 * (this container, the GPU box); on a GNU Radio host this directory is left off the include path and the real header
 * is found.  The class the reference builds one output sample with (/root/reference/lib/
 * baz_fractional_resampler_cc.cc:28,41,87,172,203): 8 taps x 129 phases, imu = rint(mu * 128), float dot product with
 * the reversed tap row.  The host block does NOT compute samples with it -- it reads the library's tap table out of it
 * at construction (recover_mmse_taps() in baz_fractional_resampler_cc.cc) and hands that table to the gfx950 engine, so
 * that on a real host the engine interpolates with exactly the table of the gnuradio-filter that is installed.  This
 * stand-in's table is the engine's own closed-form table (baz_resamp_default_taps), which makes the recovery a no-op
 * here -- and lets tests/test_resamp.py check the recovery bit for bit without a GPU. */
#ifndef GR_BAZ_AMD_SHIM_MMSE_FIR_INTERPOLATOR_CC_H
#define GR_BAZ_AMD_SHIM_MMSE_FIR_INTERPOLATOR_CC_H

#include <gnuradio/types.h>

#include <baz_resamp_hip.h>

#include <cmath>
#include <stdexcept>

namespace gr {
namespace filter {

class mmse_fir_interpolator_cc
{
public:
    mmse_fir_interpolator_cc() { baz_resamp_default_taps(&d_taps[0][0]); }
    unsigned ntaps() const { return BAZ_RESAMP_NTAPS; }
    unsigned nsteps() const { return BAZ_RESAMP_NSTEPS; }
    gr_complex interpolate(const gr_complex input[], float mu) const
    {
        const int imu = (int)rintf(mu * (float)BAZ_RESAMP_NSTEPS);
        if (imu < 0 || imu > BAZ_RESAMP_NSTEPS) throw std::runtime_error("mmse_fir_interpolator_cc: imu out of bounds.");
        const float* t = d_taps[imu];
        float re = 0.0f, im = 0.0f;
        for (int k = 0; k < BAZ_RESAMP_NTAPS; ++k) {
            re += input[k].real() * t[BAZ_RESAMP_NTAPS - 1 - k];
            im += input[k].imag() * t[BAZ_RESAMP_NTAPS - 1 - k];
        }
        return gr_complex(re, im);
    }

private:
    float d_taps[BAZ_RESAMP_NSTEPS + 1][BAZ_RESAMP_NTAPS];
};

}  // namespace filter
}  // namespace gr
#endif
