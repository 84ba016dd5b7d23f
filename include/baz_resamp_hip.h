/* baz_resamp_hip.h -- C-ABI of the MI355X (gfx950) fractional resampler (BASELINE config 5
 * front-end).
 *
 * Drop-in boundary for gr::baz::fractional_resampler_cc (gr-baz):
 *     make() / constructor   /root/reference/lib/baz_fractional_resampler_cc.cc:73-101  (phase_shift, resamp_ratio,
 *                                                                                        resamp_ratio_num/denom)
 *     forecast()             /root/reference/lib/baz_fractional_resampler_cc.cc:141-149
 *     general_work()         /root/reference/lib/baz_fractional_resampler_cc.cc:152-203  (the one-input branch)
 *     mu()/resamp_ratio()/set_mu()/set_resamp_ratio()   .cc:220-254 ; "msg" port handler .cc:109-139
 * One context resamples `nstreams` streams in lock step with ONE shared phase accumulator (the antennas of an array
 * share their sample clock), stream-major layout: stream s occupies [s*stride, s*stride + n).
 *
 * Arithmetic: out[o] = sum_k in[ii_o + k] * taps[imu_o][7-k] (float accumulation), imu_o = rint((float)mu_o * 128).
 * The 8-tap x 129-phase table belongs to gnuradio-filter's MMSE interpolator, which gr-baz does not vendor: a fresh
 * context starts from the table regenerated from the library's published criterion (closed-form least squares,
 * gr_baz_amd/csrc/baz_resamp_hip.hip: build_taps, rounded to the six significant digits the library's generator prints)
 * -- PARITY UNPINNED for that default: the library's header is not available offline, and the default may differ from it
 * by a few 1e-6 per tap (tests/test_resamp.py) --
 * and baz_resamp_set_taps() installs the host's own table, which the host block does wherever it is compiled against
 * a real gnuradio-filter: bit-exact with whatever gnuradio-filter the host has.  The reference's x87 `long double` phase recurrence mu <- frac(mu + mu_inc),
 * ii <- ii + floor(mu + mu_inc) is evaluated in closed form, P_o = mu_0 + o * mu_inc in 64.64-bit fixed point
 * (128-bit integers), one output per thread.  This is EXACTLY the reference's sequence whenever its sums are exact
 * in the 64-bit x87 mantissa -- always for ratios and phases given as `double` (make()'s signature) with
 * ratio >= 2^-11; for num/denom ratios (64-bit quotient) the two differ by < o * 2^-64 in mu, which moves imu only
 * when mu*128 sits within that distance of a rounding boundary.
 *
 * The two-input branch (.cc:205-217, per-sample ratio input) is served by baz_resamp_process2*: ii_{o+1} depends on
 * the ratio sample read at ii_o, a data-dependent serial chain that one lane walks (see resamp_walk_kernel) -- offered
 * for completeness of the block's surface, not for throughput.
 * Plain C types, no exceptions; 0 == OK, negative == error (codes shared with baz_music_hip.h).
 */
#ifndef INCLUDED_BAZ_RESAMP_HIP_H
#define INCLUDED_BAZ_RESAMP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define BAZ_RESAMP_API __attribute__((visibility("default")))
#else
#define BAZ_RESAMP_API
#endif

typedef struct baz_resamp_ctx baz_resamp_ctx;

enum { BAZ_RESAMP_OK = 0, BAZ_RESAMP_E_INVALID = -1, BAZ_RESAMP_E_NOMEM = -2, BAZ_RESAMP_E_HIP = -3,
       BAZ_RESAMP_E_UNSUPPORTED = -4, BAZ_RESAMP_E_NODEVICE = -5 };
enum { BAZ_RESAMP_NTAPS = 8, BAZ_RESAMP_NSTEPS = 128 };

/* Replaces fractional_resampler_cc::make(phase_shift, resamp_ratio, resamp_ratio_num = 0, resamp_ratio_denom = 0)
 * (.cc:73-78).  resamp_ratio = input_rate / output_rate; denom != 0 selects num/denom (.cc:89-92).
 * E_INVALID for ratio <= 0 or phase_shift outside [0, 1] (the reference throws std::out_of_range, .cc:94-97);
 * E_UNSUPPORTED for ratios below 2^-11 or above 2^31.  device_id < 0 selects the current HIP device. */
BAZ_RESAMP_API int baz_resamp_create(baz_resamp_ctx** out, uint32_t nstreams, double phase_shift, double resamp_ratio,
                                     uint64_t resamp_ratio_num, uint64_t resamp_ratio_denom, int device_id);
BAZ_RESAMP_API void baz_resamp_destroy(baz_resamp_ctx* ctx);

/* forecast(): input samples general_work needs for `noutput` outputs = ceil(noutput * ratio + 8) (.cc:141-149). */
BAZ_RESAMP_API int64_t baz_resamp_forecast(const baz_resamp_ctx* ctx, uint32_t noutput);

/* Replaces general_work() on HOST buffers: up to `noutput` outputs per stream from `ninput` available input samples
 * per stream (complex64 as interleaved floats).  Produces min(noutput, what ninput allows: every output reads 8
 * consecutive inputs); *consumed = consume_each() (advance the input by this much before the next call).
 * Returns the number of outputs produced per stream, or <0. */
BAZ_RESAMP_API int64_t baz_resamp_process(baz_resamp_ctx* ctx, const float* in_ri, uint64_t in_stride, uint64_t ninput,
                                          float* out_ri, uint64_t out_stride, uint32_t noutput, uint64_t* consumed);
/* Same on DEVICE-resident buffers, asynchronous on the context's stream (the returned counts are computed on the
 * host from the phase state and are valid immediately). */
BAZ_RESAMP_API int64_t baz_resamp_process_device(baz_resamp_ctx* ctx, const void* d_in, uint64_t in_stride,
                                                 uint64_t ninput, void* d_out, uint64_t out_stride, uint32_t noutput,
                                                 uint64_t* consumed);

/* The two-input branch of general_work() (.cc:205-217): `ratio` is the second input port, one float per INPUT sample
 * (at least ninput of them), shared by the nstreams lock-stepped streams; after output o the ratio sample at the
 * current input index becomes d_mu_inc.  Same return / *consumed convention; production also stops -- like at the end
 * of the input -- at a ratio sample that is not a finite number in (0, 2^31] (the output AT that sample is still produced,
 * *consumed points at it); a call whose window STARTS on such a sample returns BAZ_RESAMP_E_INVALID (the host block then
 * ends with a message) instead of repeating one output and consuming nothing for ever.  The pending set_mu / set_ratio /
 * adjustment requests stay pending (the reference only honours them in the one-input branch).  The device form
 * synchronises the context's stream before returning (the counts depend on the data). */
BAZ_RESAMP_API int64_t baz_resamp_process2(baz_resamp_ctx* ctx, const float* in_ri, uint64_t in_stride, uint64_t ninput,
                                           const float* ratio, float* out_ri, uint64_t out_stride, uint32_t noutput,
                                           uint64_t* consumed);
BAZ_RESAMP_API int64_t baz_resamp_process2_device(baz_resamp_ctx* ctx, const void* d_in, uint64_t in_stride,
                                                  uint64_t ninput, const void* d_ratio, void* d_out,
                                                  uint64_t out_stride, uint32_t noutput, uint64_t* consumed);

/* Deferred setters with the reference's ordering (.cc:165-189): a new mu applies to the first output of the next
 * call, a new ratio from the first phase step of the next call on, the adjustment is added to that step once. */
BAZ_RESAMP_API int baz_resamp_set_mu(baz_resamp_ctx* ctx, double mu);                                   /* .cc:233-238 */
BAZ_RESAMP_API int baz_resamp_set_ratio(baz_resamp_ctx* ctx, double resamp_ratio);                      /* .cc:240-245 */
BAZ_RESAMP_API int baz_resamp_set_ratio_rational(baz_resamp_ctx* ctx, uint64_t num, uint64_t denom);    /* .cc:247-254 */
BAZ_RESAMP_API int baz_resamp_set_ratio_ppb(baz_resamp_ctx* ctx, long whole, double frac);              /* msg pair, .cc:116-124 */
BAZ_RESAMP_API int baz_resamp_adjust(baz_resamp_ctx* ctx, double d);                                    /* msg double, .cc:127-134 */
BAZ_RESAMP_API double baz_resamp_mu(const baz_resamp_ctx* ctx);                                         /* .cc:220-224 */
BAZ_RESAMP_API double baz_resamp_ratio(const baz_resamp_ctx* ctx);                                      /* .cc:226-230 */
/* 1 when every phase quantity so far was exactly representable in 64.64 fixed point (see the header comment). */
BAZ_RESAMP_API int baz_resamp_phase_exact(const baz_resamp_ctx* ctx);
/* The 129 x 8 tap table in use (float, host copy). */
BAZ_RESAMP_API const float* baz_resamp_taps(const baz_resamp_ctx* ctx);
/* The table a fresh context starts with: the closed-form MMSE solution at six significant digits (see the header
 * comment), written to out[129 * 8].  Pure host arithmetic, no device needed. */
BAZ_RESAMP_API void baz_resamp_default_taps(float* out);
/* Replaces the table: taps[imu * 8 + j] = tap j of phase imu as gnuradio-filter's interpolator stores it (interpolate()
 * forms sum_k in[k] * taps[imu][7 - k]; /root/reference/lib/baz_fractional_resampler_cc.cc:172,203 call it).  This is how
 * a host that HAS gnuradio-filter pins the arithmetic: the host block reads the installed library's table out of
 * gr::filter::mmse_fir_interpolator_cc (8 unit impulses x 129 phases) and passes it here, after which the engine's
 * outputs are those of that library, operation for operation.  Takes effect for the next process call (the context's
 * stream is drained first).  E_INVALID for NULL or non-finite entries. */
BAZ_RESAMP_API int baz_resamp_set_taps(baz_resamp_ctx* ctx, const float* taps);

BAZ_RESAMP_API int baz_resamp_set_stream(baz_resamp_ctx* ctx, void* hip_stream);
BAZ_RESAMP_API int baz_resamp_sync(baz_resamp_ctx* ctx);
BAZ_RESAMP_API const char* baz_resamp_strerror(int code);

#ifdef __cplusplus
}
#endif
#endif /* INCLUDED_BAZ_RESAMP_HIP_H */
