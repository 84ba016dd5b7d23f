/* baz_agc_hip.h -- C-ABI of the MI355X (gfx950) AGC engine (BASELINE config 5 front-end).
 *
 * Drop-in boundary for baz_agc_cc (gr-baz):
 *     baz_agc_cc::baz_agc_cc()   /root/reference/lib/baz_agc_cc.cc:50-62   (rate, reference, gain, max_gain, state)
 *     baz_agc_cc::work()         /root/reference/lib/baz_agc_cc.cc:64-102  (the live part; :103-149 is dead code
 *                                                                            behind an unconditional `continue`)
 * One context holds `nstreams` independent AGC states (one gr-baz block instance each), so a multi-antenna
 * front-end is one launch sequence.  Plain C types, no exceptions; 0 == OK, negative == error
 * (codes shared with baz_music_hip.h).  Layout: stream-major, stream s occupies [s*stride, s*stride + n).
 */
#ifndef INCLUDED_BAZ_AGC_HIP_H
#define INCLUDED_BAZ_AGC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define BAZ_AGC_API __attribute__((visibility("default")))
#else
#define BAZ_AGC_API
#endif

typedef struct baz_agc_ctx baz_agc_ctx;

enum { BAZ_AGC_OK = 0, BAZ_AGC_E_INVALID = -1, BAZ_AGC_E_NOMEM = -2, BAZ_AGC_E_HIP = -3, BAZ_AGC_E_NODEVICE = -5 };

/* Replaces baz_make_agc_cc(rate, reference, gain, max_gain) (lib/baz_agc_cc.cc:44-62; defaults 1e-4, 1.0, 1.0,
 * 0.0 at lib/baz_agc_cc.h:41).  `gain` and `max_gain` only feed the reference's dead code and are kept for
 * signature compatibility.  device_id < 0 selects the current HIP device. */
BAZ_AGC_API int baz_agc_create(baz_agc_ctx** out, uint32_t nstreams, float rate, float reference, float gain,
                               float max_gain, int device_id);
BAZ_AGC_API void baz_agc_destroy(baz_agc_ctx* ctx);

/* Replaces work() for n samples per stream held in HOST memory (blocks until the outputs are filled).
 * in_ri/out_ri: complex64 as interleaved floats; env, mul: float per sample or NULL (output ports 1, 2,
 * lib/baz_agc_cc.cc:68-69).  State (_env, _count) carries over to the next call like the reference's members.
 * Returns n or <0. */
BAZ_AGC_API int baz_agc_process(baz_agc_ctx* ctx, const float* in_ri, uint64_t n, uint64_t stride, float* out_ri,
                                float* env, float* mul);
/* Same on DEVICE-resident buffers, asynchronous on the context's stream.  Returns 0 or <0. */
BAZ_AGC_API int baz_agc_process_device(baz_agc_ctx* ctx, const void* d_in, uint64_t n, uint64_t stride, void* d_out,
                                       void* d_env, void* d_mul);
/* Config-5 form (the AGC sits right in front of MUSIC-DoA): same arithmetic and state as baz_agc_process_device, but
 * the nstreams (<= 16) per-antenna outputs are written already INTERLEAVED, d_items[t * nstreams + s] = out_s[t] --
 * i.e. as baz_music_doa items of nsamples = nstreams * K (x(r,c) = in[c*m + r], lib/baz_music_doa.cc:82-84) -- which is
 * what GNU Radio's stock interleave / streams_to_vector blocks would produce between the two gr-baz blocks.  No env /
 * gain ports in this form.  Returns 0 or <0. */
BAZ_AGC_API int baz_agc_process_device_interleaved(baz_agc_ctx* ctx, const void* d_in, uint64_t n, uint64_t stride,
                                                   void* d_items);
/* Restarts every stream (count = 0, env = 0), i.e. a freshly constructed block. */
BAZ_AGC_API int baz_agc_reset(baz_agc_ctx* ctx);
BAZ_AGC_API int baz_agc_set_stream(baz_agc_ctx* ctx, void* hip_stream);
BAZ_AGC_API int baz_agc_sync(baz_agc_ctx* ctx);
/* Number of samples consumed per stream so far (the reference's _count). */
BAZ_AGC_API uint64_t baz_agc_count(const baz_agc_ctx* ctx);
BAZ_AGC_API const char* baz_agc_strerror(int code);
/* Test tap: full tiles of ordinary values run a lean square root and division (the rsq / rcp + fma cores of the rounded
 * ones, without their scaling: gr_baz_amd/csrc/agc_kernels.hip.h).  This compares them with the rounded library functions
 * on n device-resident doubles a[i] (sqrt(a[i])) and pairs (a[i] / b[i]), bit for bit, wherever the fast path would take
 * them (2^-400 <= v <= 2^400); mismatches[0] = differing square roots, mismatches[1] = differing quotients. */
BAZ_AGC_API int baz_agc_debug_selfcheck(baz_agc_ctx* ctx, const void* d_a, const void* d_b, uint64_t n, uint64_t mismatches[2]);

#ifdef __cplusplus
}
#endif
#endif /* INCLUDED_BAZ_AGC_HIP_H */
