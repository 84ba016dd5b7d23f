/* baz_music_hip.h -- C-ABI of the MI355X (gfx950) MUSIC direction-of-arrival engine.
 *
 * This is the drop-in boundary for ONE path of balint256/gr-baz: the body of
 *     baz_music_doa::work()            /root/reference/lib/baz_music_doa.cc:72-161
 * plus the state that work() reads:
 *     baz_music_doa::baz_music_doa()   /root/reference/lib/baz_music_doa.cc:35-53   (m, n, nsamples, resolution, table)
 *     set_array_response()             /root/reference/lib/baz_music_doa.cc:60-70   (table replacement)
 * The GNU Radio host block (gr_baz_amd/host/baz_music_doa.{h,cc}) keeps the reference's
 * make()/work()/set_array_response() signatures (lib/baz_music_doa.h:36,48,59) and does
 * nothing but marshal gr_complex / float buffers across this ABI.
 *
 * Conventions: plain C types, no exceptions, no torch/GNU Radio types.  The caller owns
 * every host buffer; the library owns all device memory it allocates.  0 == success,
 * negative == error (baz_music_strerror).  A context is single-producer: one thread calls
 * process*(); baz_music_set_table() may be called from any other thread (it is serialised
 * against process*() like the reference's d_mutex, lib/baz_music_doa.cc:67,101).
 *
 * Data layouts (identical to what the reference block sees on its ports):
 *   in        : batch items, each nsamples gr_complex (float re, float im), antenna-
 *               interleaved  x(r,c) = in[c*m + r]              (lib/baz_music_doa.cc:82-84)
 *   table     : resolution x m gr_complex, row-major [bin][antenna]  (array_response_t,
 *               lib/baz_music_doa.h:32-33, as delivered by swig/baz_swig.i:564)
 *   ang, lvl  : batch x n float   (output ports 0 and 1, lib/baz_music_doa.cc:146-155)
 *   spectrum  : batch x resolution float (optional port 2, lib/baz_music_doa.cc:120-121)
 */
#ifndef INCLUDED_BAZ_MUSIC_HIP_H
#define INCLUDED_BAZ_MUSIC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define BAZ_MUSIC_API __attribute__((visibility("default")))
#else
#define BAZ_MUSIC_API
#endif

typedef struct baz_music_ctx baz_music_ctx;

enum {
    BAZ_MUSIC_OK = 0,
    BAZ_MUSIC_E_INVALID = -1,     /* bad argument (the reference only assert()s, .cc:45-50,62-63) */
    BAZ_MUSIC_E_NOMEM = -2,       /* host or device allocation failed */
    BAZ_MUSIC_E_HIP = -3,         /* HIP runtime error (baz_music_last_hip_error) */
    BAZ_MUSIC_E_UNSUPPORTED = -4, /* valid for the reference but not built here (m > BAZ_MUSIC_MAX_M ...) */
    BAZ_MUSIC_E_NODEVICE = -5     /* no gfx950 device / device_id out of range */
};

#define BAZ_MUSIC_MAX_M 64u       /* antennas handled by the gfx950 kernels */
#define BAZ_MUSIC_FAST_M 16u      /* ... by the kernels specialised per m (config 5 uses 16); 17..64 run the run-time-m path */
#define BAZ_MUSIC_MAX_N 63u       /* expected emitters (n < m) */

/* Stage indices for baz_music_stage_ms / baz_music_stage_name. */
enum { BAZ_MUSIC_STAGE_COV = 0, BAZ_MUSIC_STAGE_EVD = 1, BAZ_MUSIC_STAGE_SCAN = 2, BAZ_MUSIC_STAGE_MERGE = 3,
       BAZ_MUSIC_NUM_STAGES = 4 };

/* Replaces baz_music_doa::baz_music_doa (lib/baz_music_doa.cc:35-53).  Validates what the
 * reference only assert()s: m>0, 0<n<m (n==m underflows .cc:93), nsamples>0, nsamples%m==0,
 * resolution>0.  table_ri = resolution*m complex64 as interleaved floats.  device_id < 0
 * selects the current HIP device. */
BAZ_MUSIC_API int baz_music_create(baz_music_ctx** out, uint32_t m, uint32_t n, uint32_t nsamples,
                                   uint32_t resolution, const float* table_ri, int device_id);

/* Replaces baz_music_doa::~baz_music_doa (lib/baz_music_doa.cc:55-58). */
BAZ_MUSIC_API void baz_music_destroy(baz_music_ctx* ctx);

/* Replaces baz_music_doa::set_array_response (lib/baz_music_doa.cc:60-70): takes effect for
 * every item submitted after it returns. Thread-safe against process*().
 * The reference holds d_mutex (.cc:67) for one vector copy (.cc:69).  Here every device image of the new table (bilinear-form
 * table, raw-table operand, f16 pieces of the gated scan, int8 digit planes, ||a||^2) is built BY THE DEVICE on a side stream
 * into a second set of buffers while process*() keeps running on the old one; the lock that serialises against process*() is
 * taken only to exchange the two sets (gr_baz_amd/csrc/table_kernels.hip.h).  A batch sees the old table or the new one, never a
 * mixture.  Concurrent callers of set_table are serialised among themselves. */
BAZ_MUSIC_API int baz_music_set_table(baz_music_ctx* ctx, const float* table_ri);
/* Wall time of the last baz_music_set_table() of this context in milliseconds, and the part of it spent waiting for and
 * holding the lock shared with process*() (what a running work() can be held up by). */
BAZ_MUSIC_API int baz_music_last_retune_ms(baz_music_ctx* ctx, double* total_ms, double* swap_ms);

/* NUMERIC CONTRACT of the float outputs.  The reference stores (float)(1.0 / d), d = ||G^H a||^2 in fp64
 * (lib/baz_music_doa.cc:114-121,153).  Here d is evaluated in fp64 (projector form, literal form near nulls) and the
 * stored value is v_rcp_f32((float)d): within ~2 ulp_f32 (2.4e-7 relative) of the reference's correctly rounded value,
 * against north_star's 1e-5 -- so spectrum / lvl floats are close to, not bit-equal with, a CPU run of the reference.
 * Where the two differ in KIND: for 0 < d < FLT_MIN (1.2e-38) the reference stores a finite 1e38+ value while (float)d
 * is subnormal or 0 and v_rcp_f32 returns +inf; d == 0 gives +inf in both.  lvl[i] == spectrum[bin_i] holds bit for bit,
 * as in the reference.  ang is (float)(bin * 360.0 / resolution), exact.
 * ACROSS WIRINGS: with 6 to 8 antennas lvl is produced by different arithmetic with and without the spectrum port (the int8
 * scan, good to 7.5e-7, against exact fp64 values of the gated scan): lvl of one block agrees between its two wirings to
 * 2 ulp_f32 (tests/test_i8_scan.py pins 1.5e-6), the DoA bins except between bins whose strengths tie that closely.  Up
 * to 5 and from 9 antennas on both wirings give the same bits.
 *
 * Replaces the body of baz_music_doa::work (lib/baz_music_doa.cc:72-161) for `batch`
 * consecutive items held in HOST memory; blocks until ang/lvl/spectrum are filled.
 * lvl and spectrum may be NULL (ports 1 / 2 not wired; guards the reference's NULL-lvl
 * dereference, .cc:147-154). Returns the number of items processed (== batch) or <0. */
BAZ_MUSIC_API int baz_music_process(baz_music_ctx* ctx, const float* in_ri, uint32_t batch,
                                    float* ang, float* lvl, float* spectrum);

/* Same arithmetic on DEVICE-resident buffers (HBM), asynchronous on the context's stream:
 * d_in batch*nsamples complex64; d_ang, d_lvl batch*n float; d_spectrum batch*resolution
 * float or NULL. d_lvl may be NULL. Returns 0 or <0.
 * ORDERING: the launches go to the context's stream -- its own non-blocking stream unless baz_music_set_stream()
 * installed the caller's -- and nothing orders that stream against whatever produced d_in or pre-filled / will read
 * the outputs.  Either make it the producer's stream (set_stream), or call the _on form below, or synchronise. */
BAZ_MUSIC_API int baz_music_process_device(baz_music_ctx* ctx, const void* d_in, uint32_t batch,
                                           void* d_ang, void* d_lvl, void* d_spectrum);

/* The same call with stream semantics relative to `caller_stream` (a hipStream_t; NULL = the legacy default
 * stream): the batch starts after everything enqueued on caller_stream so far, and work enqueued on caller_stream
 * afterwards sees the outputs (two event record / wait pairs; none when caller_stream IS the context's stream). */
BAZ_MUSIC_API int baz_music_process_device_on(baz_music_ctx* ctx, void* caller_stream, const void* d_in,
                                              uint32_t batch, void* d_ang, void* d_lvl, void* d_spectrum);

/* Use an externally owned hipStream_t (e.g. the host framework's current stream) for all
 * subsequent launches; NULL restores the context's own stream (so the legacy default stream, whose
 * handle is NULL, cannot be selected: pass a created stream). */
BAZ_MUSIC_API int baz_music_set_stream(baz_music_ctx* ctx, void* hip_stream);

/* Blocks until everything submitted on the context's stream has finished. */
BAZ_MUSIC_API int baz_music_sync(baz_music_ctx* ctx);

/* Pre-allocates device workspace (covariances, projector coefficients) for `max_batch`
 * items so that process_device() never allocates inside a timed / captured region. */
BAZ_MUSIC_API int baz_music_reserve(baz_music_ctx* ctx, uint32_t max_batch);

/* Per-stage device timing with hipEvents recorded on the launch stream around each
 * kernel of process_device().  enable: 1 = start recording every stage (and reset), 2 = only the dominant
 * (scan) stage -- each recorded event pair costs ~10 us of launch gap --, 0 = stop. */
BAZ_MUSIC_API int baz_music_profile(baz_music_ctx* ctx, int enable);
/* Synchronises, then returns total milliseconds and number of launches recorded for `stage`. */
BAZ_MUSIC_API int baz_music_stage_ms(baz_music_ctx* ctx, int stage, double* total_ms, uint64_t* launches);
/* Kernel (symbol) name launched for `stage` with the context's configuration. */
BAZ_MUSIC_API const char* baz_music_stage_name(baz_music_ctx* ctx, int stage);

/* Test / diagnostic taps (device pointers): run a single stage.
 *   cov : d_in -> d_R      batch * m*m complex128 (row-major R[i][j], (re,im) doubles)   .cc:82-85
 *   evd : d_R  -> d_Q      m*m doubles per item, item-minor: d_Q[e*q_stride + item]; the
 *                          real coefficients of the noise-subspace projector G G^H         .cc:88-93
 * q_stride is returned by baz_music_q_stride() for the given batch.  Past BAZ_MUSIC_FAST_M antennas no projector is
 * formed (the literal form runs straight from the noise eigenvectors): `evd` and `q` return BAZ_MUSIC_E_UNSUPPORTED. */
BAZ_MUSIC_API int baz_music_debug_cov(baz_music_ctx* ctx, const void* d_in, uint32_t batch, void* d_R);
BAZ_MUSIC_API int baz_music_debug_evd(baz_music_ctx* ctx, const void* d_R, uint32_t batch, void* d_Q);
/*   q   : d_in -> d_Q      the two stages back to back, exactly as process_device() runs them for this
 *                          configuration; projector coefficients as for `evd` */
BAZ_MUSIC_API int baz_music_debug_q(baz_music_ctx* ctx, const void* d_in, uint32_t batch, void* d_Q);
/*   coarse margin : the scan that runs when port 2 is NOT wired and m <= 8 (lib/baz_music_doa.cc:97-99: only the top-n list
 *                          is observable then) evaluates every 16-item x 16-bin tile in a coarse f16-matrix-core form first and
 *                          the exact fp64 form only where a tile can still hold a top-n member; ang / lvl are bit-identical to
 *                          the full scan as long as |coarse - exact| <= NG 2^-16 (S + |exact|), NG = 1 (m <= 4) or ceil(m^2 / 32) (gr_baz_amd/csrc/
 *                          scan_coarse_kernels.hip.h).  This tap runs covariance + EVD of the batch and then BOTH forms on every
 *                          (item, bin); *worst = the largest observed error / allowance (sound below 1; derived with a factor
 *                          > 2 to spare).  BAZ_MUSIC_E_UNSUPPORTED for m > 8 or a table whose scale does not fit. */
BAZ_MUSIC_API int baz_music_debug_coarse_margin(baz_music_ctx* ctx, const void* d_in, uint32_t batch, float* worst);
/*   lab statistic: exact (16-item row group x 16-bin tile) evaluations of the coarse-gated scan's launches since the last
 *                          read (the counter resets); -1 unless the context was created under BAZ_MUSIC_COARSE_STATS=1. */
BAZ_MUSIC_API int64_t baz_music_debug_coarse_fired(baz_music_ctx* ctx);
/*   int8 scan   : from 6 to BAZ_MUSIC_FAST_M antennas (n <= 4) the scan evaluates d = a^H Q a on the int8 matrix core with both
 *                          operands cut into balanced base-256 digits -- integer accumulation, no rounding.  Every value is
 *                          first evaluated with four digits (error <= E4 = m^2 Fscale 4.04 2^-30) and keeps that form where it is
 *                          accurate to 7.5e-7; the others take five digits (E5 = m^2 Fscale 5.05 2^-38 + Fscale 2^-36), and where
 *                          even that is not accurate to 7.5e-7, seven (error <= m^2 Fscale 7.07 2^-54 + 2^-53 d: the accuracy class
 *                          of the fp64 form) -- decided per value by that value alone; near-null values take the reference's
 *                          literal form as in the fp64 scan (gr_baz_amd/csrc/scan_i8_kernels.hip.h; BAZ_MUSIC_EXACT=1 at create
 *                          keeps the fp64 scan).  `margin` runs covariance + EVD of the batch and then every form on every
 *                          (item, bin): worst[0] = the largest observed |d5 - d| / E5 (the bound holds while it stays below 1),
 *                          worst[1] = the largest |d7 - d| / (its bound + the fp64 form's own worst-case error), worst[2] = the
 *                          largest |d4 - d| / E4.  `stats` returns and resets the wave tiles that ran the seven-digit form /
 *                          walked since the last read.  `uses_i8_scan`: 1 when that scan is the one this context runs.
 *                          `i8_image` needs no device: the digit images of a table (size returned; written when out_bytes
 *                          suffices: five leading digits, then digits 5 and 6) and params[16] = {7 level weights, 2^54, T, E5,
 *                          refined allowance, 5, 7, E4, T4}. */
BAZ_MUSIC_API int baz_music_debug_i8_margin(baz_music_ctx* ctx, const void* d_in, uint32_t batch, float worst[3]);
BAZ_MUSIC_API int baz_music_debug_i8_stats(baz_music_ctx* ctx, uint64_t* refined_tiles, uint64_t* tiles);
BAZ_MUSIC_API int baz_music_uses_i8_scan(const baz_music_ctx* ctx);
BAZ_MUSIC_API size_t baz_music_debug_i8_image(uint32_t m, uint32_t resolution, const float* table_ri, uint8_t* out,
                                              size_t out_bytes, double* params);
/*   table images : `table_image` copies image `which` of the table IN FORCE back from the device -- 0 the bilinear-form table in
 *                          MFMA B-operand order, 1 the raw table in that order, 2 the gated scan's f16 pieces + fp64 operand, 3 the
 *                          int8 digit planes, 4 ||a||^2 per bin (padded), 5 / 6 the run-time-m path's transposed table and
 *                          ||a||^2, 7 the scalar parameters (BAZ_MUSIC_TABLE_NPARAMS doubles) -- and returns its size in bytes
 *                          (0: no such image for this configuration / table; nothing is written when out_bytes is too small).
 *                          `host_table_image` (needs no device) builds the same image with the round-4 host routines: the
 *                          checker the device builders must agree with byte for byte (tests/test_retune.py). */
#define BAZ_MUSIC_TABLE_NPARAMS 22
BAZ_MUSIC_API size_t baz_music_debug_table_image(baz_music_ctx* ctx, int which, void* out, size_t out_bytes);
BAZ_MUSIC_API size_t baz_music_debug_host_table_image(uint32_t m, uint32_t n, uint32_t resolution, const float* table_ri,
                                                      int which, void* out, size_t out_bytes);
/*   sorting     : LAB BUILD ONLY (libbaz_music_hip_lab.so, BAZ_MUSIC_SORT): measured and NOT shipped (profiles/r05_sort_negative.txt) -- the
 *                          release library has none of its kernels, never orders a batch and keeps no fire statistic; there `sort_state`
 *                          returns five zeros.  In the lab build, without port 2 (m <= 4) the context can order the items of a batch by the
 *                          position of their two deepest nulls (a counting sort on a 16-bit key from a float32 sample of the spectrum;
 *                          gr_baz_amd/csrc/sort_kernels.hip.h) and hand the gated scan the order as an index list; ang / lvl are bit-identical
 *                          either way.  `sort_state`: launches of the gated scan with / without the sort, and the exact evaluations / tile
 *                          pairs walked / sorted flag of the last finished one. */
BAZ_MUSIC_API int baz_music_debug_sort_state(baz_music_ctx* ctx, uint64_t out[5]);
/*   guard zones : LAB BUILD ONLY, process environment BAZ_MUSIC_GUARD=1: every device buffer of the library lies between two 64-KiB zones
 *                          filled with a pattern (and is itself pre-filled with it).  `guard_check` synchronises the device, compares the
 *                          zones of every live buffer and returns the number of zones found overwritten since the process started (a free
 *                          checks its buffer too) -- 0 means no kernel wrote outside a buffer of this library; details go to stderr.
 *                          `guard_active`: 1 when the guard is on.  Both return 0 in the release library. */
BAZ_MUSIC_API int baz_music_debug_guard_check(void);
BAZ_MUSIC_API int baz_music_debug_guard_active(void);
/*   (host only) the bin ranges per item the int8 scan launches with: whole rounds of the `slots` resident workgroups. */
BAZ_MUSIC_API uint32_t baz_music_debug_i8_nsplit(uint32_t batch, uint32_t nsteps, uint32_t slots);
BAZ_MUSIC_API uint32_t baz_music_q_stride(uint32_t batch);

/* Algorithmic HBM bytes per item: 8*nsamples + 8*n + 4*resolution (the last
 * term only when the spectrum port is wired). */
BAZ_MUSIC_API uint64_t baz_music_bytes_per_item(const baz_music_ctx* ctx, int with_spectrum);

BAZ_MUSIC_API const char* baz_music_strerror(int code);
/* hipGetErrorString of the last failing HIP call in this context ("" if none). */
BAZ_MUSIC_API const char* baz_music_last_hip_error(const baz_music_ctx* ctx);
BAZ_MUSIC_API const char* baz_music_version(void);
/* Number of usable gfx950 devices (0 when none) and the device a context lives on.  The host block deals its
 * instances over the devices round-robin (instance i -> device i mod count, SURVEY.md 8e: stream s -> GPU s mod G)
 * unless BAZ_MUSIC_DEVICE pins one. */
/* OPT-IN extension, not reference behaviour: mode 1 makes ang/lvl the n strongest LOCAL MAXIMA
 * of the pseudo-spectrum (bin b with s[b] > s[b-1] and s[b] >= s[b+1] on the circle) instead of the reference's n
 * strongest bins (lib/baz_music_doa.cc:129-141, which usually are neighbours on one lobe).  Same output format,
 * descending strength, (0, 0) for missing peaks.  Mode 0 (default) is the reference.  Mode 1 is offered up to
 * BAZ_MUSIC_FAST_M antennas (BAZ_MUSIC_E_UNSUPPORTED beyond). */
BAZ_MUSIC_API int baz_music_set_peak_mode(baz_music_ctx* ctx, int mode);
/* Statistic: how many (item, bin) values of the LAST process call were recomputed in the reference's literal form
 * ||G^H a||^2 because the projector form a^H Q a put them at or below ~m 1e-8 max||a||^2 (near-nulls of the noise
 * subspace, SNR >~ 55 dB); blocks until that call is done (a host-fed call cut into chunks reports their sum).
 * -1 on error.  See DESIGN.md 2 (near-nulls).  (The name is kept from round 1, which redid whole items.) */
BAZ_MUSIC_API int64_t baz_music_refined_values(baz_music_ctx* ctx);
/* The same number under its round-1 name (round 1 redid whole ITEMS; since round 2 the unit is one (item, bin) value, so
 * do not compare it with a batch size).  Past BAZ_MUSIC_FAST_M antennas only the matrix-core scan (n <= 8) counts; the
 * general wide scan evaluates the literal form near nulls per bin and keeps no count (0).  Without the spectrum port only values that could still enter
 * the top-n list are evaluated at all, so fewer are counted than with it. */
BAZ_MUSIC_API int64_t baz_music_refined_items(baz_music_ctx* ctx);
BAZ_MUSIC_API int baz_music_device_count(void);
BAZ_MUSIC_API int baz_music_device(const baz_music_ctx* ctx);

/* Host-fed path: page-locking of the CALLER's buffers.  baz_music_process() copies straight
 * between the caller's host memory and HBM; when that memory is pageable the runtime stages every copy through its own
 * bounce buffers (measured: 1.1e6 items/s pageable against 2.5e6 page-locked for 8-MiB calls of config 2).  A
 * scheduler's stream buffers live as long as the flowgraph and are handed to work() over and over, so they are worth
 * locking once:
 *   baz_music_host_register(ctx, p, bytes)   page-locks [p, p+bytes) for this context (hipHostRegister on exactly
 *       that range, or on its union with the registrations of this context it overlaps or touches: a buffer may be
 *       registered piecewise, and may map the same physical pages twice like GNU Radio's circular buffers; every
 *       request ends up inside ONE registration, because the runtime rejects copies that are only partly inside one).
 *       0 when the range is locked (or already was, also by its owner: hipHostMalloc / torch pinned memory);
 *       BAZ_MUSIC_E_HIP when the runtime refuses it (remembered; asked again after 1,024 further requests for it);
 *       BAZ_MUSIC_E_UNSUPPORTED when the context's limit (BAZ_MUSIC_PIN_LIMIT_MIB, default 4096) would be exceeded or
 *       the union would have to replace a registration another context shares.  A range that cannot be locked as
 *       a whole is left entirely pageable (its own registrations it touches are dropped); process() works on it either way.
 *       Registrations are process-wide and counted: a range inside another context's registration takes a share of
 *       that one (two blocks on one stream buffer), and the memory is unlocked when the last holder lets go.
 *   baz_music_set_host_pinning(ctx, 1)       makes baz_music_process() do that for the input and spectrum ranges of
 *       every call before it copies (a lookup per call once they are known).  Default 0: the caller must guarantee that the
 *       memory outlives the registration -- true for scheduler buffers, not for temporaries.
 *   (A call below 64 MiB of traffic (BAZ_MUSIC_SINGLE_MIB) whose input and spectrum are page-locked -- by this or by their owner -- runs
 *   without copies: the kernels address the caller's buffers over PCIe, hipHostGetDevicePointer; BAZ_MUSIC_ZERO_COPY=0
 *   keeps the copies.)
 *   baz_music_host_unregister_all(ctx)       gives up every registration (share) of this context (also done by destroy);
 *       call it before the buffers are unmapped (the host block does in stop()).
 *   baz_music_host_pinned_bytes(ctx)         bytes this context holds locked. */
BAZ_MUSIC_API int baz_music_host_register(baz_music_ctx* ctx, const void* p, size_t bytes);
BAZ_MUSIC_API int baz_music_set_host_pinning(baz_music_ctx* ctx, int enable);
BAZ_MUSIC_API int baz_music_host_unregister_all(baz_music_ctx* ctx);
BAZ_MUSIC_API uint64_t baz_music_host_pinned_bytes(baz_music_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* INCLUDED_BAZ_MUSIC_HIP_H */
