/* -*- c++ -*- */
/* Host side of the MI355X MUSIC-DoA block: argument checking, table flattening and buffer
 * marshalling across the C-ABI (include/baz_music_hip.h).  Mirrors the roles of
 * /root/reference/lib/baz_music_doa.cc:29-33 (factory), :35-53 (constructor/ports/banner),
 * :60-70 (setter) and :72-161 (work); all arithmetic of :74-155 runs in the HIP kernels. */
#ifdef HAVE_CONFIG_H
#include "config.h"
#endif

#include <baz_music_doa.h>
#include <baz_music_hip.h>

#include <gnuradio/io_signature.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

namespace {

/* resolution x m nested vectors -> the contiguous [bin][antenna] (re,im) float image of the ABI */
std::vector<float> flatten_response(const array_response_t& table, unsigned int m, unsigned int resolution)
{
    if (table.size() != resolution)
        throw std::invalid_argument("music_doa: array_response must have `resolution` rows");
    std::vector<float> flat((size_t)resolution * m * 2);
    for (unsigned int s = 0; s < resolution; ++s) {
        if (table[s].size() != m)
            throw std::invalid_argument("music_doa: every array_response row must have m entries");
        for (unsigned int t = 0; t < m; ++t) {
            flat[2 * ((size_t)s * m + t)] = table[s][t].real();
            flat[2 * ((size_t)s * m + t) + 1] = table[s][t].imag();
        }
    }
    return flat;
}

void check_config(unsigned int m, unsigned int n, unsigned int nsamples, unsigned int resolution)
{
    if (m == 0) throw std::invalid_argument("music_doa: m must be > 0");
    if (n == 0 || n >= m) throw std::invalid_argument("music_doa: need 0 < n < m");
    if (nsamples == 0 || (nsamples % m) != 0)
        throw std::invalid_argument("music_doa: nsamples must be a positive multiple of m");
    if (resolution == 0) throw std::invalid_argument("music_doa: resolution must be > 0");
}

}  // namespace

/* Independent block instances (BASELINE config 4: 64 streams in one flowgraph) are dealt over the node's GPUs
 * round-robin -- instance i -> device i mod G -- unless BAZ_MUSIC_DEVICE pins one; -1 = the current
 * HIP device when no gfx950 device is visible (create() then reports the error). */
int baz_music_doa_deal_device(unsigned int instance, int device_count)
{
    return device_count > 0 ? (int)(instance % (unsigned)device_count) : -1;
}

static int next_device()
{
    if (const char* v = getenv("BAZ_MUSIC_DEVICE")) return atoi(v);
    static std::atomic<unsigned> s_instances(0);
    return baz_music_doa_deal_device(s_instances.fetch_add(1), baz_music_device_count());
}

static long env_long(const char* name, long dflt, long lo, long hi)
{
    const char* v = getenv(name);
    if (!v || !*v) return dflt;
    char* end = NULL;
    const long x = strtol(v, &end, 10);
    if (end == v) return dflt;
    return x < lo ? lo : (x > hi ? hi : x);
}

baz_music_doa_sptr baz_make_music_doa(unsigned int m, unsigned int n, unsigned int nsamples,
                                      const array_response_t& array_response, unsigned int resolution)
{
    check_config(m, n, nsamples, resolution);   /* before the io_signature sizes are formed */
    return baz_music_doa_sptr(new baz_music_doa(m, n, nsamples, array_response, resolution));
}

baz_music_doa::baz_music_doa(unsigned int m, unsigned int n, unsigned int nsamples,
                             const array_response_t& array_response, unsigned int resolution)
    : gr::sync_block("music_doa",
                     gr::io_signature::make(1, 1, nsamples * sizeof(gr_complex)),
                     gr::io_signature::make3(1, 3, n * sizeof(float), n * sizeof(float),
                                             resolution * sizeof(float))),
      d_m(m), d_n(n), d_nsamples(nsamples), d_resolution(resolution),
      d_array_response(array_response), d_ctx(NULL), d_pin_buffers(false)
{
    const std::vector<float> flat = flatten_response(array_response, m, resolution);
    const int rc = baz_music_create(&d_ctx, m, n, nsamples, resolution, flat.data(), next_device());
    if (rc == BAZ_MUSIC_E_INVALID || rc == BAZ_MUSIC_E_UNSUPPORTED)
        throw std::invalid_argument(std::string("music_doa: ") + baz_music_strerror(rc));
    if (rc != BAZ_MUSIC_OK)
        throw std::runtime_error(std::string("music_doa: cannot open the gfx950 engine: ") + baz_music_strerror(rc));

    /* Scheduler hints.  The reference handles ONE item per work() call (.cc:74,160); here a call
     * is one launch sequence over all its items, so the block wants LARGE calls -- without giving up what the reference
     * guarantees: every item of a finite stream is processed, and an item is processed as soon as it has arrived.
     * GNU Radio sizes a buffer for 2 x (output multiple + history) items of every reader
     * (flat_flowgraph::allocate_buffer); the default is 64 KiB = 8 cfg2 items.  Two ways to ask for more:
     *   set_output_multiple(N)     also makes N the MINIMUM call: N items must have arrived before work() runs, and what
     *                              does not fill a last multiple when a finite source ends is never processed;
     *   set_history(H + 1)         declares H items of look-back the block never reads: the upstream buffer grows to
     *                              2 (H + 2) items, calls of up to ~H items form whenever the block is the bottleneck,
     *                              a single item is still a valid call, and nothing is lost at the end of a stream: the
     *                              runtime preloads the H look-back items as zeros (buffer_add_reader(.., history - 1)),
     *                              output i is computed from input i + H, i.e. from real item i.
     * The block uses the second: BAZ_MUSIC_INPUT_LOOKBACK = H (0 = none; default 2048 items, fewer where items are large:
     * H = 32 MiB / the larger of an input item and a spectrum row, between 8 and 2048 -- the doubly mapped circular buffers
     * are 2 (H + 2) items and must fit /dev/shm: config 2 gets 2048 (33.6 MB in, 59 MB spectrum), config 3 (144 KB rows)
     * 233, 64 antennas x 256 columns (128 KiB items) 256.  Round 5 doubled it: a host-fed call is one launch sequence that reads
     * its input over the link and then writes its spectrum over it, and its fixed costs shrink with the call -- config 2 with port 2 on
     * page-locked buffers: 2.05e6 items/s in 1,024-item calls, 2.24e6 in 2,048-item calls; without port 2 4.9e6 / 5.7e6
     * (profiles/r05_hostfed_calls.txt; cutting a call into overlapped sub-chunks LOSES on this stack, same file)), output multiple 1,
     * set_min_output_buffer(2 H) so that the output side admits the same calls (a call takes at most half a buffer).
     * BAZ_MUSIC_OUTPUT_MULTIPLE (1), BAZ_MUSIC_MIN_OUTPUT_BUFFER, BAZ_MUSIC_MAX_NOUTPUT (0 = no cap) override.
     * What the runtime makes of them (call sizes per work()) is modelled in gr_shim/gnuradio/flowgraph_model.h;
     * INTEGRATION.md 5 has the memory these requests cost and the measured rates. */
    const size_t big_item = std::max<size_t>((size_t)nsamples * sizeof(gr_complex), (size_t)resolution * sizeof(float));
    const long lookback_dflt = (long)std::max<size_t>(8, std::min<size_t>(2048, ((size_t)32 << 20) / big_item));
    const long lookback = env_long("BAZ_MUSIC_INPUT_LOOKBACK", lookback_dflt, 0, 1 << 20);
    const long multiple = env_long("BAZ_MUSIC_OUTPUT_MULTIPLE", 1, 1, 1 << 20);
    const long min_buffer = env_long("BAZ_MUSIC_MIN_OUTPUT_BUFFER", std::max(2 * lookback, multiple > 1 ? 8 * multiple : 0L), 0, 1L << 30);
    const long cap = env_long("BAZ_MUSIC_MAX_NOUTPUT", 0, 0, 1L << 30);
    set_history((unsigned)lookback + 1);
    set_output_multiple((int)multiple);
    if (min_buffer > 0) set_min_output_buffer(min_buffer);
    if (cap > 0) set_max_noutput_items((int)std::max(multiple, cap - cap % multiple));   /* the runtime does not round a cap */

    set_pin_buffers(env_long("BAZ_MUSIC_PIN_BUFFERS", 1, 0, 1) != 0);

    fprintf(stderr, "[%s<%li>] MUSIC DOA: M: %d, N: %d, # samples: %d, angular resolution: %d\n",
            name().c_str(), unique_id(), m, n, nsamples, resolution);
}

baz_music_doa::~baz_music_doa()
{
    baz_music_destroy(d_ctx);
}

int baz_music_doa::device() const { return baz_music_device(d_ctx); }

void baz_music_doa::set_pin_buffers(bool on)
{
    if (!on) (void)baz_music_host_unregister_all(d_ctx);
    (void)baz_music_set_host_pinning(d_ctx, on ? 1 : 0);
    d_pin_buffers = on;
}

unsigned long long baz_music_doa::pinned_bytes() const { return baz_music_host_pinned_bytes(d_ctx); }

bool baz_music_doa::start() { return true; }

bool baz_music_doa::stop()
{
    (void)baz_music_host_unregister_all(d_ctx);   /* the flowgraph's buffers go away after this */
    return true;
}

void baz_music_doa::set_peak_mode(bool local_maxima)
{
    const int rc = baz_music_set_peak_mode(d_ctx, local_maxima ? 1 : 0);
    if (rc != BAZ_MUSIC_OK) throw std::runtime_error(std::string("music_doa: set_peak_mode: ") + baz_music_strerror(rc));
}

void baz_music_doa::set_array_response(const array_response_t& array_response)
{
    const std::vector<float> flat = flatten_response(array_response, d_m, d_resolution);
    fprintf(stderr, "[%s<%li>] Updating array response\n", name().c_str(), unique_id());

    gr::thread::scoped_lock guard(d_mutex);
    const int rc = baz_music_set_table(d_ctx, flat.data());   /* serialised against work() inside */
    if (rc != BAZ_MUSIC_OK)
        throw std::runtime_error(std::string("music_doa: set_array_response: ") + baz_music_strerror(rc));
    d_array_response = array_response;
}

array_response_t baz_music_doa::array_response()
{
    gr::thread::scoped_lock guard(d_mutex);
    return d_array_response;
}

int baz_music_doa::work(int noutput_items, gr_vector_const_void_star& input_items,
                        gr_vector_void_star& output_items)
{
    if (noutput_items <= 0) return 0;
    if (input_items.empty() || output_items.empty()) return -1;

    /* input_items[0] points at the OLDEST look-back item (history() - 1 of them precede the item output 0 belongs to) */
    const float* in = static_cast<const float*>(input_items[0]) + (size_t)(history() - 1) * d_nsamples * 2;   /* gr_complex == (float re, float im) */
    float* ang = static_cast<float*>(output_items[0]);
    float* lvl = (output_items.size() > 1) ? static_cast<float*>(output_items[1]) : NULL;
    float* spectrum = (output_items.size() > 2) ? static_cast<float*>(output_items[2]) : NULL;

    const int rc = baz_music_process(d_ctx, in, (uint32_t)noutput_items, ang, lvl, spectrum);
    if (rc < 0) {
        fprintf(stderr, "[%s<%li>] MUSIC DOA: device error: %s (%s)\n", name().c_str(), unique_id(),
                baz_music_strerror(rc), baz_music_last_hip_error(d_ctx));
        return -1;   /* WORK_DONE */
    }
    return noutput_items;
}
