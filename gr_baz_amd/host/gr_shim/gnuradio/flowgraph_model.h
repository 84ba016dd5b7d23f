/* GNU Radio 3.7 stand-in, part 2: how the runtime would DRIVE a sync block.
 *
 * The host block's scheduler hints (set_output_multiple / set_min_output_buffer / set_max_noutput_items) only mean
 * something through what GNU Radio's runtime does with them.  With no GNU Radio in this image, this header restates
 * the two pieces of gnuradio-runtime 3.7 that turn hints into work() calls, so that the block can be run -- on the GPU
 * box too -- the way a flowgraph `source -> block -> sinks` would run it:
 *
 *   buffer sizing     flat_flowgraph::allocate_buffer + buffer::allocate_buffer: 64 KiB per output port by default
 *                     (2 x GR_FIXED_BUFFER_SIZE), at least two output multiples, clamped by the block's max / raised to
 *                     its min output buffer, at least 2 x (decimation x multiple + history) of every downstream block,
 *                     rounded up to page_size / gcd(item size, page_size) items;
 *   circular buffers  vmcircbuf: every buffer is mapped twice back to back, so a work() call sees contiguous items
 *                     across the wrap; a buffer holds at most bufsize - 1 items;
 *   one iteration     block_executor::run_one_iteration for a fixed-rate block: output space = min over the ports of
 *                     min(round_down(space, multiple), round_down(bufsize / 2, multiple)); the items the input holds,
 *                     rounded down to the multiple, replace it when smaller; capped by max_noutput_items (never below
 *                     one multiple); halved while the input cannot cover noutput + history - 1.
 *
 * The neighbours are idealised: the source never starves the block (it refills the input buffer, in calls of at most
 * half a buffer like any block, whenever the block looks) and the sinks never stall it (they drain every call's
 * output at once).  That is the regime in which the hints decide the call size.  Items that do not fill a last
 * output multiple when the source ends are dropped, as in GNU Radio.  A reader with history h starts h - 1 zero items
 * behind the write pointer (buffer_add_reader's nzero_preload), so it sees every real item as the NEWEST of a window.  Restated from the published behaviour of
 * gnuradio-runtime 3.7 (lib/flat_flowgraph.cc, lib/buffer.cc, lib/block_executor.cc); no GNU Radio source is
 * available here, so this is a model, and tests/test_scheduler_model.py pins its arithmetic on hand-computed cases. */
#ifndef GR_BAZ_AMD_SHIM_FLOWGRAPH_MODEL_H
#define GR_BAZ_AMD_SHIM_FLOWGRAPH_MODEL_H

#include <gnuradio/types.h>

#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace gr {
namespace shim {

static const long FIXED_BUFFER_SIZE = 32 * (1L << 10);   /* GR_FIXED_BUFFER_SIZE */

struct downstream_t {
    double decimation;   /* 1 / relative_rate of the reader */
    int multiple;        /* its output multiple */
    int history;         /* its history */
};

inline long gcd_long(long a, long b) { return b ? gcd_long(b, a % b) : a; }
inline long round_down(long n, long m) { return n - n % m; }
inline long round_up(long n, long m) { return round_down(n + m - 1, m); }

/* items of one output buffer (flat_flowgraph::allocate_buffer, then buffer::allocate_buffer's granularity) */
inline long buffer_items(long item_size, int output_multiple, long min_output_buffer, long max_output_buffer,
                         const std::vector<downstream_t>& readers, long page_size = 0)
{
    if (item_size <= 0 || output_multiple <= 0) throw std::invalid_argument("buffer_items");
    if (page_size <= 0) page_size = sysconf(_SC_PAGESIZE);
    long nitems = FIXED_BUFFER_SIZE * 2 / item_size;
    if (nitems < 2L * output_multiple) nitems = 2L * output_multiple;
    if (max_output_buffer > 0) {
        nitems = std::min(nitems, max_output_buffer);
        nitems -= nitems % output_multiple;
        if (nitems < 1) throw std::runtime_error("problems allocating a buffer with the given max output buffer constraint!");
    } else if (min_output_buffer > 0) {
        nitems = std::max(nitems, min_output_buffer);
        nitems -= nitems % output_multiple;
        if (nitems < 1) throw std::runtime_error("problems allocating a buffer with the given min output buffer constraint!");
    }
    for (size_t i = 0; i < readers.size(); ++i)
        nitems = std::max(nitems, (long)(2 * (readers[i].decimation * readers[i].multiple + readers[i].history)));
    return round_up(nitems, page_size / gcd_long(item_size, page_size));
}

/* one doubly mapped stream buffer */
class circ_buffer {
public:
    circ_buffer(long nitems, long item_size) : d_base(NULL), d_bufsize(nitems), d_item(item_size), d_wr(0), d_rd(0)
    {
        const size_t bytes = (size_t)nitems * (size_t)item_size;
        if (bytes == 0 || bytes % (size_t)sysconf(_SC_PAGESIZE)) throw std::invalid_argument("circ_buffer: not a page multiple");
        const int fd = memfd_create("gr_shim_vmcircbuf", 0);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) {
            if (fd >= 0) close(fd);
            throw std::runtime_error("circ_buffer: memfd");
        }
        void* both = mmap(NULL, 2 * bytes, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        bool ok = both != MAP_FAILED;
        ok = ok && mmap(both, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED, fd, 0) != MAP_FAILED;
        ok = ok && mmap((char*)both + bytes, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED, fd, 0) != MAP_FAILED;
        close(fd);
        if (!ok) {
            if (both != MAP_FAILED) munmap(both, 2 * bytes);
            throw std::runtime_error("circ_buffer: mmap");
        }
        d_base = (char*)both;
    }
    ~circ_buffer()
    {
        if (d_base) munmap(d_base, 2 * (size_t)d_bufsize * (size_t)d_item);
    }
    long bufsize() const { return d_bufsize; }
    long item_size() const { return d_item; }
    char* base() const { return d_base; }
    size_t mapped_bytes() const { return 2 * (size_t)d_bufsize * (size_t)d_item; }
    long items_available() const { return (d_wr - d_rd + d_bufsize) % d_bufsize; }
    long space_available() const { return d_bufsize - items_available() - 1; }
    char* write_pointer() const { return d_base + (size_t)d_wr * (size_t)d_item; }
    const char* read_pointer() const { return d_base + (size_t)d_rd * (size_t)d_item; }
    void produce(long n) { d_wr = (d_wr + n) % d_bufsize; }
    void consume(long n) { d_rd = (d_rd + n) % d_bufsize; }

private:
    circ_buffer(const circ_buffer&);
    circ_buffer& operator=(const circ_buffer&);
    char* d_base;
    long d_bufsize, d_item, d_wr, d_rd;
};

struct run_stats {
    long calls, items, dropped_at_end, in_bufsize;
    std::vector<long> out_bufsize;
    std::map<long, long> call_sizes;   /* noutput_items -> number of work() calls */
    double work_seconds, total_seconds;
    long steady_items;             /* items and work() time after the first pass over the source data (passes > 1): */
    double steady_work_seconds;    /* buffers touched, page locks taken, device workspace allocated */
    int last_return;
    run_stats() : calls(0), items(0), dropped_at_end(0), in_bufsize(0), work_seconds(0), total_seconds(0), steady_items(0),
                  steady_work_seconds(0), last_return(0) {}
};

/* noutput_items of one executor iteration; 0 = blocked (on input, or on output space) */
inline long plan_noutput(long items_in, const std::vector<long>& out_space, const std::vector<long>& out_bufsize, int multiple,
                         int history, bool max_is_set, int block_max_noutput, long top_max_noutput = 100000000L)
{
    long noutput = -1;
    for (size_t i = 0; i < out_space.size(); ++i) {
        const long n = std::min(round_down(out_space[i], multiple), round_down(out_bufsize[i] / 2, multiple));
        if (n < 1) return 0;                                   /* blocked on output */
        noutput = noutput < 0 ? n : std::min(noutput, n);
    }
    if (noutput < 0) return 0;
    const long reqd = round_down(std::max(0L, items_in - (history - 1)), multiple);   /* fixed_rate_ninput_to_noutput */
    if (reqd > 0 && reqd <= noutput) noutput = reqd;
    long max_noutput = max_is_set ? (long)block_max_noutput : top_max_noutput;
    max_noutput = std::max((long)multiple, max_noutput);
    noutput = std::min(noutput, max_noutput);
    while (noutput + history - 1 > items_in) {                  /* forecast() not covered: try half */
        noutput >>= 1;
        if (noutput < multiple) return 0;                       /* blocked on input */
        noutput = round_up(noutput, multiple);
    }
    return noutput;
}

/* source -> blk -> (n_outputs sinks).  `src` holds n_items input items, replayed `passes` times; sinks[i] (may be NULL)
 * receives port i of the first pass. */
template <class Block>
run_stats run_sync_block(Block& blk, const char* src, long n_items, int n_outputs, char* const* sinks,
                         unsigned long long* pinned_bytes_at_stop = NULL, int passes = 1)
{
    typedef std::chrono::steady_clock clk;
    const int multiple = blk.output_multiple(), history = blk.history();
    const long in_item = blk.input_signature()->sizeof_stream_item(0);
    std::vector<downstream_t> me(1);
    me[0].decimation = 1.0 / blk.relative_rate();
    me[0].multiple = multiple;
    me[0].history = history;
    circ_buffer in(buffer_items(in_item, 1, -1, -1, me), in_item);       /* a source with no hints of its own */
    /* buffer_add_reader(buffer, nzero_preload = history - 1, ..): the reader starts history - 1 items BEHIND the write
     * pointer, over zeroed memory -- the look-back of the first real item */
    in.produce(history - 1);
    std::vector<circ_buffer*> out;
    struct cleanup {
        std::vector<circ_buffer*>& v;
        ~cleanup() { for (size_t i = 0; i < v.size(); ++i) delete v[i]; }
    } guard = {out};
    const std::vector<downstream_t> sink_readers(1, downstream_t{1.0, 1, 1});
    for (int p = 0; p < n_outputs; ++p) {
        const long sz = blk.output_signature()->sizeof_stream_item(p);
        out.push_back(new circ_buffer(buffer_items(sz, multiple, blk.min_output_buffer(), blk.max_output_buffer(), sink_readers), sz));
    }
    run_stats st;
    st.in_bufsize = in.bufsize();
    for (size_t i = 0; i < out.size(); ++i) st.out_bufsize.push_back(out[i]->bufsize());

    const clk::time_point t_begin = clk::now();
    if (!blk.start()) throw std::runtime_error("run_sync_block: start() failed");
    struct stopper {   /* stop() runs before the buffers are unmapped, on every path */
        Block& b;
        unsigned long long* pinned;
        ~stopper()
        {
            if (pinned) *pinned = b.pinned_bytes();
            b.stop();
        }
    } stop_guard = {blk, pinned_bytes_at_stop};
    long fed = 0;
    const long total_items = n_items * (long)std::max(1, passes);
    std::vector<long> space(out.size()), sizes(out.size());
    gr_vector_const_void_star in_ptrs(1);
    gr_vector_void_star out_ptrs(out.size());
    for (;;) {
        while (fed < total_items && in.space_available() > 0) {     /* the source's work() calls */
            const long at = fed % n_items;
            const long n = std::min(std::min(in.space_available(), std::max(1L, in.bufsize() / 2)), n_items - at);
            std::memcpy(in.write_pointer(), src + (size_t)at * (size_t)in_item, (size_t)n * (size_t)in_item);
            in.produce(n);
            fed += n;
        }
        for (size_t i = 0; i < out.size(); ++i) {
            space[i] = out[i]->space_available();
            sizes[i] = out[i]->bufsize();
        }
        const long noutput = plan_noutput(in.items_available(), space, sizes, multiple, history,
                                          blk.is_set_max_noutput_items(), blk.max_noutput_items());
        if (noutput == 0) {                                     /* blocked on input and the source is done */
            st.dropped_at_end = in.items_available() - (history - 1);   /* (the look-back items stay in the buffer) */
            break;
        }
        in_ptrs[0] = in.read_pointer();
        for (size_t i = 0; i < out.size(); ++i) out_ptrs[i] = out[i]->write_pointer();
        const clk::time_point t0 = clk::now();
        const int produced = blk.work((int)noutput, in_ptrs, out_ptrs);
        const double dt = std::chrono::duration<double>(clk::now() - t0).count();
        st.work_seconds += dt;
        if (produced > 0 && st.items >= n_items) {
            st.steady_items += produced;
            st.steady_work_seconds += dt;
        }
        st.last_return = produced;
        if (produced < 0) break;                                /* WORK_DONE */
        ++st.calls;
        ++st.call_sizes[noutput];
        st.items += produced;
        in.consume(produced);
        for (size_t i = 0; i < out.size(); ++i) {               /* the sinks */
            out[i]->produce(produced);
            const long first = st.items - produced, keep = std::min((long)produced, n_items - first);
            if (sinks && sinks[i] && keep > 0)
                std::memcpy(sinks[i] + (size_t)first * (size_t)out[i]->item_size(), out[i]->read_pointer(),
                            (size_t)keep * (size_t)out[i]->item_size());
            out[i]->consume(produced);
        }
    }
    st.total_seconds = std::chrono::duration<double>(clk::now() - t_begin).count();
    return st;
}

}  // namespace shim
}  // namespace gr
#endif
