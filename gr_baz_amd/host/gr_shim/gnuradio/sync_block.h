/* Minimal stand-in for the GNU Radio 3.7 runtime API that the MUSIC-DoA host block touches
 * (SURVEY.md Appendix E).  Used ONLY where GNU Radio is not installed (this container, the GPU
 * box): it lets gr_baz_amd/host/baz_music_doa.{h,cc} compile and be driven by tests exactly as the
 * scheduler would drive it.  On a real GNU Radio host this directory is simply not put on the
 * include path and <gnuradio/...> resolves to the real headers. */
#ifndef GR_BAZ_AMD_SHIM_SYNC_BLOCK_H
#define GR_BAZ_AMD_SHIM_SYNC_BLOCK_H

#include <gnuradio/io_signature.h>
#include <gnuradio/types.h>

#include <string>

namespace gr {

class sync_block {
public:
    virtual ~sync_block() {}

    std::string name() const { return d_name; }
    long unique_id() const { return d_unique_id; }
    io_signature::sptr input_signature() const { return d_input_signature; }
    io_signature::sptr output_signature() const { return d_output_signature; }

    /* scheduler hints (recorded; flowgraph_model.h acts on them the way the real runtime does) */
    void set_output_multiple(int multiple) { d_output_multiple = multiple; }
    int output_multiple() const { return d_output_multiple; }
    void set_max_noutput_items(int m) { d_max_noutput_items = m; d_max_noutput_items_set = true; }
    int max_noutput_items() const { return d_max_noutput_items; }
    bool is_set_max_noutput_items() const { return d_max_noutput_items_set; }
    void set_min_output_buffer(long min_output_buffer) { d_min_output_buffer = min_output_buffer; }
    long min_output_buffer() const { return d_min_output_buffer; }
    void set_max_output_buffer(long max_output_buffer) { d_max_output_buffer = max_output_buffer; }
    long max_output_buffer() const { return d_max_output_buffer; }
    unsigned history() const { return d_history; }
    void set_history(unsigned history) { d_history = history; }
    double relative_rate() const { return 1.0; }   /* sync block */

    /* called by the runtime when the flowgraph starts / stops (gr::block::start / stop) */
    virtual bool start() { return true; }
    virtual bool stop() { return true; }

    /* 1:1 rate block: returns the number of items produced == consumed on every input; -1 = done */
    virtual int work(int noutput_items, gr_vector_const_void_star& input_items,
                     gr_vector_void_star& output_items) = 0;

protected:
    sync_block(const std::string& name, io_signature::sptr input_signature, io_signature::sptr output_signature)
        : d_name(name), d_input_signature(input_signature), d_output_signature(output_signature),
          d_unique_id(next_unique_id()), d_output_multiple(1), d_max_noutput_items(0), d_max_noutput_items_set(false),
          d_min_output_buffer(-1), d_max_output_buffer(-1), d_history(1)
    {
    }

private:
    static long next_unique_id()
    {
        static long s_next = 1;
        return s_next++;
    }
    std::string d_name;
    io_signature::sptr d_input_signature, d_output_signature;
    long d_unique_id;
    int d_output_multiple, d_max_noutput_items;
    bool d_max_noutput_items_set;
    long d_min_output_buffer, d_max_output_buffer;
    unsigned d_history;
};

}  // namespace gr
#endif
