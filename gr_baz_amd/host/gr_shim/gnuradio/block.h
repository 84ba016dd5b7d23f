/* GNU Radio 3.7 API stand-in: the gr::block surface the fractional-resampler host block touches
 * (general_work / forecast / consume_each / set_relative_rate).  Used ONLY where GNU Radio is not installed. */
#ifndef GR_BAZ_AMD_SHIM_BLOCK_H
#define GR_BAZ_AMD_SHIM_BLOCK_H

#include <gnuradio/io_signature.h>
#include <gnuradio/types.h>

#include <string>

namespace gr {

class block {
public:
    virtual ~block() {}
    std::string name() const { return d_name; }
    long unique_id() const { return d_unique_id; }
    io_signature::sptr input_signature() const { return d_input_signature; }
    io_signature::sptr output_signature() const { return d_output_signature; }

    virtual void forecast(int noutput_items, gr_vector_int& ninput_items_required) = 0;
    virtual int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                             gr_vector_void_star& output_items) = 0;

    void consume_each(int how_many_items) { d_consumed = how_many_items; }
    void set_relative_rate(double relative_rate) { d_relative_rate = relative_rate; }
    double relative_rate() const { return d_relative_rate; }
    /* shim only: what the last general_work() passed to consume_each() (the real runtime advances the read pointers) */
    int last_consumed() const { return d_consumed; }

protected:
    block() : d_unique_id(-1), d_consumed(0), d_relative_rate(1.0) {}
    block(const std::string& name, io_signature::sptr input_signature, io_signature::sptr output_signature)
        : d_name(name), d_input_signature(input_signature), d_output_signature(output_signature),
          d_unique_id(next_unique_id()), d_consumed(0), d_relative_rate(1.0)
    {
    }

private:
    static long next_unique_id()
    {
        static long s_next = 100000;
        return s_next++;
    }
    std::string d_name;
    io_signature::sptr d_input_signature, d_output_signature;
    long d_unique_id;
    int d_consumed;
    double d_relative_rate;
};

}  // namespace gr

namespace gnuradio {
template <class T> inline boost::shared_ptr<T> get_initial_sptr(T* p) { return boost::shared_ptr<T>(p); }
}
#endif
