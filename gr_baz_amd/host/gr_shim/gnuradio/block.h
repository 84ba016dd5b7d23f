/* GNU Radio 3.7 API stand-in: the gr::block surface the fractional-resampler host block touches
 * (general_work / forecast / consume_each / set_relative_rate, message_port_register_in / set_msg_handler).  Used ONLY where GNU Radio is not installed. */
#ifndef GR_BAZ_AMD_SHIM_BLOCK_H
#define GR_BAZ_AMD_SHIM_BLOCK_H

#include <gnuradio/io_signature.h>
#include <gnuradio/types.h>
#include <pmt/pmt.h>

#include <functional>
#include <map>
#include <stdexcept>
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

    /* message ports (basic_block in the real runtime): registration and handler table */
    void message_port_register_in(pmt::pmt_t port_id) { d_msg_handlers[pmt::symbol_to_string(port_id)]; }
    template <class F> void set_msg_handler(pmt::pmt_t which_port, F handler)
    {
        const std::string port = pmt::symbol_to_string(which_port);
        if (!d_msg_handlers.count(port)) throw std::runtime_error("set_msg_handler: port not registered: " + port);
        d_msg_handlers[port] = handler;
    }
    bool has_msg_port(const std::string& port) const { return d_msg_handlers.count(port) != 0; }
    /* shim only: deliver one message now (the real scheduler runs the handler between two general_work() calls) */
    void shim_post(pmt::pmt_t which_port, pmt::pmt_t msg)
    {
        std::map<std::string, std::function<void(pmt::pmt_t)> >::iterator it = d_msg_handlers.find(pmt::symbol_to_string(which_port));
        if (it == d_msg_handlers.end() || !it->second) throw std::runtime_error("shim_post: no handler on that port");
        it->second(msg);
    }

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
    std::map<std::string, std::function<void(pmt::pmt_t)> > d_msg_handlers;
};

}  // namespace gr

namespace gnuradio {
template <class T> inline boost::shared_ptr<T> get_initial_sptr(T* p) { return boost::shared_ptr<T>(p); }
}
#endif
