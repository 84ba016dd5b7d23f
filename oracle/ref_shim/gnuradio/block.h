// ORACLE-ONLY API SHIM (test infrastructure): the gr::block surface that
// /root/reference/lib/baz_fractional_resampler_cc.{h,cc} touch -- constructor, name()/unique_id(),
// forecast()/general_work(), consume_each(), set_relative_rate(), and the message-port calls of
// .cc:99-100 (registered, never dispatched here: the oracle calls handle_msg's setters directly).
#ifndef BAZ_ORACLE_GR_BLOCK_SHIM
#define BAZ_ORACLE_GR_BLOCK_SHIM

#include <gnuradio/sync_block.h>
#include <pmt/pmt.h>
#include <cmath>
#include <functional>

typedef std::vector<int> gr_vector_int;

namespace boost {
// boost::bind(&C::method, this, _1) as used at .cc:100
struct shim_placeholder1 {};
template <class C> inline std::function<void(pmt::pmt_t)> bind(void (C::*m)(pmt::pmt_t), C* self, shim_placeholder1)
{
    return [self, m](pmt::pmt_t p) { (self->*m)(p); };
}
}  // namespace boost
static const boost::shim_placeholder1 _1 = boost::shim_placeholder1();

namespace gr {

class block {
public:
    block() : d_id(-1), d_consumed(0), d_relative_rate(1.0) {}          // for virtual inheritance
    block(const std::string& name, io_signature::sptr in, io_signature::sptr out)
        : d_name(name), d_in(in), d_out(out), d_id(next_id()++), d_consumed(0), d_relative_rate(1.0) {}
    virtual ~block() {}
    std::string name() const { return d_name; }
    long unique_id() const { return d_id; }
    virtual void forecast(int noutput_items, gr_vector_int& ninput_items_required) = 0;
    virtual int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items,
                             gr_vector_void_star& output_items) = 0;
    void consume_each(int n) { d_consumed = n; }
    void set_relative_rate(double r) { d_relative_rate = r; }
    double relative_rate() const { return d_relative_rate; }
    void message_port_register_in(pmt::pmt_t) {}
    template <class F> void set_msg_handler(pmt::pmt_t, F f) { d_handler = f; }
    // shim-only accessors for the oracle entry points
    int shim_consumed() const { return d_consumed; }
    void shim_post(pmt::pmt_t m) { if (d_handler) d_handler(m); }
private:
    static long& next_id() { static long id = 1000; return id; }
    std::string d_name;
    io_signature::sptr d_in, d_out;
    long d_id;
    int d_consumed;
    double d_relative_rate;
    std::function<void(pmt::pmt_t)> d_handler;
};

}  // namespace gr
#endif
