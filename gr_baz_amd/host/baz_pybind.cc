/* pybind11 module `_baz_music`: the in-container stand-in for the reference's SWIG stanza
 * (swig/baz_swig.i:560-574): `music_doa(m, n, nsamples, array_response, resolution)` returns a
 * handle on the C++ host block with `.set_array_response(list[list[complex]])`.  `.work(items)` is a
 * test/demo convenience that drives the block's virtual work() the way the GNU Radio scheduler
 * does (pointer vectors into numpy buffers). */
#include <baz_agc_cc.h>
#include <baz_fractional_resampler_cc.h>
#include <baz_music_doa.h>

#include <pybind11/complex.h>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <gnuradio/flowgraph_model.h>

namespace py = pybind11;

namespace {

struct music_doa_handle {
    baz_music_doa_sptr blk;
};

py::tuple drive_work(music_doa_handle& h, py::array_t<std::complex<float>, py::array::c_style | py::array::forcecast> items,
                     int n_outputs)
{
    if (n_outputs < 1 || n_outputs > 3) throw std::invalid_argument("n_outputs must be 1, 2 or 3");
    py::buffer_info bi = items.request();
    const size_t N = h.blk->nsamples();
    if (bi.size == 0 || (size_t)bi.size % N != 0) throw std::invalid_argument("items must hold k * nsamples complex64");
    const int nitems = (int)((size_t)bi.size / N);
    py::array_t<float> ang({(size_t)nitems, (size_t)h.blk->n()});
    py::array_t<float> lvl({(size_t)nitems, (size_t)h.blk->n()});
    py::array_t<float> spec({(size_t)nitems, (size_t)h.blk->resolution()});
    // work() is handed the address of the OLDEST look-back item (history() - 1 items in front of the first real one); the
    // block never reads them (it only declares them, to make the runtime size its input buffer), so no storage backs them here
    const void* window = reinterpret_cast<const void*>(reinterpret_cast<uintptr_t>(bi.ptr) -
                                                       (uintptr_t)(h.blk->history() - 1) * N * sizeof(gr_complex));
    gr_vector_const_void_star in(1, window);
    gr_vector_void_star out;
    out.push_back(ang.mutable_data());
    if (n_outputs > 1) out.push_back(lvl.mutable_data());
    if (n_outputs > 2) out.push_back(spec.mutable_data());
    int produced;
    {
        py::gil_scoped_release nogil;
        produced = h.blk->work(nitems, in, out);
    }
    py::object none = py::none();
    return py::make_tuple(produced, ang, n_outputs > 1 ? py::object(lvl) : none, n_outputs > 2 ? py::object(spec) : none);
}

py::dict stats_dict(const gr::shim::run_stats& st)
{
    py::dict d, sizes;
    for (std::map<long, long>::const_iterator it = st.call_sizes.begin(); it != st.call_sizes.end(); ++it)
        sizes[py::int_(it->first)] = it->second;
    d["calls"] = st.calls;
    d["items"] = st.items;
    d["dropped_at_end"] = st.dropped_at_end;
    d["in_bufsize"] = st.in_bufsize;
    d["out_bufsize"] = st.out_bufsize;
    d["call_sizes"] = sizes;
    d["work_seconds"] = st.work_seconds;
    d["total_seconds"] = st.total_seconds;
    d["steady_items"] = st.steady_items;
    d["steady_work_seconds"] = st.steady_work_seconds;
    d["last_return"] = st.last_return;
    return d;
}

// the block between a saturating source and draining sinks, driven as gnuradio-runtime 3.7 would (flowgraph_model.h):
// persistent doubly mapped buffers sized from the block's hints, one work() per executor iteration
py::tuple run_flowgraph(music_doa_handle& h, py::array_t<std::complex<float>, py::array::c_style | py::array::forcecast> items,
                        int n_outputs, bool collect, bool pin, int passes)
{
    if (n_outputs < 1 || n_outputs > 3) throw std::invalid_argument("n_outputs must be 1, 2 or 3");
    py::buffer_info bi = items.request();
    const size_t N = h.blk->nsamples();
    if (bi.size == 0 || (size_t)bi.size % N != 0) throw std::invalid_argument("items must hold k * nsamples complex64");
    const long nitems = (long)((size_t)bi.size / N);
    const size_t keep = collect ? (size_t)nitems : 0;
    py::array_t<float> ang({keep, (size_t)h.blk->n()});
    py::array_t<float> lvl({keep, (size_t)h.blk->n()});
    py::array_t<float> spec({keep, (size_t)h.blk->resolution()});
    char* sinks[3] = {NULL, NULL, NULL};
    if (collect) {
        sinks[0] = (char*)ang.mutable_data();
        if (n_outputs > 1) sinks[1] = (char*)lvl.mutable_data();
        if (n_outputs > 2) sinks[2] = (char*)spec.mutable_data();
    }
    gr::shim::run_stats st;
    unsigned long long pinned = 0;
    {
        py::gil_scoped_release nogil;
        const bool before = h.blk->pin_buffers();
        h.blk->set_pin_buffers(pin);
        struct restore {
            baz_music_doa& b;
            bool to;
            ~restore()
            {
                b.set_pin_buffers(false);          // releases whatever a failed run left registered (the buffers are gone)
                if (to) b.set_pin_buffers(true);
            }
        } r = {*h.blk, before};
        st = gr::shim::run_sync_block(*h.blk, (const char*)bi.ptr, nitems, n_outputs, sinks, &pinned, passes);
    }
    py::dict d = stats_dict(st);
    d["pinned_bytes_at_stop"] = pinned;
    py::object none = py::none();
    return py::make_tuple(d, collect ? py::object(ang) : none, collect && n_outputs > 1 ? py::object(lvl) : none,
                          collect && n_outputs > 2 ? py::object(spec) : none);
}

// a device-free sync block for the model's self-test: copies its input to every output
class shim_copy_block : public gr::sync_block
{
public:
    shim_copy_block(int item, int nout, int multiple, long min_buffer, int cap, int history)
        : gr::sync_block("shim_copy", gr::io_signature::make(1, 1, item), gr::io_signature::make(1, 3, item)), d_item(item), d_nout(nout)
    {
        set_history((unsigned)history);          // look-back it never reads, like the MUSIC / AGC blocks' buffer request
        set_output_multiple(multiple);
        if (min_buffer > 0) set_min_output_buffer(min_buffer);
        if (cap > 0) set_max_noutput_items(cap);
    }
    int work(int noutput_items, gr_vector_const_void_star& in, gr_vector_void_star& out)
    {
        const char* newest = static_cast<const char*>(in[0]) + (size_t)(history() - 1) * (size_t)d_item;
        for (int p = 0; p < d_nout; ++p) std::memcpy(out[p], newest, (size_t)noutput_items * (size_t)d_item);
        return noutput_items;
    }
    unsigned long long pinned_bytes() const { return 0; }

private:
    int d_item, d_nout;
};

py::tuple model_selftest(py::array_t<unsigned char, py::array::c_style | py::array::forcecast> data, int item_size, int n_outputs,
                         int multiple, long min_buffer, int cap, int history)
{
    py::buffer_info bi = data.request();
    if (item_size <= 0 || bi.size % item_size) throw std::invalid_argument("data must hold k * item_size bytes");
    const long nitems = (long)(bi.size / item_size);
    shim_copy_block blk(item_size, n_outputs, multiple, min_buffer, cap, history < 1 ? 1 : history);
    std::vector<py::array_t<unsigned char> > outs;
    char* sinks[3] = {NULL, NULL, NULL};
    for (int p = 0; p < n_outputs; ++p) {
        outs.push_back(py::array_t<unsigned char>((size_t)bi.size));
        sinks[p] = (char*)outs[p].mutable_data();
    }
    unsigned long long pinned = 0;
    const gr::shim::run_stats st = gr::shim::run_sync_block(blk, (const char*)bi.ptr, nitems, n_outputs, sinks, &pinned);
    py::list l;
    for (int p = 0; p < n_outputs; ++p) l.append(outs[p]);
    return py::make_tuple(stats_dict(st), l);
}

struct agc_handle {
    baz_agc_cc_sptr blk;
};

py::tuple drive_agc(agc_handle& h, py::array_t<std::complex<float>, py::array::c_style | py::array::forcecast> x, int n_outputs)
{
    if (n_outputs < 1 || n_outputs > 3) throw std::invalid_argument("n_outputs must be 1, 2 or 3");
    py::buffer_info bi = x.request();
    const int n = (int)bi.size;
    py::array_t<std::complex<float>> out((size_t)n);
    py::array_t<float> env((size_t)n), mul((size_t)n);
    const void* window = reinterpret_cast<const void*>(reinterpret_cast<uintptr_t>(bi.ptr) -
                                                       (uintptr_t)(h.blk->history() - 1) * sizeof(gr_complex));   // see drive_work
    gr_vector_const_void_star in(1, window);
    gr_vector_void_star outs;
    outs.push_back(out.mutable_data());
    if (n_outputs > 1) outs.push_back(env.mutable_data());
    if (n_outputs > 2) outs.push_back(mul.mutable_data());
    int produced;
    {
        py::gil_scoped_release nogil;
        produced = h.blk->work(n, in, outs);
    }
    py::object none = py::none();
    return py::make_tuple(produced, out, n_outputs > 1 ? py::object(env) : none, n_outputs > 2 ? py::object(mul) : none);
}

struct resamp_handle {
    gr::baz::fractional_resampler_cc::sptr blk;
};

// one general_work() call the way the scheduler issues it: forecast()-sized input window, consume_each() reported
// the same with the second input wired: rr = the per-sample ratio stream (.cc:205-217)
py::tuple drive_resamp2(resamp_handle& h, py::array_t<std::complex<float>, py::array::c_style | py::array::forcecast> x,
                        py::array_t<float, py::array::c_style | py::array::forcecast> rr, int noutput)
{
    py::buffer_info bi = x.request(), br = rr.request();
    gr_vector_int nin;
    nin.push_back((int)bi.size);
    nin.push_back((int)br.size);
    py::array_t<std::complex<float>> out((size_t)(noutput > 0 ? noutput : 0));
    gr_vector_const_void_star in;
    in.push_back(bi.ptr);
    in.push_back(br.ptr);
    gr_vector_void_star outs(1, out.mutable_data());
    int produced;
    {
        py::gil_scoped_release nogil;
        produced = h.blk->general_work(noutput, nin, in, outs);
    }
    return py::make_tuple(produced, out, h.blk->last_consumed());
}

py::tuple drive_resamp(resamp_handle& h, py::array_t<std::complex<float>, py::array::c_style | py::array::forcecast> x, int noutput)
{
    py::buffer_info bi = x.request();
    gr_vector_int nin(1, (int)bi.size);
    gr_vector_int need(1, 0);
    h.blk->forecast(noutput, need);
    if ((int)bi.size < need[0]) throw std::invalid_argument("general_work needs forecast(noutput) input items");
    py::array_t<std::complex<float>> out((size_t)(noutput > 0 ? noutput : 0));
    gr_vector_const_void_star in(1, bi.ptr);
    gr_vector_void_star outs(1, out.mutable_data());
    int produced;
    {
        py::gil_scoped_release nogil;
        produced = h.blk->general_work(noutput, nin, in, outs);
    }
    return py::make_tuple(produced, out, h.blk->last_consumed());
}

}  // namespace

PYBIND11_MODULE(_baz_music, mod)
{
    mod.doc() = "baz.music_doa on the MI355X host block (stand-in for swig/baz_swig.i:560-574)";
    py::class_<music_doa_handle>(mod, "baz_music_doa_sptr")
        .def("set_array_response",
             [](music_doa_handle& h, const array_response_t& t) { h.blk->set_array_response(t); },
             py::arg("array_response"))
        .def("set_peak_mode", [](music_doa_handle& h, bool on) { h.blk->set_peak_mode(on); }, py::arg("local_maxima"))
        .def("array_response", [](music_doa_handle& h) { return h.blk->array_response(); })
        .def("name", [](music_doa_handle& h) { return h.blk->name(); })
        .def("unique_id", [](music_doa_handle& h) { return h.blk->unique_id(); })
        .def("m", [](music_doa_handle& h) { return h.blk->m(); })
        .def("n", [](music_doa_handle& h) { return h.blk->n(); })
        .def("nsamples", [](music_doa_handle& h) { return h.blk->nsamples(); })
        .def("resolution", [](music_doa_handle& h) { return h.blk->resolution(); })
        .def("input_item_sizes", [](music_doa_handle& h) { return h.blk->input_signature()->sizeof_stream_items(); })
        .def("output_item_sizes", [](music_doa_handle& h) { return h.blk->output_signature()->sizeof_stream_items(); })
        .def("output_streams", [](music_doa_handle& h) {
            return py::make_tuple(h.blk->output_signature()->min_streams(), h.blk->output_signature()->max_streams());
        })
        /* scheduler hints the block registered (recorded by the API stand-in; a real runtime acts on them) */
        .def("device", [](music_doa_handle& h) { return h.blk->device(); })
        .def("output_multiple", [](music_doa_handle& h) { return h.blk->output_multiple(); })
        .def("history", [](music_doa_handle& h) { return h.blk->history(); })
        .def("min_output_buffer", [](music_doa_handle& h) { return h.blk->min_output_buffer(); })
        .def("max_noutput_items", [](music_doa_handle& h) { return h.blk->max_noutput_items(); })
        .def("work", &drive_work, py::arg("items"), py::arg("n_outputs") = 3)
        /* page-locking of caller buffers: off in this stand-in (work() above is handed numpy temporaries), on for the
         * persistent buffers of run_flowgraph() when asked */
        .def("set_pin_buffers", [](music_doa_handle& h, bool on) { h.blk->set_pin_buffers(on); }, py::arg("on"))
        .def("pin_buffers", [](music_doa_handle& h) { return h.blk->pin_buffers(); })
        .def("pinned_bytes", [](music_doa_handle& h) { return h.blk->pinned_bytes(); })
        .def("run_flowgraph", &run_flowgraph, py::arg("items"), py::arg("n_outputs") = 3, py::arg("collect") = true,
             py::arg("pin") = false, py::arg("passes") = 1);
    py::class_<agc_handle>(mod, "baz_agc_cc_sptr")
        .def("name", [](agc_handle& h) { return h.blk->name(); })
        .def("output_multiple", [](agc_handle& h) { return h.blk->output_multiple(); })
        .def("history", [](agc_handle& h) { return h.blk->history(); })
        .def("min_output_buffer", [](agc_handle& h) { return h.blk->min_output_buffer(); })
        .def("input_item_sizes", [](agc_handle& h) { return h.blk->input_signature()->sizeof_stream_items(); })
        .def("output_item_sizes", [](agc_handle& h) { return h.blk->output_signature()->sizeof_stream_items(); })
        .def("output_streams", [](agc_handle& h) {
            return py::make_tuple(h.blk->output_signature()->min_streams(), h.blk->output_signature()->max_streams());
        })
        .def("work", &drive_agc, py::arg("items"), py::arg("n_outputs") = 3);
    py::class_<resamp_handle>(mod, "fractional_resampler_cc_sptr")
        .def("name", [](resamp_handle& h) { return h.blk->name(); })
        .def("input_item_sizes", [](resamp_handle& h) { return h.blk->input_signature()->sizeof_stream_items(); })
        .def("output_item_sizes", [](resamp_handle& h) { return h.blk->output_signature()->sizeof_stream_items(); })
        .def("relative_rate", [](resamp_handle& h) { return h.blk->relative_rate(); })
        .def("forecast", [](resamp_handle& h, int n) { gr_vector_int r(1, 0); h.blk->forecast(n, r); return r[0]; })
        .def("mu", [](resamp_handle& h) { return (double)h.blk->mu(); })
        .def("resamp_ratio", [](resamp_handle& h) { return (double)h.blk->resamp_ratio(); })
        .def("set_mu", [](resamp_handle& h, double mu) { h.blk->set_mu((long double)mu); })
        .def("set_resamp_ratio", [](resamp_handle& h, double r) { h.blk->set_resamp_ratio(r); })
        .def("set_resamp_ratio", [](resamp_handle& h, unsigned long long n, unsigned long long d) { h.blk->set_resamp_ratio(n, d); })
        .def("handle_ppb", [](resamp_handle& h, long w, double f) { h.blk->handle_ppb(w, f); })
        .def("handle_adjust", [](resamp_handle& h, double d) { h.blk->handle_adjust(d); })
        .def("general_work", &drive_resamp, py::arg("items"), py::arg("noutput_items"))
        .def("general_work2", &drive_resamp2, py::arg("items"), py::arg("ratio"), py::arg("noutput_items"))
        .def("input_streams", [](resamp_handle& h) {
            return py::make_tuple(h.blk->input_signature()->min_streams(), h.blk->input_signature()->max_streams());
        })
        .def("has_msg_port", [](resamp_handle& h, const std::string& p) { return h.blk->has_msg_port(p); })
        /* what a flowgraph's message source would deliver to the "msg" port: (whole . frac) ppb pair, or a double */
        .def("post_msg_ppb", [](resamp_handle& h, long whole, double frac) {
            h.blk->shim_post(pmt::mp("msg"), pmt::cons(pmt::from_long(whole), pmt::from_double(frac)));
        })
        .def("post_msg_double", [](resamp_handle& h, double d) { h.blk->shim_post(pmt::mp("msg"), pmt::from_double(d)); })
        .def("post_msg_symbol", [](resamp_handle& h, const std::string& s) { h.blk->shim_post(pmt::mp("msg"), pmt::mp(s)); });
    // swig/baz_swig.i:964-966: GR_SWIG_BLOCK_MAGIC2(baz, fractional_resampler_cc) -> baz.fractional_resampler_cc(...)
    mod.def("recover_mmse_taps", []() {
                const std::vector<float> t = gr::baz::recover_mmse_taps();
                py::array_t<float> a({(size_t)(t.size() / 8), (size_t)8});
                std::memcpy(a.mutable_data(), t.data(), t.size() * sizeof(float));
                return a;
            }, "the tap table of the gnuradio-filter this build sees, read out through mmse_fir_interpolator_cc::interpolate");
    mod.def("fractional_resampler_cc",
            [](double phase_shift, double resamp_ratio, unsigned long long num, unsigned long long denom) {
                resamp_handle h;
                h.blk = gr::baz::fractional_resampler_cc::make(phase_shift, resamp_ratio, num, denom);
                return h;
            },
            py::arg("phase_shift"), py::arg("resamp_ratio"), py::arg("resamp_ratio_num") = 0ull, py::arg("resamp_ratio_denom") = 0ull);
    // swig/baz_swig.i: GR_SWIG_BLOCK_MAGIC(baz, agc_cc) -> baz.agc_cc(rate, reference, gain, max_gain)
    mod.def("agc_cc",
            [](float rate, float reference, float gain, float max_gain) {
                agc_handle h;
                h.blk = baz_make_agc_cc(rate, reference, gain, max_gain);
                return h;
            },
            py::arg("rate") = 1e-4f, py::arg("reference") = 1.0f, py::arg("gain") = 1.0f, py::arg("max_gain") = 0.0f);
    /* gnuradio-runtime 3.7's buffer sizing and call planning as restated in gr_shim/gnuradio/flowgraph_model.h */
    mod.def("gr37_buffer_items",
            [](long item_size, int output_multiple, long min_output_buffer, long max_output_buffer,
               const std::vector<std::tuple<double, int, int> >& downstream, long page_size) {
                std::vector<gr::shim::downstream_t> r;
                for (size_t i = 0; i < downstream.size(); ++i)
                    r.push_back(gr::shim::downstream_t{std::get<0>(downstream[i]), std::get<1>(downstream[i]), std::get<2>(downstream[i])});
                return gr::shim::buffer_items(item_size, output_multiple, min_output_buffer, max_output_buffer, r, page_size);
            },
            py::arg("item_size"), py::arg("output_multiple") = 1, py::arg("min_output_buffer") = -1L,
            py::arg("max_output_buffer") = -1L, py::arg("downstream") = std::vector<std::tuple<double, int, int> >(),
            py::arg("page_size") = 4096L);
    mod.def("gr37_plan_noutput",
            [](long items_in, const std::vector<long>& out_space, const std::vector<long>& out_bufsize, int multiple, int history,
               int max_noutput_items) {
                return gr::shim::plan_noutput(items_in, out_space, out_bufsize, multiple, history, max_noutput_items > 0,
                                              max_noutput_items);
            },
            py::arg("items_in"), py::arg("out_space"), py::arg("out_bufsize"), py::arg("multiple") = 1, py::arg("history") = 1,
            py::arg("max_noutput_items") = 0);
    mod.def("gr37_model_selftest", &model_selftest, py::arg("data"), py::arg("item_size"), py::arg("n_outputs") = 1,
            py::arg("multiple") = 1, py::arg("min_buffer") = -1L, py::arg("cap") = 0, py::arg("history") = 1);
    mod.def("deal_device", &baz_music_doa_deal_device, py::arg("instance"), py::arg("device_count"),
            "placement rule of block instances: instance % device_count (-1 without devices)");
    mod.def("music_doa",
            [](unsigned int m, unsigned int n, unsigned int nsamples, const array_response_t& table, unsigned int resolution) {
                music_doa_handle h;
                h.blk = baz_make_music_doa(m, n, nsamples, table, resolution);
                h.blk->set_pin_buffers(false);   // work() is handed temporaries here; run_flowgraph(pin=True) opts in
                return h;
            },
            py::arg("m"), py::arg("n"), py::arg("nsamples"), py::arg("array_response"), py::arg("resolution"));
}
