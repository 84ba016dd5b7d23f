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
    gr_vector_const_void_star in(1, bi.ptr);
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
    gr_vector_const_void_star in(1, bi.ptr);
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
        .def("min_output_buffer", [](music_doa_handle& h) { return h.blk->min_output_buffer(); })
        .def("max_noutput_items", [](music_doa_handle& h) { return h.blk->max_noutput_items(); })
        .def("work", &drive_work, py::arg("items"), py::arg("n_outputs") = 3);
    py::class_<agc_handle>(mod, "baz_agc_cc_sptr")
        .def("name", [](agc_handle& h) { return h.blk->name(); })
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
    mod.def("deal_device", &baz_music_doa_deal_device, py::arg("instance"), py::arg("device_count"),
            "placement rule of block instances: instance % device_count (-1 without devices)");
    mod.def("music_doa",
            [](unsigned int m, unsigned int n, unsigned int nsamples, const array_response_t& table, unsigned int resolution) {
                music_doa_handle h;
                h.blk = baz_make_music_doa(m, n, nsamples, table, resolution);
                return h;
            },
            py::arg("m"), py::arg("n"), py::arg("nsamples"), py::arg("array_response"), py::arg("resolution"));
}
