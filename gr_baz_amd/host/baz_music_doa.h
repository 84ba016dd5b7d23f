/* -*- c++ -*- */
/* MUSIC direction-of-arrival estimator block, MI355X (gfx950) implementation.
 *
 * Drop-in for gr-baz's block of the same name: the public surface below is the one declared in
 * /root/reference/lib/baz_music_doa.h:29-60 -- sptr typedef (:30), antenna_response_t /
 * array_response_t / doa_t (:32-34), the factory (:36), a gr::sync_block subclass with a private
 * constructor and friend factory (:38-43), work() (:48) and set_array_response() (:59) -- so the SWIG
 * stanza (swig/baz_swig.i:560-574), music_doa_helper.py and existing flowgraphs keep working.
 * What differs is private: instead of Armadillo temporaries the block owns one baz_music_ctx
 * (include/baz_music_hip.h) and marshals the scheduler's buffers across that C-ABI into the HIP
 * kernels.  There is no CPU arithmetic in this block.
 */
#ifndef INCLUDED_BAZ_MUSIC_DOA_H
#define INCLUDED_BAZ_MUSIC_DOA_H

#include <gnuradio/sync_block.h>
#include <gnuradio/thread/thread.h>

#include <utility>
#include <vector>

struct baz_music_ctx;   /* include/baz_music_hip.h */

class baz_music_doa;
typedef boost::shared_ptr<baz_music_doa> baz_music_doa_sptr;

typedef std::vector<gr_complex> antenna_response_t;      /* one steering vector, length m */
typedef std::vector<antenna_response_t> array_response_t; /* resolution of those */
typedef std::pair<double, double> doa_t;                  /* (angle in degrees, strength) */

/* m antennas, n expected emitters (0 < n < m), nsamples complex samples per item (all antennas
 * interleaved, nsamples % m == 0), array_response = resolution x m steering table.
 * Throws std::invalid_argument on a bad configuration and std::runtime_error when no gfx950 device
 * can be opened (the reference only assert()s, lib/baz_music_doa.cc:45-50). */
baz_music_doa_sptr baz_make_music_doa(unsigned int m, unsigned int n, unsigned int nsamples,
                                      const array_response_t& array_response, unsigned int resolution);

class baz_music_doa : public gr::sync_block
{
private:
    friend baz_music_doa_sptr baz_make_music_doa(unsigned int m, unsigned int n, unsigned int nsamples,
                                                 const array_response_t& array_response,
                                                 unsigned int resolution);

    baz_music_doa(unsigned int m, unsigned int n, unsigned int nsamples,
                  const array_response_t& array_response, unsigned int resolution);

public:
    ~baz_music_doa();

    /* in:  input_items[0]  = noutput_items items of nsamples gr_complex
     * out: output_items[0] = ang (n floats/item), [1] = lvl (n floats/item, optional),
     *      [2] = spectrum (resolution floats/item, optional).
     * Processes ALL noutput_items and returns that count (the reference handles one item per call,
     * lib/baz_music_doa.cc:160; the produced streams are identical).  -1 on a fatal device error. */
    int work(int noutput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items);

    /* Thread-safe table replacement (python: music_doa_helper.set_frequency). */
    void set_array_response(const array_response_t& array_response);
    /* Extension, off by default (not in the reference): emit the n strongest local maxima of the pseudo-spectrum
     * instead of the n strongest bins (baz_music_set_peak_mode, include/baz_music_hip.h). */
    void set_peak_mode(bool local_maxima);

    /* Page-locking of the scheduler's stream buffers (baz_music_set_host_pinning, include/baz_music_hip.h): work()
     * registers the ranges it is handed the first time it sees them, stop() and the destructor release them.  On by
     * default (BAZ_MUSIC_PIN_BUFFERS=0 turns it off): GNU Radio's buffers outlive every work() call.  A caller that
     * hands work() temporaries (the pybind stand-in does) must switch it off. */
    void set_pin_buffers(bool on);
    bool pin_buffers() const { return d_pin_buffers; }
    unsigned long long pinned_bytes() const;
    bool start();   /* gr::block::start */
    bool stop();    /* gr::block::stop: releases the page locks while the buffers still exist */

    unsigned int m() const { return d_m; }
    unsigned int n() const { return d_n; }
    unsigned int nsamples() const { return d_nsamples; }
    unsigned int resolution() const { return d_resolution; }
    array_response_t array_response();
    /* HIP device this instance's context lives on (see baz_music_doa_deal_device). */
    int device() const;

private:
    unsigned int d_m;
    unsigned int d_n;
    unsigned int d_nsamples;
    unsigned int d_resolution;
    array_response_t d_array_response;   /* host copy, guarded by d_mutex */
    gr::thread::mutex d_mutex;
    baz_music_ctx* d_ctx;                /* device-side state (table, workspace, stream) */
    bool d_pin_buffers;
};

/* Placement rule of independent block instances (BASELINE config 4: 64 streams in one flowgraph; SURVEY.md 8e,
 * stream s -> GPU s mod G): the instance-th block made by this process goes to device instance % device_count;
 * -1 (= the current HIP device) when no gfx950 device is visible.  BAZ_MUSIC_DEVICE pins every instance instead. */
int baz_music_doa_deal_device(unsigned int instance, int device_count);

#endif /* INCLUDED_BAZ_MUSIC_DOA_H */
