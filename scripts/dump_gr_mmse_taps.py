#!/usr/bin/env python
"""Run on a host that HAS GNU Radio (3.7 or later) -- it cannot run in this repo's build image.

Writes tests/golden/mmse_taps_gr37.npz: the 129 x 8 float32 tap table of gnuradio-filter's MMSE interpolator
(gr::filter::mmse_fir_interpolator_cc, the arithmetic behind /root/reference/lib/baz_fractional_resampler_cc.cc:87,172,203),
read out of the installed library rather than copied from its header: for every phase imu = 0 .. 128 and every tap k,
interpolate() of a unit impulse at position k with mu = imu / 128 returns taps[imu][7 - k] exactly (seven zero products
and one product by 1.0f in a float accumulation).  Once that file is committed, tests/test_resamp.py requires the
engine's DEFAULT table (closed form, six digits) to equal it bit for bit -- the pin this path is missing offline.

Two read-out routes, tried in order:
  1. gnuradio.filter.mmse_fir_interpolator_cc, where the python bindings expose it (GNU Radio >= 3.8 does);
  2. a flowgraph: vector_source_c(impulse train) -> filter.fractional_resampler_cc(mu, 1.0) -> vector_sink_c, which works on
     3.7 (the stock block is the same interpolator driven with a fixed phase: output n = interpolate(&in[n], mu)).
usage: python scripts/dump_gr_mmse_taps.py [out.npz]"""
from __future__ import print_function

import os
import sys

import numpy as np

NSTEPS, NTAPS = 128, 8


def via_interpolator_object():
    from gnuradio import filter as grfilter
    interp = grfilter.mmse_fir_interpolator_cc()
    assert interp.ntaps() == NTAPS and interp.nsteps() == NSTEPS
    taps = np.zeros((NSTEPS + 1, NTAPS), dtype=np.float32)
    for imu in range(NSTEPS + 1):
        for k in range(NTAPS):
            impulse = np.zeros(NTAPS, dtype=np.complex64)
            impulse[k] = 1.0
            taps[imu, NTAPS - 1 - k] = np.float32(interp.interpolate(impulse.tolist(), float(imu) / NSTEPS).real)
    return taps, "gnuradio.filter.mmse_fir_interpolator_cc.interpolate"


def via_flowgraph():
    from gnuradio import blocks, gr
    from gnuradio import filter as grfilter
    taps = np.zeros((NSTEPS + 1, NTAPS), dtype=np.float32)
    gap = 32                                             # impulses far enough apart that their responses do not overlap
    for imu in range(NSTEPS + 1):
        mu = float(imu) / NSTEPS
        if imu == NSTEPS:
            mu = np.nextafter(np.float32(1.0), np.float32(0.0)).item()     # the block keeps mu in [0, 1): rint(mu * 128) is still 128
        x = np.zeros(gap * 4, dtype=np.complex64)
        x[gap] = 1.0
        tb = gr.top_block()
        src = blocks.vector_source_c(x.tolist(), False)
        rs = grfilter.fractional_resampler_cc(mu, 1.0)
        snk = blocks.vector_sink_c()
        tb.connect(src, rs, snk)
        tb.run()
        y = np.array(snk.data(), dtype=np.complex64)
        # output n = sum_k in[n + k] taps[imu][7 - k]: the impulse at `gap` shows tap 7 - k at output gap - k
        for k in range(NTAPS):
            taps[imu, NTAPS - 1 - k] = np.float32(y[gap - k].real)
    return taps, "flowgraph vector_source_c -> filter.fractional_resampler_cc(mu, 1.0) -> vector_sink_c"


def main():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "tests", "golden", "mmse_taps_gr37.npz")
    from gnuradio import gr
    try:
        taps, how = via_interpolator_object()
    except Exception as e:                               # 3.7's python does not wrap the interpolator class
        print("interpolator object not available (%r): reading the table through a flowgraph" % (e,))
        taps, how = via_flowgraph()
    assert taps.shape == (NSTEPS + 1, NTAPS)
    assert np.array_equal(taps[0], [0, 0, 0, 0, 1, 0, 0, 0]) and np.array_equal(taps[NSTEPS], [0, 0, 0, 1, 0, 0, 0, 0]), \
        "rows 0 / 128 are pure delays in every gnuradio-filter: the read-out is misaligned"
    assert np.array_equal(taps[::-1, ::-1], taps), "taps(1 - mu) = reversed taps(mu) does not hold: misaligned read-out"
    np.savez(out, taps=taps, gnuradio_version=np.array(gr.version()), read_out=np.array(how))
    print("wrote %s (GNU Radio %s, %s); row 1 = %s" % (out, gr.version(), how, " ".join("%.5e" % v for v in taps[1])))
    print("now: python -m pytest tests/test_resamp.py -k gnuradio_filters -q   (and commit the file)")


if __name__ == "__main__":
    main()
