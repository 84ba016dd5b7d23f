#!/usr/bin/env python
"""Runs ON A GNU RADIO HOST with an MI355X (written for python 2.7 / GNU Radio 3.7, runs under python 3 / 3.8+ too); it
cannot run in this repository's build image (no GNU Radio).  Started by scripts/validate_on_gr37.sh after the SWIG module is
built; may also be run by hand:  python validate_flowgraph.py <module dir> <repo root>

For every golden fixture tests/golden/cfg{1,2}*.npz (outputs of the reference's own lib/baz_music_doa.cc):
    vector_source_c(items, vlen = nsamples) -> baz.music_doa(m, n, nsamples, table, resolution) -> 3 x vector_sink_f
under the REAL scheduler (flat_flowgraph buffer allocation, thread-per-block executor), compared with the fixture:
    ang exactly (bin -> degrees), lvl and spectrum within 1e-5 relative (north_star's tolerance).
It also prints what the real runtime did with the block's scheduling requests:
GNU Radio's own performance counters of the block (average / last noutput_items per work() call, work time; enabled here
through GR_CONF_PERFCOUNTERS_ON, present when the runtime was built with ENABLE_PERFORMANCE_COUNTERS) -- the number to look
at is the average call size on the x40 runs: the look-back request (history 1,025 => 2 (H + 2) input items) should make it
hundreds of items, not 1 --, and that a finite stream of 1, 7 and 2,500 items arrives complete (the look-back trick must
not lose or delay items)."""
from __future__ import print_function

import glob
import os
import sys

os.environ.setdefault("GR_CONF_PERFCOUNTERS_ON", "True")      # before gnuradio is imported

import numpy as np


def main():
    moddir, root = sys.argv[1], sys.argv[2]
    sys.path.insert(0, moddir)
    from gnuradio import blocks, gr
    import baz_music_swig as baz                       # the module swig/baz_music.i builds (gr-baz: `from baz import music_doa`)

    failures = 0
    fixtures = sorted(glob.glob(os.path.join(root, "tests", "golden", "cfg1*.npz")) +
                      glob.glob(os.path.join(root, "tests", "golden", "cfg2*.npz")))
    for path in fixtures:
        g = np.load(path, allow_pickle=False)
        m, n, N, res = int(g["m"]), int(g["n"]), int(g["nsamples"]), int(g["res"])
        items = np.ascontiguousarray(g["items"])
        table = [[complex(v) for v in row] for row in g["table"]]      # array_response_t: [bin][antenna]
        for reps in (1, 40):                                           # the fixture once; and often enough to fill big calls
            x = np.tile(items, (reps, 1))
            tb = gr.top_block()
            src = blocks.vector_source_c(x.reshape(-1).tolist(), False, N)
            doa = baz.music_doa(m, n, N, table, res)
            s_ang, s_lvl, s_spec = blocks.vector_sink_f(n), blocks.vector_sink_f(n), blocks.vector_sink_f(res)
            tb.connect(src, doa)
            tb.connect((doa, 0), s_ang)
            tb.connect((doa, 1), s_lvl)
            tb.connect((doa, 2), s_spec)
            tb.run()
            ang = np.array(s_ang.data(), np.float32).reshape(-1, n)
            lvl = np.array(s_lvl.data(), np.float32).reshape(-1, n)
            spec = np.array(s_spec.data(), np.float32).reshape(-1, res)
            ok = ang.shape[0] == x.shape[0]                             # every item of the finite stream arrived
            if ok:
                want_ang, want_lvl, want_spec = (np.tile(g[k], (reps, 1)) for k in ("ang", "lvl", "spectrum"))
                ok = (np.array_equal(ang, want_ang) and np.allclose(lvl, want_lvl, rtol=1e-5, atol=0)
                      and np.allclose(spec, want_spec, rtol=1e-5, atol=0))
            worst = float(np.max(np.abs(spec / np.tile(g["spectrum"], (reps, 1)) - 1.0))) if ang.shape[0] == x.shape[0] else float("nan")
            print("%-34s x%-3d items %6d -> %6d   worst spectrum error %.2e   %s" % (
                os.path.basename(path), reps, x.shape[0], ang.shape[0], worst, "ok" if ok else "MISMATCH"))
            for name in ("pc_noutput_items_avg", "pc_noutput_items", "pc_work_time_avg", "pc_work_time_total"):
                try:                                                    # gr::block's performance counters, where compiled in
                    print("      %s = %s" % (name, getattr(doa, name)()))
                except Exception:
                    pass
            failures += 0 if ok else 1
    # finite streams: nothing lost, nothing delayed (look-back instead of an output multiple)
    g = np.load(fixtures[0], allow_pickle=False)
    m, n, N, res = int(g["m"]), int(g["n"]), int(g["nsamples"]), int(g["res"])
    table = [[complex(v) for v in row] for row in g["table"]]
    for count in (1, 7, 2500):
        x = np.tile(g["items"], ((count + g["items"].shape[0] - 1) // g["items"].shape[0], 1))[:count]
        tb = gr.top_block()
        src = blocks.vector_source_c(x.reshape(-1).tolist(), False, N)
        doa = baz.music_doa(m, n, N, table, res)
        s_ang, s_lvl = blocks.vector_sink_f(n), blocks.vector_sink_f(n)
        tb.connect(src, doa)
        tb.connect((doa, 0), s_ang)
        tb.connect((doa, 1), s_lvl)
        tb.run()
        got = len(s_ang.data()) // n
        want = np.tile(g["ang"], ((count + g["ang"].shape[0] - 1) // g["ang"].shape[0], 1))[:count]
        ok = got == count and np.array_equal(np.array(s_ang.data(), np.float32).reshape(-1, n), want)
        print("finite stream of %5d items -> %5d   %s" % (count, got, "ok" if ok else "LOST / DELAYED ITEMS"))
        failures += 0 if ok else 1
    print("GNU Radio %s: %s" % (gr.version(), "ALL OK" if failures == 0 else "%d FAILURE(S)" % failures))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
