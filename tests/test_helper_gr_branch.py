"""The gr.hier_block2 branch of music_doa_helper (python/music_doa_helper.py:49-103), executed against a
recording stand-in for `gnuradio.gr`: port signatures, the wiring of :91-96, the retune of :100-103.  CPU only -- the
wrapped factory is replaced by a recorder, no device is touched."""
import importlib.util
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Sig(object):
    def __init__(self, lo, hi, *sizes):
        self.lo, self.hi, self.sizes = lo, hi, list(sizes)


def _fake_gr():
    gr = types.ModuleType("gnuradio.gr")
    gr.sizeof_float, gr.sizeof_gr_complex = 4, 8
    gr.io_signature = lambda lo, hi, a: _Sig(lo, hi, a)
    gr.io_signature2 = lambda lo, hi, a, b: _Sig(lo, hi, a, b)
    gr.io_signature3 = lambda lo, hi, a, b, c: _Sig(lo, hi, a, b, c)

    class hier_block2(object):
        def __init__(self, name, insig, outsig):
            self.block_name, self.insig, self.outsig, self.edges = name, insig, outsig, []

        def connect(self, *points):
            ends = [p if isinstance(p, tuple) else (p, 0) for p in points]
            for a, b in zip(ends[:-1], ends[1:]):
                self.edges.append((a, b))

    gr.hier_block2 = hier_block2
    pkg = types.ModuleType("gnuradio")
    pkg.gr = gr
    return pkg, gr


class _Impl(object):
    def __init__(self, *args):
        self.args, self.tables = args, []

    def set_array_response(self, table):
        self.tables.append(table)


@pytest.fixture
def helper_gr(monkeypatch):
    import gr_baz_amd.baz as baz
    pkg, gr = _fake_gr()
    monkeypatch.setitem(sys.modules, "gnuradio", pkg)
    monkeypatch.setitem(sys.modules, "gnuradio.gr", gr)
    monkeypatch.setattr(baz, "music_doa", _Impl)
    spec = importlib.util.spec_from_file_location("gr_baz_amd.baz._music_doa_helper_under_gr",
                                                  os.path.join(ROOT, "gr_baz_amd", "baz", "music_doa_helper.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod._HAVE_GR and issubclass(mod.music_doa_helper, gr.hier_block2)
    return mod


@pytest.mark.parametrize("spectrum", [False, True])
def test_hier_block_signatures_and_wiring(helper_gr, spectrum, capsys):
    m, n, N, res = 4, 2, 1024, 3600
    h = helper_gr.music_doa_helper(m, n, N, res, 299792458.0, 0.5, [[0, 0], [1, 0], [1, 1], [0, 1]], output_spectrum=spectrum)
    assert h.block_name == "music_doa_helper"
    assert (h.insig.lo, h.insig.hi, h.insig.sizes) == (1, 1, [8 * N])                       # :68
    if spectrum:
        assert (h.outsig.lo, h.outsig.hi, h.outsig.sizes) == (3, 3, [4 * n, 4 * n, 4 * res])  # :61-62
    else:
        assert (h.outsig.lo, h.outsig.hi, h.outsig.sizes) == (2, 2, [4 * n, 4 * n])           # :63-64
    want = [((h, 0), (h.impl, 0)), ((h.impl, 0), (h, 0)), ((h.impl, 1), (h, 1))]
    if spectrum:
        want.append(((h.impl, 2), (h, 2)))
    assert h.edges == want                                                                    # :91-96
    assert h.impl.args[:3] == (m, n, N) and h.impl.args[4] == res
    assert h.l == 1.0 and h.antenna_array == [[0, 0], [0.5, 0], [0.5, 0.5], [0, 0.5]]
    tab = np.array(h.impl.args[3])
    assert tab.shape == (res, m)
    want_tab = np.exp(-2j * np.pi * (np.array(h.antenna_array) @ np.array([np.cos(np.radians(40.3)), np.sin(np.radians(40.3))])))
    assert np.allclose(tab[403], want_tab, atol=1e-12)
    assert "MUSIC DOA Helper: M: 4, N: 2, # samples: 1024" in capsys.readouterr().out


def test_hier_block_retune_and_nsamples_check(helper_gr):
    h = helper_gr.music_doa_helper(4, 1, 512, 360, 1.0e9, 0.1, [[0, 0], [1, 0], [2, 0], [3, 0]])
    assert h.impl.tables == []
    h.set_frequency(2.0e9)                                                                    # :100-103
    assert h.l == 299792458.0 / 2.0e9 and len(h.impl.tables) == 1
    assert h.impl.tables[0] == h.array_response and np.array(h.array_response).shape == (360, 4)
    assert h.array_response != h.impl.args[3]
    with pytest.raises(Exception, match="nsamples must be multiple of m"):                    # :58-59
        helper_gr.music_doa_helper(4, 1, 510, 360, 1.0e9, 0.1, [[0, 0], [1, 0], [2, 0], [3, 0]])
