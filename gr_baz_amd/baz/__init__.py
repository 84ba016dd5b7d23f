"""`baz` python surface of the MUSIC-DoA path (the part of /root/reference/python/__init__.py:61-63
that this repo covers): `baz.music_doa(m, n, nsamples, array_response, resolution)` -- the name
GR_SWIG_BLOCK_MAGIC(baz, music_doa) gives the factory (swig/baz_swig.i:562-564) -- and the
`music_doa_helper` module.

Here the factory is served by the pybind11 module built from gr_baz_amd/host/baz_pybind.cc on top of
the C++ host block; on a real GNU Radio host the unchanged SWIG stanza serves it.
Importing fails loudly when the native module has not been built: there is no python fallback.
"""
import importlib
import os
import sys

_HOST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "host")


def _load_native():
    if _HOST not in sys.path:
        sys.path.insert(0, _HOST)
    try:
        return importlib.import_module("_baz_music")
    except ImportError as e:
        raise ImportError("gr_baz_amd.baz: native module _baz_music is not built (%s); run "
                          "`python -m gr_baz_amd.build` -- there is no CPU fallback" % e)


_native = _load_native()
music_doa = _native.music_doa
baz_music_doa_sptr = _native.baz_music_doa_sptr
agc_cc = _native.agc_cc                    # GR_SWIG_BLOCK_MAGIC(baz, agc_cc) in the reference
baz_agc_cc_sptr = _native.baz_agc_cc_sptr
fractional_resampler_cc = _native.fractional_resampler_cc   # GR_SWIG_BLOCK_MAGIC2(baz, fractional_resampler_cc)
deal_device = _native.deal_device          # placement rule of block instances (not part of the reference surface)

from . import music_doa_helper  # noqa: E402,F401

__all__ = ["music_doa", "baz_music_doa_sptr", "music_doa_helper", "agc_cc", "baz_agc_cc_sptr", "fractional_resampler_cc"]
