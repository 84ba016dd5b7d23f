"""gr_baz_amd -- MI355X (gfx950) native implementation of gr-baz's MUSIC direction-of-arrival path.

Scope: baz_music_doa::work (/root/reference/lib/baz_music_doa.cc:72-161) and its python helper
(/root/reference/python/music_doa_helper.py), behind the reference's own interfaces:

  gr_baz_amd.capi            ctypes view of the C-ABI (include/baz_music_hip.h)
  gr_baz_amd.baz             `baz.music_doa(...)` / `music_doa_helper(...)` python surface
  gr_baz_amd/host/           C++ gr::sync_block host block (make()/work()/set_array_response())
  gr_baz_amd/csrc/           hand-written HIP kernels + the C-ABI library

There is NO CPU arithmetic path in this package: everything fails loudly without the built
libbaz_music_hip.so and a gfx950 device.  (The CPU oracle lives in /oracle and is test-only.)
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
