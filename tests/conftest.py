import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


def pytest_sessionstart(session):
    """A fresh clone has no built artefacts (the .so files are git-ignored): build what is MISSING once, so that the
    suite does not depend on __graft_entry__.build() having run first.  Never rebuilds what is there."""
    from gr_baz_amd import build as native
    missing = [p for p in (native.HIP_LIB, native.HIP_LAB_LIB, native.AGC_LIB, native.RESAMP_LIB, native.HOST_LIB, native.pybind_module_path())
               if not os.path.exists(p)]
    if missing:
        native.build_all(verbose=False)
    oracle_dir = os.path.join(ROOT, "oracle")
    if any(not os.path.exists(os.path.join(oracle_dir, n)) for n in ("libmusic_ref.so", "libagc_ref.so", "libresamp_ref.so")):
        import subprocess
        subprocess.check_call(["make", "-C", oracle_dir, "libmusic_ref.so", "libagc_ref.so", "libresamp_ref.so"],
                              stdout=subprocess.DEVNULL)


def golden_names():
    """MUSIC-DoA fixtures (agc_* and resamp_* fixtures belong to tests/test_agc.py, tests/test_resamp.py)."""
    return sorted(n for n in (os.path.splitext(os.path.basename(p))[0]
                              for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))) if not n.startswith(("agc_", "resamp_")))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    g = {k: z[k] for k in z.files}
    for k in ("m", "n", "nsamples", "res"):
        g[k] = int(g[k])
    return g


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("gpu-marked test started without a GPU: there is no CPU fallback for the HIP path")
    return torch.device("cuda:0")


# ---- guard zones (round 6) --------------------------------------------------------------------------------------------------------
# BAZ_MUSIC_LAB_LIB=lab BAZ_MUSIC_GUARD=1 python -m pytest tests -m gpu: every device buffer of the MUSIC library lies between two
# 64-KiB zones filled with a pattern (gr_baz_amd/csrc/baz_music_hip.hip, dev_malloc); after EVERY test the zones of all live buffers
# are compared with the pattern, and a test that made a kernel write outside its buffers fails by name.  Off (and free) otherwise.
def _guard_active():
    if os.environ.get("BAZ_MUSIC_GUARD", "0") in ("", "0") or not os.environ.get("BAZ_MUSIC_LAB_LIB"):
        return False
    from gr_baz_amd import capi
    return capi.guard_active(lab=True)


@pytest.fixture(autouse=True)
def _guard_zones_intact(request):
    yield
    if request.node.get_closest_marker("gpu") is not None and _guard_active():
        from gr_baz_amd import capi
        damaged = capi.guard_check(lab=True)
        assert damaged == 0, "%d guard zone(s) of the library's device buffers were overwritten (details on stderr)" % damaged


def pytest_sessionfinish(session, exitstatus):
    try:
        if _guard_active():
            from gr_baz_amd import capi
            print("\n[guard zones] lab library under BAZ_MUSIC_GUARD=1: %d damaged zone(s) over the whole session" % capi.guard_check(lab=True))
    except Exception as e:      # noqa: BLE001  (never turn a finished session into an error)
        print("\n[guard zones] check failed:", e)
