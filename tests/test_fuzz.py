"""Randomised differential runs (HIP vs the C oracles) as part of the GPU suite: a few hundred random shapes /
parameter draws per block, seeds fixed.  The harnesses live in tests/lab/ (they are also run by hand with many more
cases, profiles/r01l_fuzz_differential.txt)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _run(script, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "lab", script)] + [str(a) for a in args],
                       capture_output=True, text=True, timeout=900)
    tail = "\n".join(r.stdout.splitlines()[-15:])      # (stderr carries the blocks' constructor banners)
    assert r.returncode == 0, tail + "\n" + "\n".join(r.stderr.splitlines()[-5:])
    return tail


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [5, 6])
def test_music_path_random_shapes(seed, gpu_device):
    assert "0 failures" in _run("fuzz.py", 250, seed)


@pytest.mark.gpu
def test_frontend_blocks_random_parameters(gpu_device):
    assert "0 failures" in _run("fuzz_frontend.py", 150, 8)


@pytest.mark.gpu
def test_host_fed_path_equals_device_path_for_random_shapes(gpu_device):
    """Chunked host-fed calls, optional ports and peak mode against the device-resident path of the same context:
    bit-identical, i.e. an item's result does not depend on how the batch was cut (converged Jacobi lanes freeze)."""
    assert "0 failures" in _run("fuzz_host.py", 60, 3)
