"""bench.py's N > 1 path on the GPU box (row e of SURVEY.md 8): `python bench.py --gpus 2` with NO launcher around it
must start its own two ranks and print n_gpus = 2.  The box has one GPU, so the two ranks share it
(BAZ_BENCH_SHARE_DEVICES=1: the only thing the hook changes is local_rank %= ndev and the barrier backend, RCCL cannot
put two ranks on one device); everything else -- the deal s mod N, one context per rank, barrier + max-over-ranks
clock, the gathered per-rank records -- is the code an 8-GPU SCALE run executes."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, env=None, timeout=420):
    e = dict(os.environ, **(env or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT, env=e)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_two_ranks_share_one_gpu(gpu_device):
    common = ("--steps", "3", "--warmup", "2", "--no-extras", "--no-cpu-baseline", "--min-seconds", "0.2")
    r1, one = _bench("--gpus", "1", *common)
    assert r1.returncode == 0 and one is not None, r1.stderr[-2000:]
    assert one["n_gpus"] == 1 and len(one["config"]["ranks"]) == 1
    r2, two = _bench("--gpus", "2", *common, env={"BAZ_BENCH_SHARE_DEVICES": "1"})
    assert r2.returncode == 0 and two is not None, r2.stderr[-2000:]
    assert two["n_gpus"] == 2 and two["scaling"] == "weak"
    ranks = two["config"]["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1]
    s0, s1 = set(ranks[0]["streams"]), set(ranks[1]["streams"])
    assert not (s0 & s1) and sorted(s0 | s1) == list(range(16))          # 8 streams per rank, dealt s mod 2
    assert all(r["items_per_step"] == one["config"]["ranks"][0]["items_per_step"] for r in ranks)   # weak scaling
    # both ranks time-share ONE device here: the whole-job rate is what that device delivers (within 2x of N = 1)
    assert 0.5 * one["value"] <= two["value"] <= 1.5 * one["value"], (one["value"], two["value"])


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_strong_scaling_is_config_4_as_written(gpu_device):
    """BASELINE configs[3] / SURVEY.md 8d cfg4: the SAME 64 streams at every GPU count, dealt s mod G.  N = 1 runs all
    2,097,152 items per step (8 launch sequences of 262,144); at N = 2 every rank holds 32 disjoint streams and half the
    items, and the whole-job items per step do not change."""
    common = ("--scaling", "strong", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--min-seconds", "0.05",
              "--ramp-seconds", "0.05")
    r1, one = _bench("--gpus", "1", *common)
    assert r1.returncode == 0 and one is not None, r1.stderr[-2000:]
    assert one["scaling"] == "strong" and one["n_gpus"] == 1
    c1 = one["config"]
    assert c1["items_per_gpu_per_step"] == 2097152 == c1["items_per_step_all_gpus"] and c1["streams_total"] == 64
    assert c1["ranks"][0]["streams"] == list(range(64)) and c1["launch_sequences_per_step"] == 8
    assert abs(one["value"] - 2097152 / (one["ms_per_step"] * 1e-3)) <= 1e-6 * one["value"]
    r2, two = _bench("--gpus", "2", *common, env={"BAZ_BENCH_SHARE_DEVICES": "1"})
    assert r2.returncode == 0 and two is not None, r2.stderr[-2000:]
    assert two["scaling"] == "strong" and two["n_gpus"] == 2
    ranks = two["config"]["ranks"]
    s0, s1 = set(ranks[0]["streams"]), set(ranks[1]["streams"])
    assert not (s0 & s1) and sorted(s0 | s1) == list(range(64))          # disjoint and complete
    assert s0 == set(range(0, 64, 2)) and s1 == set(range(1, 64, 2))     # s mod 2
    assert all(r["items_per_step"] == 1048576 for r in ranks)            # half the items per rank ...
    assert two["config"]["items_per_step_all_gpus"] == 2097152           # ... the same job
    assert 0.5 * one["value"] <= two["value"] <= 1.5 * one["value"], (one["value"], two["value"])   # one shared device


def test_bench_strong_scaling_needs_a_gpu_count_that_divides_64():
    e = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--scaling", "strong", "--steps", "1"],
                       capture_output=True, text=True, timeout=240, cwd=ROOT, env=e)
    assert r.returncode != 0 and "must divide" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_bench_refuses_more_ranks_than_devices(gpu_device):
    import torch
    n = torch.cuda.device_count() + 1
    r, line = _bench("--gpus", str(n), "--steps", "1", "--warmup", "0", "--no-extras", "--no-cpu-baseline")
    assert r.returncode != 0 and line is None
    assert "visible" in (r.stderr + r.stdout)


def test_bench_rejects_a_world_size_that_is_not_gpus():
    """The line must never carry an n_gpus other than the one asked for (round-2 review: --gpus 8 under WORLD_SIZE=1
    printed n_gpus 1 with rc 0).  No GPU needed: the check comes before any device work."""
    e = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1"], capture_output=True,
                       text=True, timeout=240, cwd=ROOT, env=e)
    assert r.returncode != 0 and "does not match --gpus 8" in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_64_block_instances_on_their_own_threads(gpu_device):
    """Config 4 as ONE flowgraph process: 64 music_doa blocks, a host thread each (GNU Radio's thread-per-
    block scheduler), dealt over the visible devices by the host block (instance i -> device i mod G).  Every stream must
    equal its single-threaded run -- 64 contexts with their own streams, tables and workspaces do not disturb each other --
    and the harness reports the whole-process rate."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "deal_harness.py"), "64", "128", "3"], capture_output=True,
                       text=True, timeout=540, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, r.stderr[-2000:]
    d = json.loads(lines[-1])
    assert d["blocks"] == 64 and d["streams_identical_to_single_threaded_run"] is True
    per_dev = d["instances_per_device"]
    assert sum(per_dev.values()) == 64 and len(per_dev) == d["devices_visible"]
    assert max(per_dev.values()) - min(per_dev.values()) <= 1                      # dealt evenly
    assert d["items_per_s_all_blocks"] > 1e5



@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_rccl_really_initialises_under_the_bench(gpu_device):
    """No 8-GPU node has been available to this work, so the N > 1 runs above fall back to gloo (two ranks cannot share a device under
    RCCL).  What CAN run on the 1-GPU box is the RCCL side of it with a group of one rank (BAZ_BENCH_FORCE_DIST=1): the communicator is
    created on the device, the barrier (device_ids) and the max / sum all-reduces of the clock run on GPU tensors through RCCL, the rank
    records are gathered -- and the line says nccl, not a fall-back."""
    r, d = _bench("--gpus", "1", "--steps", "3", "--warmup", "2", "--no-extras", "--no-cpu-baseline", "--min-seconds", "0.2",
                  env={"BAZ_BENCH_FORCE_DIST": "1"})
    assert r.returncode == 0 and d is not None, r.stderr[-2000:]
    c = d["config"]
    assert c["collective_backend_requested"] == "nccl" and c["collective_backend_for_barrier_and_clock"] == "nccl", c
    assert c["collective_backend_fell_back"] is False and not c["collective_backend_fallback_reason"]
    assert d["n_gpus"] == 1 and len(c["ranks"]) == 1 and c["verified_ok"] is True
    assert d["value"] > 1e8
