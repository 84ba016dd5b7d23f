"""The N>1 path on CPU: world_size-2 gloo processes exercise the stream deal (s mod G), the barrier and
the whole-job-rate arithmetic bench.py uses (no data-path collective exists to test)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from gr_baz_amd import sharding


def test_stream_deal_is_disjoint_and_complete():
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            mine = sharding.streams_of_rank(64, world, r)
            assert all(sharding.stream_owner(s, world) == r for s in mine)
            assert len(mine) == 64 // world
            seen += mine
        assert sorted(seen) == list(range(64))
    assert sharding.streams_of_rank(8, 1, 0) == list(range(8))


def test_block_instances_are_dealt_like_streams():
    """The host block's in-process placement (config 4 as one flowgraph of 64 music_doa blocks): instance i ->
    device i mod G, the same rule as stream -> rank; -1 (current device) when no gfx950 device is visible."""
    from gr_baz_amd import baz
    for g in (1, 2, 4, 8):
        placed = [baz.deal_device(i, g) for i in range(64)]
        assert placed == [sharding.stream_owner(i, g) for i in range(64)]
        assert all(placed.count(d) == 64 // g for d in range(g))
    assert baz.deal_device(5, 0) == -1 and baz.deal_device(0, -3) == -1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    active = sharding.init_process_group(use_gpu=False)
    assert active
    mine = sharding.streams_of_rank(16, world, rank)
    sharding.barrier(active, False)
    # fake "steps": rank r processed len(mine)*100 items in (1 + r) seconds
    rate, tmax, total = sharding.whole_job_rate(len(mine) * 100.0, 1.0 + rank, active, False)
    sharding.barrier(active, False)
    q.put((rank, mine, rate, tmax, total))
    import torch.distributed as dist
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_world_size_2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    (r0, s0, rate0, t0, tot0), (r1, s1, rate1, t1, tot1) = res
    assert sorted(s0 + s1) == list(range(16)) and not set(s0) & set(s1)
    assert tot0 == tot1 == 1600.0          # items of ALL ranks
    assert t0 == t1 == 2.0                 # max over ranks
    assert rate0 == rate1 == 800.0         # whole-job throughput


def test_launcher_rehearsal_eight_dry_ranks():
    """The SCALE day without GPUs (task 7): `python bench.py --gpus 8 --scaling strong --dry-ranks` starts its own
    eight ranks under torch.distributed.run exactly like the real run, deals BASELINE configs[3]'s 64 streams s mod 8, asks for
    RCCL for the barrier / clock, cannot have it here, falls back to gloo -- and the line must SAY so.  (This rehearsal found
    that the fall-back used to pick a new rendezvous port, which under the launcher's agent store has no server behind it:
    every rank waited forever.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "BAZ_BENCH_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--scaling", "strong", "--dry-ranks",
                        "--steps", "2", "--warmup", "1", "--min-seconds", "0.05"], capture_output=True, text=True, timeout=420,
                       cwd=root, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, r.stderr[-2000:]
    d = json.loads(lines[0])
    c = d["config"]
    assert d["dry_run"] is True and d["n_gpus"] == 8 and d["scaling"] == "strong"
    ranks = c["ranks"]
    assert [x["rank"] for x in ranks] == list(range(8))
    assert all(x["streams"] == list(range(x["rank"], 64, 8)) for x in ranks)                 # s mod 8: disjoint and complete
    assert sorted(s for x in ranks for s in x["streams"]) == list(range(64))
    assert all(x["items_per_step"] == 8 * 32768 for x in ranks) and c["items_per_step_all_gpus"] == 64 * 32768
    # asked for RCCL; where it cannot come up (this box: fewer than 8 GPUs) the fall-back is NAMED -- runs on gloo, with the reason.
    # On a box where RCCL does initialise for the 8 ranks the same line reports "nccl" and no fall-back.
    assert c["collective_backend_requested"] == "nccl"
    if c["collective_backend_fell_back"]:
        assert c["collective_backend_for_barrier_and_clock"] == "gloo" and c["collective_backend_fallback_reason"]
        assert "RCCL init FAILED" in r.stderr
    else:
        assert c["collective_backend_for_barrier_and_clock"] == "nccl" and not c["collective_backend_fallback_reason"]
