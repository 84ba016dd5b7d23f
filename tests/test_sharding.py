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
