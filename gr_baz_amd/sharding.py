"""Multi-GPU layout of the MUSIC-DoA path: independent snapshot streams, no data-path collective.

Every item (one work() call, lib/baz_music_doa.cc:72-161) is independent of every other -- the block
carries no state across items except the read-only steering table -- so streams shard
embarrassingly: stream s lives on rank s mod G (BASELINE.json config 4).  One process
per GPU; torch.distributed (backend "nccl" = RCCL on the GPU box, "gloo" in CPU tests) is used ONLY
for the start/stop barrier and the max-over-ranks clock of the benchmark contract, never for data.
"""
from __future__ import annotations

import os


def stream_owner(stream: int, world: int) -> int:
    return stream % world


def streams_of_rank(n_streams: int, world: int, rank: int):
    """Round-robin deal (s mod G): disjoint over ranks, complete over [0, n_streams)."""
    return [s for s in range(n_streams) if stream_owner(s, world) == rank]


def dist_env():
    """(rank, local_rank, world) from the torch.distributed.run environment (1 process per GPU)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(use_gpu: bool, local_rank: int = 0, try_nccl: bool = False):
    """Initialises torch.distributed when WORLD_SIZE > 1.  Returns True if a group is active.
    The backend asked for is "nccl" (= RCCL) on a GPU run -- or wherever try_nccl says so (bench.py --dry-ranks: the launcher
    rehearsal on a box without GPUs, which must go through the same fall-back) -- unless BAZ_BENCH_BACKEND names another.  When
    RCCL cannot initialise, the barrier / clock fall back to gloo LOUDLY: on stderr and in backend_info(), which bench.py prints
    into its JSON line (collective_backend_*): a fall-back must never pass for an RCCL run."""
    import torch.distributed as dist
    rank, _, world = dist_env()
    # BAZ_BENCH_FORCE_DIST=1: a group of ONE rank (tests/test_bench_multi.py on the 1-GPU box: RCCL really initialises under this
    # code, the barrier and the clock really run through it -- everything an N-rank run does except a second rank)
    if world <= 1 and os.environ.get("BAZ_BENCH_FORCE_DIST") != "1":
        return False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        global _BACKEND, _REQUESTED, _FALLBACK_REASON
        backend = "gloo"
        if use_gpu:
            import torch
            torch.cuda.set_device(local_rank)
        elif try_nccl:
            import torch
            if torch.cuda.is_available() and local_rank < torch.cuda.device_count():
                torch.cuda.set_device(local_rank)          # a rehearsal on a box WITH GPUs: RCCL comes up, its tensors live on this rank's device
        if (use_gpu or try_nccl) and os.environ.get("BAZ_BENCH_BACKEND", "nccl") == "nccl":
            backend = "nccl"          # = RCCL on ROCm; used for the barrier / clock only
        _REQUESTED = backend
        if backend == "nccl":
            import torch
            try:
                dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                        device_id=torch.device("cuda", local_rank))
                dist.barrier(device_ids=[local_rank])          # first collective: creates the RCCL communicator
            except Exception as e:   # RCCL unusable on this box: the data path needs no collective anyway
                import sys
                _FALLBACK_REASON = "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:200] if str(e) else "")
                print("[gr_baz_amd.sharding] RCCL init FAILED (%s); using gloo for the barrier/clock" % _FALLBACK_REASON,
                      file=sys.stderr, flush=True)
                if dist.is_initialized():
                    dist.destroy_process_group()
                # Under torch.distributed.run the AGENT hosts the rendezvous store on MASTER_PORT and the ranks are its clients:
                # the gloo group is set up through the same store (found by bench.py --dry-ranks: a new port has no server
                # behind it and every rank waits forever).  Without an agent rank 0 hosted the store itself and the failed
                # attempt may still hold the port: move on by one.
                if os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "").lower() != "true":
                    os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29511")) + 1)
                backend = "gloo"
                dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        _BACKEND = backend
    return True


_BACKEND = "gloo"
_REQUESTED = "gloo"
_FALLBACK_REASON = None


def backend_info():
    """What the barrier / clock actually run on: {"backend", "requested", "fell_back", "fallback_reason"}."""
    return {"backend": _BACKEND, "requested": _REQUESTED, "fell_back": _BACKEND != _REQUESTED, "fallback_reason": _FALLBACK_REASON}


def _on_gpu(use_gpu: bool) -> bool:
    return use_gpu and _BACKEND == "nccl"


def barrier(active: bool, use_gpu: bool):
    if active:
        import torch.distributed as dist
        if _on_gpu(use_gpu):
            import torch
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def max_over_ranks(value: float, active: bool, use_gpu: bool) -> float:
    if not active:
        return value
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device="cuda" if _on_gpu(use_gpu) else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, active: bool, use_gpu: bool) -> float:
    if not active:
        return value
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device="cuda" if _on_gpu(use_gpu) else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def whole_job_rate(items_this_rank: float, elapsed_this_rank: float, active: bool, use_gpu: bool):
    """value of the benchmark contract: units all ranks processed / max-over-ranks time."""
    total = sum_over_ranks(items_this_rank, active, use_gpu)
    tmax = max_over_ranks(elapsed_this_rank, active, use_gpu)
    return total / tmax, tmax, total
