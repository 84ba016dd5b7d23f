#!/usr/bin/env python3
"""MUSIC-DoA benchmark (driver contract: python bench.py --gpus N --steps K --warmup W).

metric   : BASELINE.json's "MUSIC-DoA snapshots/s (4 ant, 1024 samp, 3600 bins)".
workload : BASELINE.json configs[1] = SURVEY.md 8d cfg2: m=4, n=2, nsamples=1024 (K=256 columns),
           resolution=3600, spectrum port wired; per GPU 8 independent synthetic streams of 32,768
           items (262,144 items = one "step" = one pass of the hot path: covariance -> EVD -> scan),
           device-resident in HBM before the timed region.
N GPUs   : one process per GPU (torch.distributed.run), streams dealt s mod N, NO data-path
           collective; torch.distributed (RCCL) only provides the barrier and the max-over-ranks
           clock.  scaling = weak (per-GPU work fixed).
roofline : dominant kernel = the scan (scan_mfma_kernel); achieved = its algorithmic bytes per launch
           (4*resolution + 8*n per item) / its average launch duration, measured with
           hipEvents recorded on the launch stream inside the timed region (baz_music_profile).
cpu_baseline : rank 0, N=1 only: oracle/_ref (the reference's own baz_music_doa.cc compiled in place, kind
           "reference") when its prebuilt .so is present, else the plain-C restatement (oracle/music_ref.c, kind
           "port"); one work() per item like the GNU Radio scheduler drives the reference, on all host cores
           (one independent block instance per thread) for a bounded sample (~2 s per leg).
"""
import argparse
import json
import os
import sys
import threading
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: required by RCCL on this driver

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

M, N_EMIT, NSAMPLES, RES = 4, 2, 1024, 3600
FREQUENCY, SPACING = 299792458.0, 0.5
STREAMS_PER_GPU, ITEMS_PER_STREAM = 8, 32768
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_MFMA_PEAK_TF = 78.6   # AMD datasheet; ubench: v_mfma_f64_16x16x4_f64 = 65 cycles/SIMD -> 77 TF (profiles/r01_ubench_fp64_rates.txt)


def _usable_cpus():
    """Threads worth starting: the affinity mask, capped by a cgroup CPU quota when one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()            # cgroup v2: "max 100000" | "<quota> <period>"
        if q[0] != "max":
            n = min(n, max(1, int(float(q[0]) / float(q[1]) + 0.5)))
    except Exception:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except Exception:
            pass
    return n


def _time_cpu(fn, items, table, seconds_per_thread, chunk):
    """Items/s of `fn` (one work() per item inside) on 1 thread and on all host threads (independent block
    instances, one per thread -- how T gr-baz blocks would run)."""
    import numpy as np
    sample = items.shape[0]
    fn(items[:16], table, M, N_EMIT)                       # warm-up / page-in
    t0 = time.perf_counter()
    fn(items, table, M, N_EMIT)
    one = sample / (time.perf_counter() - t0)
    cores = _usable_cpus()
    counts = [0] * cores
    stop_at = time.perf_counter() + seconds_per_thread

    def worker(i):
        o = (i * 37) % (sample - chunk)
        x = np.ascontiguousarray(items[o:o + chunk])
        while time.perf_counter() < stop_at:
            fn(x, table, M, N_EMIT)                        # ctypes releases the GIL
            counts[i] += chunk

    th = [threading.Thread(target=worker, args=(i,)) for i in range(cores)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    return sum(counts) / dt, one, cores, sum(counts), dt


def cpu_baseline(table, seconds_per_thread=2.0):
    """Oracle leg (the ONLY use of oracle/ in this file): times the CPU path on the host cores.
    kind "reference": oracle/_ref = the reference's own lib/baz_music_doa.cc compiled in place (prebuilt .so, travels
    with the snapshot), arma::eig_sym backed by LAPACK zheev (scipy's OpenBLAS) when loadable, else the shim's Jacobi.
    kind "port": oracle/music_ref.c, the plain-C restatement (used when oracle/_ref is not there; always reported)."""
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    from oracle import music_oracle as mo
    from oracle import music_ref as mr
    sample = 512
    items = mo.synth_items(sample, M, NSAMPLES, mo.array_geometry(M), FREQUENCY, SPACING, seed=1002)
    pv, pone, cores, pn, pdt = _time_cpu(mr.work_batch, items, table, seconds_per_thread, 64)
    out = {"value": pv, "unit": "snapshots/s", "cores": cores, "kind": "port", "value_1thread": pone,
           "sample": "%d items in %.1f s on %d threads (+%d items on 1 thread); oracle/music_ref.c, "
                     "one work() per item, cfg2 inputs" % (pn, pdt, cores, sample)}
    if mr.have_ref():
        try:
            lapack = mr.ref_use_lapack(True)
            saved = os.dup(2)                              # the reference prints a banner per block instance
            devnull = os.open(os.devnull, os.O_WRONLY)
            os.dup2(devnull, 2)
            try:
                rv, rone, cores, rn, rdt = _time_cpu(mr.ref_work_batch, items, table, seconds_per_thread, 64)
            finally:
                os.dup2(saved, 2)
                os.close(saved)
                os.close(devnull)
            out = {"value": rv, "unit": "snapshots/s", "cores": cores, "kind": "reference", "value_1thread": rone,
                   "eig_backend": "LAPACKE_zheev (scipy OpenBLAS)" if lapack else "Jacobi (oracle/ref_shim)",
                   "port_value": pv, "port_value_1thread": pone,
                   "sample": "%d items in %.1f s on %d threads (+%d items on 1 thread); oracle/_ref = the reference's "
                             "baz_music_doa.cc, one work() per item, cfg2 inputs" % (rn, rdt, cores, sample)}
        except Exception as e:                             # keep the port figure if _ref cannot run here
            out["reference_error"] = repr(e)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ramp-seconds", type=float, default=0.25, help="untimed steady load before warm-up (clock ramp)")
    args = ap.parse_args()

    import numpy as np
    import torch
    from gr_baz_amd import capi, sharding, synth
    from gr_baz_amd.baz.music_doa_helper import calculate_antenna_array_response

    rank, local_rank, world = sharding.dist_env()
    if world != max(1, args.gpus) and world > 1:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MUSIC-DoA path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    active = sharding.init_process_group(use_gpu=True, local_rank=local_rank)

    # steering table exactly as music_doa_helper builds it, rounded to complex64 like SWIG does
    arr = synth.array_geometry(M)
    lam = synth.C_LIGHT / FREQUENCY
    table = np.array(calculate_antenna_array_response([[SPACING * x, SPACING * y] for x, y in arr], RES, lam)
                     ).astype(np.complex64)

    # this rank's streams: global stream s lives on rank s mod world (config 4), seed = 1002 + s
    n_streams = STREAMS_PER_GPU * world
    mine = sharding.streams_of_rank(n_streams, world, rank)
    batch = len(mine) * ITEMS_PER_STREAM
    x = torch.cat([synth.synth_stream(torch, dev, ITEMS_PER_STREAM, M, NSAMPLES, arr, FREQUENCY, SPACING,
                                      seed=1002 + s) for s in mine], dim=0)
    ang = torch.zeros(batch, N_EMIT, dtype=torch.float32, device=dev)
    lvl = torch.zeros_like(ang)
    spec = torch.zeros(batch, RES, dtype=torch.float32, device=dev)

    ctx = capi.Context(M, N_EMIT, NSAMPLES, RES, table, device_id=local_rank)
    ctx.reserve(batch)

    def step():
        ctx.process_device(x.data_ptr(), batch, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())

    # Clock ramp (untimed, before the W warm-up steps): the GPU's power management needs tens of milliseconds of
    # continuous load to leave its idle clocks -- a 3-step (1 ms) warm-up measures the ramp, not the steady state a
    # streaming block runs in (0.42 vs 0.36 ms/step on the same box, profiles/r01g_clock_ramp.txt).
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < args.ramp_seconds:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    sharding.barrier(active, True)
    ctx.profile(int(os.environ.get("BAZ_BENCH_PROFILE", "2")))   # 2: hipEvents around the dominant kernel only
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    sharding.barrier(active, True)
    scan_ms, scan_n = ctx.stage_ms(capi.STAGE_SCAN)       # dominant kernel, timed region only
    ctx.profile(False)
    # informational per-stage breakdown from a separate short pass (events around every kernel add launch gaps,
    # so they stay out of the timed region)
    ctx.profile(1)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    stage = [ctx.stage_ms(s) for s in range(capi.NUM_STAGES)]
    ctx.profile(False)

    value, tmax, total_items = sharding.whole_job_rate(batch * args.steps, elapsed, active, True)

    if rank == 0:
        scan_avg_s = scan_ms / max(scan_n, 1) * 1e-3
        scan_bytes = (4 * RES + 8 * N_EMIT) * batch                 # spectrum + ang/lvl-equivalent written per launch
        achieved = scan_bytes / scan_avg_s / 1e9 if scan_avg_s > 0 else 0.0
        traffic = None
        tj = os.path.join(ROOT, "profiles", "r01_scan_pmc_traffic.json")
        if os.path.exists(tj):
            try:
                tinfo = json.load(open(tj))
                traffic = tinfo.get("scan_hbm_bytes_per_launch")
                if traffic and tinfo.get("items_per_launch") and tinfo["items_per_launch"] != batch:
                    traffic = int(traffic * batch / tinfo["items_per_launch"])   # the profile used another launch size
            except Exception:
                traffic = None
        cov_s = stage[capi.STAGE_COV][0] / max(stage[capi.STAGE_COV][1], 1) * 1e-3
        cov_tf = 8.0 * M * NSAMPLES * batch / cov_s / 1e12 if cov_s > 0 else 0.0
        cov_mfma = {"useful_tflops": cov_tf, "fp64_matrix_peak_tflops": FP64_MFMA_PEAK_TF,
                    "frac_of_peak": cov_tf / FP64_MFMA_PEAK_TF, "issued_over_useful": 2.0,
                    "hbm_read_GBs": 8.0 * NSAMPLES * batch / cov_s / 1e9 if cov_s > 0 else 0.0,
                    "note": "16x16x4 fp64 MFMA tiles hold 2 items block-diagonally at m=4 (half the issued flops are "
                            "structural zeros); the kernel is HBM-read bound"}
        line = {
            "metric": "MUSIC-DoA snapshots/s (4 ant, 1024 samp, 3600 bins)",
            "value": value, "unit": "snapshots/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": tmax / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "cfg2 (BASELINE.json configs[1]): m=4 n=2 nsamples=1024 (K=256) resolution=3600, "
                                   "spectrum port wired, %d streams x %d items per GPU per step, device-resident"
                                   % (STREAMS_PER_GPU, ITEMS_PER_STREAM),
                       "items_per_gpu_per_step": batch, "parallelism": "independent streams, s mod %d, no collective" % world,
                       "algorithmic_bytes_per_item": ctx.bytes_per_item(True),
                       "pipeline_hbm_fraction_of_8TBs": value / world * ctx.bytes_per_item(True) / 8e12,
                       "stage_ms_per_launch_separate_pass": {nm: stage[s][0] / max(stage[s][1], 1)
                                               for s, nm in enumerate(("cov_mfma", "evd_proj", "scan_mfma", "topn_merge"))},
                       # the one dense contraction (north_star): useful fp64 flops 8*m*N per item against the fp64
                       # matrix peak; rocprofv3 MFMA-busy for the same kernel is in profiles/r01g_bench_pmc_summary.txt
                       "covariance_mfma": cov_mfma},
            "roofline": {"bound": "hbm", "kernel": "scan_mfma_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": scan_bytes, "avg_launch_ms": scan_avg_s * 1e3,
                         "launches": scan_n},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(table)
        print(json.dumps(line), flush=True)
    ctx.close()
    if active:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
