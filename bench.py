#!/usr/bin/env python3
"""MUSIC-DoA benchmark (driver contract: python bench.py --gpus N --steps K --warmup W).

metric   : BASELINE.json's "MUSIC-DoA snapshots/s (4 ant, 1024 samp, 3600 bins)".
workload : BASELINE.json configs[1] = SURVEY.md 8d cfg2: m=4, n=2, nsamples=1024 (K=256 columns),
           resolution=3600, spectrum port wired; per GPU 8 independent synthetic streams of 32,768
           items (262,144 items = one "step" = one pass of the hot path: covariance -> EVD -> scan -> merge),
           device-resident in HBM before the timed region.
timing   : W warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides, max over ranks.
           A K-step region is ~25 ms at the driver's K = 20, so the measurement is REPEATED (rounds of exactly K
           steps, each bracketed the same way) until >= --min-seconds of timed work has accumulated; the line
           reports the MEDIAN round (ms_per_step, value) and lists every round under "rounds".
N GPUs   : one process per GPU, streams dealt s mod N, NO data-path collective; torch.distributed (RCCL) only
           provides the barrier and the max-over-ranks clock.  --scaling weak (default; per-GPU work fixed: 8 streams per
           GPU) or --scaling strong = BASELINE configs[3] as written: the SAME 64 streams (seed 1002 + s) at every N,
           dealt s mod N, a step = every stream of the rank once (groups of 8 streams per launch sequence), so N = 1
           runs 2,097,152 items per step (17 GB in, 30 GB out, device-resident) and N = 8 262,144 per GPU.  "ranks" lists
           what every rank processed (asserted to be N entries).  Started either by the driver's
           `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` or as plain
           `python bench.py --gpus N`, which re-runs itself under that launcher (self_launch); a WORLD_SIZE that is
           not N, or fewer than N visible devices, is an error, never a silently smaller run.
roofline : dominant kernel = the scan (scan_mfma_kernel); achieved = its algorithmic bytes per launch
           (4*resolution + 8*n per item) / its average launch duration, measured with
           hipEvents recorded on the launch stream inside the timed rounds (baz_music_profile).  "traffic" = HBM
           bytes per launch from the rocprofv3 PMC passes kept under profiles/ -- used ONLY when that profile was
           taken on the kernel sources this run executes (sha256 over gr_baz_amd/csrc + include), else null and
           "traffic_stale": true.
output   : exactly ONE JSON line on stdout.  Plain `python bench.py` (one GPU) measures in a CHILD of itself (supervise()): the child hands its
           complete line -- metric, roofline, cpu_baseline, verification -- to the GPU-free parent (and to stderr) BEFORE the secondary legs start and
           the same line with the legs' figures after them; the parent prints the last one it has got, restarts the child once if a signal killed
           it before any line existed (config.headline_attempt), and otherwise passes its exit code on.
extras   : config.extra carries secondary, clearly labelled measurements (rank 0, N=1), EACH IN ITS OWN PROCESS (python bench.py --extra-leg NAME,
           time limit 150 s): the cost of retunes beside the headline's steps, cfg2 WITHOUT the spectrum port (the GRC default wiring), the
           unfavourable cfg2 cases (an incoherent batch with and without the spectrum port, 60 dB SNR), cfg3 (8 ant, 4096 samp, 36000 bins; and without
           port 2), 32 and 64 antennas, the cfg5 chain (16 ant: resampler -> AGC -> MUSIC on one stream) and the host-fed flowgraph model, each with its
           own ms, bound, fraction and oracle check.  A leg that faults, hangs or raises becomes config.extra.<leg>.error + config.extras_failed; the
           exit code speaks for the headline's verification only.  (Round 5's driver run died with a GPU memory fault inside one of these, in-process
           then, before anything had been printed: DESIGN.md 6.1.)
cpu_baseline : rank 0, N=1 only: oracle/_ref (the reference's own baz_music_doa.cc compiled in place, kind
           "reference") when its prebuilt .so is present, else the plain-C restatement (oracle/music_ref.c, kind
           "port"); one work() per item like the GNU Radio scheduler drives the reference, on all host cores
           (one independent block instance per thread) for a bounded sample (~2 s per leg).
"""
import argparse
import hashlib
import json
import os
import statistics
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
STRONG_STREAMS = 64        # BASELINE configs[3]: 64 independent 4-antenna streams, the same ones at every GPU count
T_START = time.perf_counter()
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
FP64_MFMA_PEAK_TF = 78.6   # AMD datasheet; ubench: v_mfma_f64_16x16x4_f64 = 65 cycles/SIMD -> 77 TF (profiles/r01_ubench_fp64_rates.txt)
# the newest kept PMC traffic profile (scripts/gpu/rNN_final.sh writes it before the evidence bench line is taken)
TRAFFIC_PROFILE = next((p for p in (os.path.join(ROOT, "profiles", "%s_scan_pmc_traffic.json" % r) for r in ("r06", "r05")) if os.path.exists(p)),
                       os.path.join(ROOT, "profiles", "r06_scan_pmc_traffic.json"))


def kernel_sources_sha():
    """sha256 over the sources the device code is built from: ties a kept rocprof profile to the code that ran."""
    h = hashlib.sha256()
    for d in (os.path.join(ROOT, "gr_baz_amd", "csrc"), os.path.join(ROOT, "include")):
        for name in sorted(os.listdir(d)):
            if name.endswith((".hip", ".h")):
                h.update(name.encode())
                h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


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


I8_MFMA_PEAK_TOPS = 5000.0   # dense int8 = 2 x the ~2.5 PF bf16 dense peak (MI355X_MICROARCH.md, matrix cores; ubench >= 3944 TOPS)
VERIFY_TOL = 1e-5            # north_star's relative tolerance on float outputs


def verify_against_oracle(torch, np, x, ang, lvl, spec, table, m, n, nsamples, res, count):
    """The line proves what it timed: AFTER the timed region, `count` items spread over the batch are
    copied back together with what the LAST timed step left in the output buffers, and the CPU oracle (the plain-C restatement
    of work(), oracle/music_ref.c -- the checker, never the thing measured) recomputes them.  Returns scalar keys:
    verified_items, verified_max_rel_err (spectrum and lvl, relative), verified_bins_identical (DoA bins equal, or different
    only between bins whose reference strengths agree to 2e-5: the reference's own tie rule), verified_ok."""
    from oracle import music_ref as mr
    B = x.shape[0]
    idx = np.unique(np.linspace(0, B - 1, min(count, B)).astype(np.int64))
    ti = torch.from_numpy(idx).to(x.device)
    items = x.index_select(0, ti).cpu().numpy().view(np.complex64).reshape(len(idx), nsamples)
    ga = ang.index_select(0, ti).cpu().numpy()
    gl = lvl.index_select(0, ti).cpu().numpy().astype(np.float64)
    ao, lo, so = mr.work_batch(np.ascontiguousarray(items), table, m, n)
    worst = float(np.max(np.abs(gl - lo) / lo))
    if spec is not None:
        gs = spec.index_select(0, ti).cpu().numpy().astype(np.float64)
        worst = max(worst, float(np.max(np.abs(gs - so) / so)))
    swaps, wrong = 0, 0
    if not np.array_equal(ga, ao):
        for r in np.flatnonzero(np.any(ga != ao, axis=1)):
            # same bins in another order, or a bin whose reference strength ties with the reference's choice
            ref_bins = np.rint(ao[r].astype(np.float64) * res / 360.0).astype(np.int64) % res
            got_bins = np.rint(ga[r].astype(np.float64) * res / 360.0).astype(np.int64) % res
            tie = np.all(np.abs(so[r][got_bins] - so[r][ref_bins]) <= 2e-5 * so[r][ref_bins])
            swaps += 1 if tie else 0
            wrong += 0 if tie else 1
    ok = bool(worst <= VERIFY_TOL and wrong == 0 and np.isfinite(worst))
    return {"verified_items": int(len(idx)), "verified_max_rel_err": worst, "verified_bins_identical": bool(wrong == 0),
            "verified_bins_tie_swaps": int(swaps), "verified_ok": ok, "verified_against": "oracle/music_ref.c (plain-C restatement of work())"}


def time_retunes(np, synth, ctx, table, m, res, arr, step, sync, count=5):
    """set_array_response while batches are in flight: wall milliseconds of baz_music_set_table (median of `count`, alternating
    between two tables; the last call restores `table`) and the part spent on the lock shared with work()."""
    lam2 = synth.C_LIGHT / (FREQUENCY * 0.9)
    from gr_baz_amd.baz.music_doa_helper import calculate_antenna_array_response
    t2 = np.array(calculate_antenna_array_response([[SPACING * x, SPACING * y] for x, y in arr], res, lam2)).astype(np.complex64)
    walls, locks = [], []
    for _ in range(8):          # (the device idled through the oracle check: steps at ramping clocks are not what a retune is measured beside)
        step()
    sync()
    for k in range(2 * count):
        step()
        step()
        t0 = time.perf_counter()
        ctx.set_table(t2 if (k & 1) == 0 else table)
        walls.append((time.perf_counter() - t0) * 1e3)
        locks.append(ctx.last_retune_ms()[1])
    sync()
    if os.environ.get("BAZ_BENCH_TRACE_RETUNE"):
        print("retune walls ms:", [round(w, 3) for w in walls], file=sys.stderr)
    return statistics.median(walls), max(walls), max(locks)


def helper_table(np, synth, m, res):
    """Steering table exactly as music_doa_helper builds it, rounded to complex64 like SWIG does."""
    from gr_baz_amd.baz.music_doa_helper import calculate_antenna_array_response
    arr = synth.array_geometry(m)
    lam = synth.C_LIGHT / FREQUENCY
    return arr, np.array(calculate_antenna_array_response([[SPACING * x, SPACING * y] for x, y in arr], res, lam)
                         ).astype(np.complex64)


def timed_loop(torch, step, sync, min_seconds, chunk=5, max_steps=2000):
    """ms per step of `step()` after a short ramp: at least min_seconds of back-to-back launches."""
    for _ in range(chunk):
        step()
    sync()
    n, t0 = 0, time.perf_counter()
    while True:
        for _ in range(chunk):
            step()
        sync()
        n += chunk
        if time.perf_counter() - t0 >= min_seconds or n >= max_steps:
            break
    return (time.perf_counter() - t0) / n * 1e3, n


def extra_music(torch, np, capi, synth, dev, stream, m, nsamples, res, batch, with_spectrum, seconds, scene="coherent", snr_db=20.0,
                emit=None, retunes=3):
    """One secondary MUSIC configuration on the bench stream: ms/step, per-stage ms, dominant-kernel rooflines.
    scene "coherent": 8 streams, every item of a stream sees the same two emitters (the headline's inputs);
    scene "incoherent": the emitter angles are drawn per ITEM (the unfavourable case for everything the scan decides per
    wave of 16 items: the top-n gate, the literal-form refinement, the coarse-gated scan's tile votes).
    `emit(dict)`: called with the leg's figures BEFORE the retunes-in-flight part starts (a leg runs in its own process and prints
    what it has as it goes: a fault in the later part does not take the earlier figures with it)."""
    arr, table = helper_table(np, synth, m, res)
    per = batch // 8
    if scene == "incoherent":
        x = synth.synth_scenes(torch, dev, batch, m, nsamples, arr, FREQUENCY, SPACING, N_EMIT, snr_db=snr_db, seed=1000 + 7)
    else:
        x = torch.cat([synth.synth_stream(torch, dev, per, m, nsamples, arr, FREQUENCY, SPACING, snr_db=snr_db, seed=1000 + 3 + s)
                       for s in range(8)], dim=0)
    ang = torch.zeros(batch, N_EMIT, dtype=torch.float32, device=dev)
    lvl = torch.zeros_like(ang)
    spec = torch.zeros(batch, res, dtype=torch.float32, device=dev) if with_spectrum else None
    torch.cuda.synchronize()                     # the fills above ran on torch's current stream, the engine launches on `stream`
    with capi.Context(m, N_EMIT, nsamples, res, table, device_id=dev.index) as ctx:
        ctx.set_stream(stream.cuda_stream)
        ctx.reserve(batch)
        sp = spec.data_ptr() if with_spectrum else None
        step = lambda: ctx.process_device(x.data_ptr(), batch, ang.data_ptr(), lvl.data_ptr(), sp)
        ms, n = timed_loop(torch, step, stream.synchronize, seconds)
        ctx.profile(1)
        for _ in range(3):
            step()
        stream.synchronize()
        st = [ctx.stage_ms(s) for s in range(capi.NUM_STAGES)]
        ctx.profile(False)
        step()
        refined = ctx.refined_values()
        bpi = ctx.bytes_per_item(with_spectrum)
        scan_kernel = ctx.stage_name(capi.STAGE_SCAN)       # the kernel the last launch took
        uses_i8 = "scan_i8_kernel" in scan_kernel
        stream.synchronize()
        verified = verify_against_oracle(torch, np, x, ang, lvl, spec, table, m, N_EMIT, nsamples, res, 16)
        out = _music_leg_figures(m, nsamples, res, batch, with_spectrum, scene, snr_db, ms, n, st, refined, bpi, scan_kernel, uses_i8)
        out.update(verified)
        if emit is not None:
            emit(dict(out, partial="before the retunes-in-flight part"))
        if retunes:
            note("  retunes with steps in flight")
            retune_med, retune_max, retune_lock = time_retunes(np, synth, ctx, table, m, res, arr, step, stream.synchronize, retunes)
            out.update({"retune_ms": retune_med, "retune_ms_worst": retune_max, "retune_lock_ms_worst": retune_lock})
            # ... and the outputs of a step AFTER the last retune (which restored `table`) still agree with the oracle
            step()
            stream.synchronize()
            again = verify_against_oracle(torch, np, x, ang, lvl, spec, table, m, N_EMIT, nsamples, res, 8)
            out["verified_after_retunes_ok"] = again["verified_ok"]
            out["verified_ok"] = bool(out["verified_ok"] and again["verified_ok"])
        ctx.set_stream(None)
    return out


def _music_leg_figures(m, nsamples, res, batch, with_spectrum, scene, snr_db, ms, n, st, refined, bpi, scan_kernel, uses_i8):
    stage = {nm: st[s][0] / max(st[s][1], 1) for s, nm in enumerate(("cov", "evd", "scan", "merge"))}
    scan_s = stage["scan"] * 1e-3
    mm = m * m
    scan_tf = 2.0 * mm * res * batch / scan_s / 1e12 if scan_s > 0 else 0.0
    out = {"items_per_step": batch, "ms_per_step": ms, "steps_timed": n, "snapshots_per_s": batch / ms * 1e3,
           "scene": scene, "snr_db": snr_db, "values_recomputed_in_literal_form_per_step": refined,
           "algorithmic_bytes_per_item": bpi, "pipeline_hbm_fraction_of_8TBs": batch / ms * 1e3 * bpi / 8e12,
           "stage_ms_per_launch": stage, "scan_kernel_launched": scan_kernel}
    if uses_i8:
        # int8 matrix core: the first tier's 10 v_mfma_i32_16x16x64_i8 per 16 x 16 tile and block of 64 terms (every tile runs
        # them; the second / third tiers add to it where values need more digits) against the int8 dense peak -- a lower bound of
        # the matrix work, never an "fp64-equivalent" rate
        nkb = -(-mm // 64)
        tiles = -(-res // 64) * 4
        i8_tops = 2.0 * 64 * 256 * 10 * nkb * tiles * (batch / 16.0) / scan_s / 1e12 if scan_s > 0 else 0.0
        out["scan_form"] = "int8 digits, exact int32 accumulation (scan_i8_kernel); first tier = 10 MFMAs per tile and 64 terms"
        out["scan_int8_tops_first_tier"] = i8_tops
        out["scan_frac_of_int8_matrix_peak_5000TOPS"] = i8_tops / I8_MFMA_PEAK_TOPS
    elif m > 16:                     # run-time-m kernels; to 32 antennas the short form on the fp64 matrix core (n <= 2)
        sf = 2.0 * 4 * N_EMIT * m * res * batch / scan_s / 1e12 if scan_s > 0 else 0.0
        out["scan_form"] = "short form, 4*n*m FMA per (item, bin), scan_wide_mfma_kernel"
        out["scan_fp64_tflops"] = sf
        out["scan_frac_of_fp64_matrix_peak"] = sf / FP64_MFMA_PEAK_TF
    elif with_spectrum or m > 8:   # (without the spectrum port and m <= 8 the scan is the coarse-gated one: its fp64 work is a few tiles)
        out["scan_fp64_tflops"] = scan_tf
        out["scan_frac_of_fp64_matrix_peak"] = scan_tf / FP64_MFMA_PEAK_TF
    else:
        # two coarse passes of K = 32 f16 MFMAs per 16 x 16 tile -- 2 (m <= 4: [qh|ql] x [Fh|Fh], [Fl|Fl]) or 3 per group of 32
        # terms (qh Fh, ql Fh, qh Fl) --, bins padded to 16 x 8
        tiles = -(-res // 128) * 8
        nmfma = 2 if m <= 4 else 3 * -(-mm // 32)
        f16_tf = 2.0 * 2 * 32 * nmfma * 16 * tiles * batch / scan_s / 1e12 if scan_s > 0 else 0.0
        out["scan_kernel"] = "scan_coarse_kernel (f16-matrix-core coarse form gates the fp64 tiles; ang / lvl bit-identical to the full scan)"
        out["scan_f16_tflops"] = f16_tf
        out["scan_frac_of_f16_matrix_peak_2500TF"] = f16_tf / 2500.0
        out["hbm_read_fraction_of_8TBs"] = batch / ms * 1e3 * 8.0 * nsamples / 8e12
    if with_spectrum:
        wr = (4 * res + 8 * N_EMIT) * batch / scan_s / 1e9 if scan_s > 0 else 0.0
        out["scan_write_GBs"] = wr
        out["scan_frac_of_hbm_8TBs"] = wr / HBM_PEAK_GBS
    return out


def extra_cfg5(torch, np, capi, synth, dev, stream, nitems, seconds):
    """BASELINE config 5 on one GPU: 16-antenna front-end (fractional resampler -> AGC + interleave) ahead of MUSIC
    (m16, n2, N4096 = 16 x 256, res3600), device resident, one stream, nothing leaves HBM between the engines."""
    from gr_baz_amd import agc, resamp
    m, K, res, ratio = 16, 256, 3600, 1.25
    N = m * K
    arr, table = helper_table(np, synth, m, res)
    T_out = nitems * K
    L = int(T_out * ratio) + 16
    it = synth.synth_stream(torch, dev, (L + K - 1) // K, m, N, arr, FREQUENCY, SPACING, seed=1005)
    raw = torch.view_as_real(it.view(torch.complex64).reshape(-1, m).t().contiguous()[:, :L].contiguous()).reshape(m, 2 * L)
    del it
    d_rs = torch.zeros(m, 2 * T_out, dtype=torch.float32, device=dev)
    d_items = torch.zeros(nitems, 2 * N, dtype=torch.float32, device=dev)
    ang = torch.zeros(nitems, N_EMIT, dtype=torch.float32, device=dev)
    lvl = torch.zeros_like(ang)
    spec = torch.zeros(nitems, res, dtype=torch.float32, device=dev)
    R = resamp.Resampler(0.0, ratio, nstreams=m)
    A = agc.Agc(1e-4, 1.0, nstreams=m)
    Mx = capi.Context(m, N_EMIT, N, res, table, device_id=dev.index)
    try:
        Mx.reserve(nitems)
        for e in (R, A, Mx):
            e.set_stream(stream.cuda_stream)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

        def step(timed=False):
            R.set_mu(0.0)                                    # every step resamples the same capture from its start
            if timed:
                ev[0].record(stream)
            R.process_device(raw.data_ptr(), L, L, d_rs.data_ptr(), T_out, T_out)
            if timed:
                ev[1].record(stream)
            A.process_device_interleaved(d_rs.data_ptr(), T_out, T_out, d_items.data_ptr())
            if timed:
                ev[2].record(stream)
            Mx.process_device(d_items.data_ptr(), nitems, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
            if timed:
                ev[3].record(stream)

        torch.cuda.synchronize()
        ms, n = timed_loop(torch, step, stream.synchronize, seconds)
        step(True)
        stream.synchronize()
        eng = [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]
        Mx.profile(1)
        step()
        stream.synchronize()
        st = [Mx.stage_ms(s)[0] for s in range(capi.NUM_STAGES)]
        Mx.profile(False)
        stream.synchronize()
        # the MUSIC stage of the chain on the items the front-end handed it (the front-end engines have their own parity tests)
        verified = verify_against_oracle(torch, np, d_items, ang, lvl, spec, table, m, N_EMIT, N, res, 8)
    finally:
        for e in (R, A, Mx):
            e.set_stream(None)
            e.close()
    in_b, rs_b = m * L * 8, m * T_out * 8
    chain_bytes = in_b + rs_b + rs_b + nitems * N * 8 + nitems * N * 8 + nitems * (4 * res + 8 * N_EMIT)
    return {**verified, "verified_stage": "music (on the front-end's output)",
            "items_per_step": nitems, "ms_per_step": ms, "steps_timed": n, "snapshots_per_s": nitems / ms * 1e3,
            "complex_samples_per_s_per_antenna": T_out / ms * 1e3,
            "engine_ms": {"resampler": eng[0], "agc_interleave": eng[1], "music": eng[2]},
            "music_stage_ms": dict(zip(("cov", "evd", "scan", "merge"), st)),
            "bound": "hbm (front-end: resampler + AGC) + fp64 matrix (scan); the EVD is the n = 2 signal subspace by "
                     "orthogonal iteration (evd_sub_kernel), Jacobi only for items it hands back",
            # the scan runs the short form ||a||^2 - sum_c |s_c^H a|^2: 4 n m FMAs per (item, bin), not m^2
            "scan_fp64_tflops": 2.0 * 4 * N_EMIT * m * res * nitems / (st[2] * 1e-3) / 1e12 if st[2] > 0 else None,
            "scan_form": "short form, 4*n*m = %d FMA per (item, bin) (projector form: m^2 = %d)" % (4 * N_EMIT * m, m * m),
            "resampler_GBs": (in_b + rs_b) / eng[0] / 1e6, "agc_GBs": (rs_b + 2 * nitems * N * 8) / eng[1] / 1e6,
            "chain_bytes_per_step": chain_bytes, "chain_hbm_fraction_of_8TBs": chain_bytes / (ms * 1e-3) / 8e12}


def note(msg):
    """Progress on stderr (the JSON lines own stdout): which leg a run was in when something went wrong."""
    print("bench.py: [%.1f s] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


# ---- secondary measurements: every leg runs in ITS OWN PROCESS (python bench.py --extra-leg NAME) -------------------------------
# (the driver's round-5 run died with a GPU memory fault inside one of these, after the headline had been measured and
# before anything had been printed.  Now the parent prints the complete headline line first, and a leg that faults, hangs or raises
# becomes {"error": ...} under config.extra.<leg> -- it cannot take the line, or the legs after it, with it.)
MUSIC_LEGS = {
    # name: (m, nsamples, res, batch, with_spectrum, seconds, scene, snr_db, {labels})
    "cfg2_without_spectrum_port": (4, 1024, 3600, 262144, False, 0.4, "coherent", 20.0, {
        "bound": "HBM read (covariance + EVD kernel, 63 % of the step) + f16 matrix (the scan's two coarse passes)",
        "workload": "cfg2 with only ang/lvl wired (music_doa_helper's default output_spectrum=False)"}),
    "cfg2_incoherent_scene": (4, 1024, 3600, 262144, True, 0.4, "incoherent", 20.0, {
        "workload": "cfg2, spectrum port wired, emitter angles drawn per ITEM: the top-n gate of the scan fires in nearly every step"}),
    "cfg2_incoherent_scene_without_spectrum_port": (4, 1024, 3600, 262144, False, 0.4, "incoherent", 20.0, {
        "workload": "cfg2, ang/lvl only, emitter angles drawn per ITEM: ~22 % of the (16-item, 16-bin) tiles still run the exact "
                    "form (the union over a wave's 16 unrelated items), which costs what the coarse passes save"}),
    "cfg2_snr60": (4, 1024, 3600, 262144, True, 0.4, "coherent", 60.0, {
        "workload": "cfg2, spectrum port wired, 60 dB SNR: near-null values are recomputed in the reference's literal form inside the scan"}),
    "cfg2_incoherent_snr60": (4, 1024, 3600, 262144, True, 0.4, "incoherent", 60.0, {
        "workload": "cfg2, spectrum port wired, emitter angles drawn per ITEM at 60 dB SNR: the product of the two unfavourable "
                    "cases (every wave's 16 items have their nulls in different bins, and the nulls need the literal form)"}),
    "cfg2_incoherent_snr60_without_spectrum_port": (4, 1024, 3600, 262144, False, 0.4, "incoherent", 60.0, {
        "workload": "the same with only ang/lvl wired"}),
    "cfg3": (8, 4096, 36000, 16384, True, 0.5, "coherent", 20.0, {
        "bound": "spectrum stores (HBM write) + int8 matrix core / level combination on the vector unit "
                 "(scan_i8_kernel; BAZ_MUSIC_EXACT=1: fp64 matrix, 2*m^2 flop per item and bin)",
        "workload": "BASELINE configs[2]: m=8 n=2 nsamples=4096 (K=512) resolution=36000, spectrum wired"}),
    "cfg3_without_spectrum_port": (8, 4096, 36000, 16384, False, 0.4, "coherent", 20.0, {
        "bound": "f16 matrix + LDS (coarse passes of the gated scan); covariance 0.11 ms",
        "workload": "BASELINE configs[2]'s shape with only ang/lvl wired (the helper's default)"}),
    "wide_m32_n2": (32, 4096, 3600, 4096, True, 0.3, "coherent", 20.0, {
        "bound": "fp64 matrix (covariance and scan on v_mfma_f64_16x16x4; EVD: signal subspace by orthogonal iteration)",
        "workload": "32 antennas (run-time-m kernels; the reference has no antenna limit), n=2, nsamples=4096 (K=128), "
                    "resolution=3600, spectrum wired, 4,096 items"}),
    "wide_m64_n2": (64, 4096, 3600, 2048, True, 0.3, "coherent", 20.0, {
        "bound": "fp64 matrix (covariance by pairs of 16-antenna blocks, scan with four staged phases per step) + EVD",
        "workload": "64 antennas (BAZ_MUSIC_MAX_M), n=2, nsamples=4096 (K=64), resolution=3600, spectrum wired, 2,048 items"}),
}
LEG_ORDER = ["cfg2_retune_in_flight"] + list(MUSIC_LEGS) + ["cfg5_chain", "cfg2_host_fed_gr37_model"]
LEG_TIMEOUT_S = {"cfg2_host_fed_gr37_model": 120}       # default 150 s: a leg takes 3 - 12 s, most of it start-up


def leg_retune_in_flight(torch, np, capi, synth, dev, stream, emit=None):
    """set_array_response (baz_music_set_table) while the HEADLINE's steps are in flight: cfg2, 262,144 items per step, port 2 wired."""
    arr, table = helper_table(np, synth, M, RES)
    batch = STREAMS_PER_GPU * ITEMS_PER_STREAM
    x = torch.cat([synth.synth_stream(torch, dev, ITEMS_PER_STREAM, M, NSAMPLES, arr, FREQUENCY, SPACING, seed=1002 + s)
                   for s in range(STREAMS_PER_GPU)], dim=0)
    ang = torch.zeros(batch, N_EMIT, dtype=torch.float32, device=dev)
    lvl = torch.zeros_like(ang)
    spec = torch.zeros(batch, RES, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    with capi.Context(M, N_EMIT, NSAMPLES, RES, table, device_id=dev.index) as ctx:
        ctx.set_stream(stream.cuda_stream)
        ctx.reserve(batch)
        step = lambda: ctx.process_device(x.data_ptr(), batch, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
        med, worst, lock = time_retunes(np, synth, ctx, table, M, RES, arr, step, stream.synchronize, 5)
        step()
        stream.synchronize()
        verified = verify_against_oracle(torch, np, x, ang, lvl, spec, table, M, N_EMIT, NSAMPLES, RES, 32)
        ctx.set_stream(None)
    return dict(verified, retune_ms=med, retune_ms_worst=worst, retune_lock_ms_worst=lock, retunes=10,
                workload="cfg2 as in the headline; baz_music_set_table alternating between two tables with two steps in flight; "
                         "afterwards a step's outputs against the oracle (the last retune restores the first table)")


def run_legs_here(names):
    """Child mode (python bench.py --extra-leg a[,b,...]): the named legs one after the other in THIS process; one JSON object per
    line on stdout, {"leg": name, ...}; a leg may print a partial object first -- the parent keeps the last line of each leg."""
    import numpy as np
    import torch
    from gr_baz_amd import capi, synth
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MUSIC-DoA path has no CPU fallback")
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    stream = torch.cuda.Stream(device=dev)
    failed = False
    for name in names:
        note("leg %s" % name)
        emit = lambda d, name=name: print(json.dumps(dict(d, leg=name)), flush=True)
        inject = os.environ.get("BAZ_BENCH_INJECT_FAULT", "")     # test hook (tests/test_bench_driver_cmd.py): "<kind>:<leg>"
        if inject.endswith(":" + name):
            kind = inject.split(":")[0]
            note("INJECTED FAULT (%s) in leg %s" % (kind, name))
            if kind == "abort":
                os.abort()                                        # what the runtime does after a GPU memory fault
            if kind == "hang":
                time.sleep(3600)
            if kind == "gpufault":                                # a REAL GPU memory fault: the covariance kernel is handed an unmapped input address
                arr, table = helper_table(np, synth, M, RES)
                with capi.Context(M, N_EMIT, NSAMPLES, RES, table, device_id=dev.index) as ctx:
                    o = torch.zeros(64, N_EMIT, dtype=torch.float32, device=dev)
                    ctx.process_device(0x10000, 64, o.data_ptr(), o.data_ptr(), None)
                    ctx.sync()
            raise RuntimeError("injected failure (%s)" % kind) if kind == "raise" else SystemExit(3)
        try:
            if name in MUSIC_LEGS:
                m, nsamples, res, batch, with_spec, seconds, scene, snr, labels = MUSIC_LEGS[name]
                out = dict(extra_music(torch, np, capi, synth, dev, stream, m, nsamples, res, batch, with_spec, seconds, scene=scene,
                                       snr_db=snr, emit=lambda d: emit(dict(d, **labels))), **labels)
            elif name == "cfg2_retune_in_flight":
                out = leg_retune_in_flight(torch, np, capi, synth, dev, stream)
            elif name == "cfg5_chain":
                out = dict(extra_cfg5(torch, np, capi, synth, dev, stream, 16384, 0.5),
                           workload="BASELINE configs[4] on one GPU: 16 antennas, fractional_resampler_cc (ratio 1.25) -> agc_cc -> "
                                    "music_doa (m16 n2 N4096 res3600), one stream")
            elif name == "cfg2_host_fed_gr37_model":
                out = host_fed_here()
            else:
                raise SystemExit("unknown leg %r (known: %s)" % (name, ", ".join(LEG_ORDER)))
        except SystemExit:
            raise
        except Exception as e:
            out = {"error": repr(e)}
            failed = True
        emit(out)
        torch.cuda.empty_cache()
    if os.environ.get("BAZ_MUSIC_GUARD", "0") not in ("", "0") and os.environ.get("BAZ_MUSIC_LAB_LIB"):
        # lab library with guard zones around every device buffer (scripts/gpu/r06a.sh): what the legs left behind
        note("guard zones damaged over these legs: %d" % capi.guard_check(lab=True))
    return 1 if failed else 0


def host_fed_here():
    """scripts/hostfed_extra.py (its own interpreter: it needs the pybind module and none of this file's state)."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "hostfed_extra.py")], capture_output=True, text=True, cwd=ROOT)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": "rc %d: %s" % (r.returncode, (r.stderr or r.stdout).strip()[-300:])}
    return json.loads(lines[-1])


def run_leg_subprocess(name, timeout_s=None):
    """One leg in its own interpreter with a time limit.  Returns the last object the leg printed; on a crash / hang / bad exit the
    object carries "error" (next to whatever partial figures the leg had printed before)."""
    import signal
    import subprocess
    timeout_s = timeout_s or int(os.environ.get("BAZ_BENCH_LEG_TIMEOUT_S", "0")) or LEG_TIMEOUT_S.get(name, 150)
    cmd = [sys.executable, os.path.abspath(__file__), "--extra-leg", name]
    t0 = time.perf_counter()
    try:
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, start_new_session=True)
    except Exception as e:
        return {"error": "could not start the leg: %r" % (e,)}
    try:
        so, se = p.communicate(timeout=timeout_s)
        err = None if p.returncode == 0 else "rc %d" % p.returncode
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)             # the leg's own process group (start_new_session), nothing else
        except Exception:
            p.kill()
        so, se = p.communicate()
        err = "timed out after %d s" % timeout_s
    out = None
    for l in so.splitlines():
        if l.startswith("{"):
            try:
                d = json.loads(l)
                if d.get("leg") == name:
                    out = d
            except Exception:
                pass
    if out is None:
        out = {}
        err = err or "no result line"
    out.pop("leg", None)
    if err and "error" not in out:
        tail = " | ".join(x for x in se.strip().splitlines()[-3:] if "amdgpu.ids" not in x)
        out["error"] = "%s: %s" % (err, tail[-400:])
    elif not err:
        out.pop("partial", None)
    out["leg_wall_s"] = round(time.perf_counter() - t0, 2)
    return out


def supervise():
    """Plain `python bench.py` with one GPU: the measurement runs in a CHILD of this process (same command line); this process holds no GPU context and
    cannot be taken down by a GPU fault.  It passes exactly ONE JSON line on to stdout -- the contract's "rank 0 prints ONE JSON line" --: the LAST one the
    child printed.  The child prints its complete line (metric, roofline, cpu_baseline, verification) before the secondary legs start and the same line with
    the legs' figures after them; should it die in between, the first is what stdout gets.  If the child is killed by a signal before it has printed any
    line -- the HIP runtime abort()s a process whose GPU work hits a memory fault, which is what took the driver's round-5 run -- it is
    started ONCE more, and the line of the second attempt says so (config.headline_attempt = 2, config.headline_previous_failure).  Nothing is retried after
    an orderly exit (wrong arguments, no GPU, a failed verification) or once a line exists.  BAZ_BENCH_SUPERVISE=0 runs everything in this process."""
    import signal
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    state = {"latest": None, "out": False}

    def put_line():
        if state["latest"] is not None and not state["out"]:
            state["out"] = True
            sys.stdout.write(state["latest"])
            sys.stdout.flush()

    def on_term(signum, frame):          # (the driver's time limit: hand over what there is)
        put_line()
        os._exit(128 + signum)
    for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        try:
            signal.signal(sig, on_term)
        except Exception:
            pass
    previous = ""
    for attempt in (1, 2):
        env = dict(os.environ, BAZ_BENCH_CHILD="1", BAZ_BENCH_ATTEMPT=str(attempt), BAZ_BENCH_PREVIOUS_FAILURE=previous)
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, bufsize=1, env=env)        # stderr: inherited
        for ln in p.stdout:
            if ln.startswith("{"):
                state["latest"] = ln if ln.endswith("\n") else ln + "\n"
            else:
                sys.stdout.write(ln)
                sys.stdout.flush()
        rc = p.wait()
        if rc >= 0 or state["latest"] is not None or attempt == 2:
            if rc < 0 and state["latest"] is not None:
                note("the measuring process was killed by signal %d after its complete line and before the legs' figures: that line is the one printed" % (-rc))
            put_line()
            return rc if rc >= 0 else 128 - rc           # (a signal death: the shell's convention, e.g. 134 for SIGABRT)
        previous = "attempt 1 was killed by signal %d before it printed its line" % (-rc)
        note("the measuring process was killed by signal %d before it printed anything; starting it once more" % (-rc))
    return 1


def self_launch(n):
    """`python bench.py --gpus N` without a launcher around it: re-run this file as N ranks (one process per GPU) under
    torch.distributed.run on 127.0.0.1 -- exactly the command line the driver uses -- and hand its exit code back.  Fails
    loudly when fewer than N devices are visible (unless the BAZ_BENCH_SHARE_DEVICES=1 test hook is set)."""
    import socket
    import subprocess
    import torch
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < n and os.environ.get("BAZ_BENCH_SHARE_DEVICES") != "1" and "--dry-ranks" not in sys.argv:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (one process per GPU)" % (n, ndev))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "1"))
    raise SystemExit(subprocess.call(cmd, env=env))


def dry_ranks_main(args):
    """--dry-ranks: the launcher, the rank environment, the dealing of the streams, the barrier / max-over-ranks clock and the
    gathered rank records of an N-rank run WITHOUT any GPU work -- a rehearsal of the SCALE day on a box that has no (or not
    enough) GPUs (tests/test_sharding.py feeds it 8 ranks).  A step is a sleep of the time the ranks' items would take at
    2.5e8 items/s; the barrier / clock ask for RCCL exactly like the real run, fall back to gloo where RCCL cannot initialise,
    and the line says so.  "dry_run": true -- the value is NOT a measurement."""
    from gr_baz_amd import sharding
    rank, local_rank, world = sharding.dist_env()
    if world != max(1, args.gpus):
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    strong = args.scaling == "strong"
    if strong and STRONG_STREAMS % world:
        raise SystemExit("--scaling strong deals %d streams: --gpus must divide it" % STRONG_STREAMS)
    active = sharding.init_process_group(use_gpu=False, local_rank=local_rank, try_nccl=True)
    # where RCCL does come up (an 8-GPU box) the barrier / clock tensors must live on the device: an NCCL-only group cannot
    # reduce CPU tensors; after the gloo fall-back they stay on the host
    on_dev = bool(active and sharding.backend_info()["backend"] == "nccl")
    n_streams = STRONG_STREAMS if strong else STREAMS_PER_GPU * world
    mine = sharding.streams_of_rank(n_streams, world, rank)
    batch = len(mine) * ITEMS_PER_STREAM
    group_items = min(len(mine), STREAMS_PER_GPU) * ITEMS_PER_STREAM
    n_groups = batch // group_items
    step = lambda: time.sleep(batch / 2.5e8)
    for _ in range(args.warmup):
        step()
    rounds, timed_total = [], 0.0
    while True:
        sharding.barrier(active, on_dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        elapsed = time.perf_counter() - t0
        sharding.barrier(active, on_dev)
        tmax = sharding.max_over_ranks(elapsed, active, on_dev)
        rounds.append(tmax)
        timed_total += tmax
        if timed_total >= args.min_seconds or len(rounds) >= 200:
            break
    t_med = statistics.median(rounds)
    total_items = sharding.sum_over_ranks(float(batch * args.steps), active, on_dev)
    ranks = [{"rank": rank, "items_per_step": batch, "streams": mine, "device": "none (dry run)"}]
    if active:
        import torch.distributed as dist
        gathered = [None] * world
        dist.all_gather_object(gathered, ranks[0])
        ranks = gathered
    assert len(ranks) == world and sorted(r["rank"] for r in ranks) == list(range(world)), ranks
    if rank == 0:
        info = sharding.backend_info() if active else {"backend": None, "requested": None, "fell_back": False, "fallback_reason": None}
        print(json.dumps({
            "metric": "MUSIC-DoA snapshots/s (4 ant, 1024 samp, 3600 bins)", "dry_run": True,
            "value": total_items / t_med, "unit": "snapshots/s (SIMULATED steps: launcher rehearsal, not a measurement)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": t_med / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "none",
            "config": {"workload": "dry run of cfg2's dealing: %d streams x %d items per rank per step, no device work"
                                   % (len(mine), ITEMS_PER_STREAM),
                       "items_per_gpu_per_step": batch, "items_per_step_all_gpus": int(total_items / args.steps + 0.5),
                       "streams_total": n_streams, "launch_sequences_per_step": n_groups, "items_per_launch_sequence": group_items,
                       "parallelism": "independent streams, s mod %d, no collective" % world,
                       "collective_backend_for_barrier_and_clock": info["backend"],
                       "collective_backend_requested": info["requested"],
                       "collective_backend_fell_back": info["fell_back"],
                       "collective_backend_fallback_reason": info["fallback_reason"],
                       "ranks": ranks}}), flush=True)
    if active:
        import torch.distributed as dist
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scaling", choices=("weak", "strong"), default=os.environ.get("BAZ_BENCH_SCALING", "weak"),
                    help="weak: 8 streams per GPU; strong: BASELINE configs[3], the same 64 streams dealt s mod N at every N")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary cfg2-no-spectrum / cfg3 / cfg5 measurements")
    ap.add_argument("--ramp-seconds", type=float, default=0.25, help="untimed steady load before warm-up (clock ramp)")
    ap.add_argument("--min-seconds", type=float, default=0.5, help="repeat the K-step timed region until this much timed work")
    ap.add_argument("--dry-ranks", action="store_true",
                    help="launcher rehearsal without GPU work: N ranks, the dealing, barrier / clock and rank records only")
    ap.add_argument("--extra-leg", default=None, metavar="NAME[,NAME...]",
                    help="child mode: run the named secondary legs in this process (ALL = every leg, in order) and print one JSON object each")
    ap.add_argument("--legs", default=None, metavar="NAME[,NAME...]", help="run only these secondary legs (default: all)")
    args = ap.parse_args()

    if args.extra_leg:
        names = LEG_ORDER if args.extra_leg == "ALL" else [n for n in args.extra_leg.split(",") if n]
        raise SystemExit(run_legs_here(names))

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        return self_launch(args.gpus)                        # plain `python bench.py --gpus N`: start the N ranks ourselves
    if args.dry_ranks:
        return dry_ranks_main(args)
    if (args.gpus <= 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ and os.environ.get("BAZ_BENCH_CHILD") != "1"
            and os.environ.get("BAZ_BENCH_SUPERVISE", "1") != "0"):
        raise SystemExit(supervise())                        # the measurement in a child of this process (see supervise)
    inject = os.environ.get("BAZ_BENCH_INJECT_FAULT", "")    # test hook: "abort:headline@1" = the first attempt's measuring process dies
    if inject == "abort:headline@%s" % os.environ.get("BAZ_BENCH_ATTEMPT", "1"):
        note("INJECTED FAULT: the measuring process abort()s before its line")
        os.abort()

    import numpy as np
    import torch
    from gr_baz_amd import capi, sharding, synth

    rank, local_rank, world = sharding.dist_env()
    if world != max(1, args.gpus):                           # never print a line whose n_gpus is not what was asked for
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if args.scaling == "strong" and STRONG_STREAMS % world:
        raise SystemExit("--scaling strong deals %d streams: --gpus must divide it" % STRONG_STREAMS)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MUSIC-DoA path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if world > ndev:
        if os.environ.get("BAZ_BENCH_SHARE_DEVICES") != "1":
            raise SystemExit("%d ranks but only %d GPU(s) visible (one process per GPU)" % (world, ndev))
        # test hook: several ranks on one GPU (exercises the N > 1 code path on a 1-GPU box); EVERY rank switches the
        # barrier / clock backend, RCCL cannot put two ranks on one device
        local_rank %= ndev
        os.environ["BAZ_BENCH_BACKEND"] = "gloo"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    active = sharding.init_process_group(use_gpu=True, local_rank=local_rank)
    binfo = sharding.backend_info() if active else {"backend": None, "requested": None, "fell_back": False, "fallback_reason": None}
    backend = binfo["backend"]

    arr, table = helper_table(np, synth, M, RES)

    # this rank's streams: global stream s lives on rank s mod world (config 4), seed = 1002 + s
    strong = args.scaling == "strong"
    n_streams = STRONG_STREAMS if strong else STREAMS_PER_GPU * world
    mine = sharding.streams_of_rank(n_streams, world, rank)
    batch = len(mine) * ITEMS_PER_STREAM
    # a launch sequence covers a group of up to 8 streams (262,144 items: the weak mode's whole step); the strong mode's
    # step walks the rank's groups one after the other on the same stream
    group_items = min(len(mine), STREAMS_PER_GPU) * ITEMS_PER_STREAM
    assert batch % group_items == 0
    n_groups = batch // group_items
    x = torch.empty(batch, 2 * NSAMPLES, dtype=torch.float32, device=dev)
    for i, s_id in enumerate(mine):
        x[i * ITEMS_PER_STREAM:(i + 1) * ITEMS_PER_STREAM] = synth.synth_stream(
            torch, dev, ITEMS_PER_STREAM, M, NSAMPLES, arr, FREQUENCY, SPACING, seed=1002 + s_id).reshape(ITEMS_PER_STREAM, -1)
    ang = torch.zeros(batch, N_EMIT, dtype=torch.float32, device=dev)
    lvl = torch.zeros_like(ang)
    spec = torch.zeros(batch, RES, dtype=torch.float32, device=dev)

    # One explicit stream for everything the engine does (ordered against the fills above by the synchronize below)
    stream = torch.cuda.Stream(device=dev)
    ctx = capi.Context(M, N_EMIT, NSAMPLES, RES, table, device_id=local_rank)
    ctx.set_stream(stream.cuda_stream)
    ctx.reserve(group_items)
    torch.cuda.synchronize()
    xb, ab, lb, sb = x.data_ptr(), ang.data_ptr(), lvl.data_ptr(), spec.data_ptr()
    x_row, al_row, sp_row = 8 * NSAMPLES, 4 * N_EMIT, 4 * RES            # bytes per item

    def step():
        for gi in range(n_groups):
            o = gi * group_items
            ctx.process_device(xb + o * x_row, group_items, ab + o * al_row, lb + o * al_row, sb + o * sp_row)

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

    # Timed rounds: each is EXACTLY args.steps steps between barrier + synchronize on both sides, max over ranks.
    ctx.profile(int(os.environ.get("BAZ_BENCH_PROFILE", "2")))   # 2: hipEvents around the dominant kernel only
    rounds, scan_ms_total, scan_launches, timed_total = [], 0.0, 0, 0.0
    while True:
        sharding.barrier(active, True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        sharding.barrier(active, True)
        tmax = sharding.max_over_ranks(elapsed, active, True)
        rounds.append(tmax)
        timed_total += tmax
        sm, sn = ctx.stage_ms(capi.STAGE_SCAN)            # cumulative since profile(): dominant kernel, timed rounds only
        scan_ms_total, scan_launches = sm, sn
        if timed_total >= args.min_seconds or len(rounds) >= 200:
            break
    ctx.profile(False)
    # informational per-stage breakdown from a separate short pass (events around every kernel add launch gaps,
    # so they stay out of the timed region)
    ctx.profile(1)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    stage = [ctx.stage_ms(s) for s in range(capi.NUM_STAGES)]
    ctx.profile(False)
    cov_name = ctx.stage_name(capi.STAGE_COV)
    scan_name = ctx.stage_name(capi.STAGE_SCAN)
    bpi = ctx.bytes_per_item(True)
    # outside the timed region: what the last timed step left in the output buffers against the CPU oracle, then the cost of
    # a retune (set_array_response) while batches are in flight
    note("headline timed (%d rounds of %d steps); checking the last step's outputs against the oracle" % (len(rounds), args.steps))
    verified = verify_against_oracle(torch, np, x, ang, lvl, spec, table, M, N_EMIT, NSAMPLES, RES, 256)
    # (the cost of a retune beside the headline's steps is measured by the leg cfg2_retune_in_flight, in its own process)
    ctx.set_stream(None)
    ctx.close()
    # What this box's memory system takes for a plain write of the same 3.78 GB (a torch fill of the spectrum buffer; hipEvents on the bench
    # stream, outside the timed region, AFTER the outputs have been checked: it overwrites them): the dominant kernel is bound by its spectrum stores, and boxes of this pool differ by 20 % in exactly
    # that -- the roofline fraction against the 8 TB/s peak does not say how far the kernel is from what can be had.
    write_ceiling_gbs = None
    try:
        sp_one = spec[:group_items]
        with torch.cuda.stream(stream):
            for _ in range(3):
                sp_one.fill_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(10):
                sp_one.fill_(1.0)
            e1.record(stream)
        stream.synchronize()
        write_ceiling_gbs = sp_one.numel() * 4 * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    except Exception:
        pass

    t_med = statistics.median(rounds)
    total_items = sharding.sum_over_ranks(float(batch * args.steps), active, True)
    value = total_items / t_med
    ranks = [{"rank": rank, "items_per_step": batch, "streams": mine, "device": torch.cuda.get_device_name(local_rank),
              "verified_items": verified["verified_items"], "verified_max_rel_err": verified["verified_max_rel_err"],
              "verified_bins_identical": verified["verified_bins_identical"], "verified_ok": verified["verified_ok"]}]
    if active:
        import torch.distributed as dist
        gathered = [None] * world
        dist.all_gather_object(gathered, ranks[0])
        ranks = gathered
    assert len(ranks) == world and sorted(r["rank"] for r in ranks) == list(range(world)), ranks

    if rank == 0:
        scan_avg_s = scan_ms_total / max(scan_launches, 1) * 1e-3
        scan_bytes = (4 * RES + 8 * N_EMIT) * group_items           # spectrum + ang/lvl-equivalent written per launch
        achieved = scan_bytes / scan_avg_s / 1e9 if scan_avg_s > 0 else 0.0
        src_sha = kernel_sources_sha()
        traffic, traffic_stale, traffic_src = None, None, None
        if os.path.exists(TRAFFIC_PROFILE):
            try:
                tinfo = json.load(open(TRAFFIC_PROFILE))
                traffic_src = os.path.relpath(TRAFFIC_PROFILE, ROOT)
                if tinfo.get("kernel_sources_sha") == src_sha and tinfo.get("items_per_launch") == group_items:
                    traffic, traffic_stale = tinfo.get("scan_hbm_bytes_per_launch"), False
                else:
                    traffic_stale = True                            # profile of other code / launch size: not reported
            except Exception:
                traffic_stale = True
        cov_s = stage[capi.STAGE_COV][0] / max(stage[capi.STAGE_COV][1], 1) * 1e-3
        cov_tf = 8.0 * M * NSAMPLES * group_items / cov_s / 1e12 if cov_s > 0 else 0.0
        x4 = "cov4_" in cov_name
        fused = "cov4_evd" in cov_name
        cov_mfma = {"kernel": cov_name,
                    "kernel_also_does": "the batched 4x4 Hermitian EVD of the same items: ~0.08 ms of fp64 VALU phases during "
                                        "which HBM idles (DESIGN.md 5.1); the rates below divide by the WHOLE kernel's "
                                        "time, the stream alone runs at ~7 TB/s" if fused else None, "useful_tflops": cov_tf, "fp64_matrix_peak_tflops": FP64_MFMA_PEAK_TF,
                    "frac_of_peak": cov_tf / FP64_MFMA_PEAK_TF, "issued_over_useful": 4.0 / 3.0 if x4 else 2.0,
                    "hbm_read_GBs": 8.0 * NSAMPLES * group_items / cov_s / 1e9 if cov_s > 0 else 0.0,
                    "hbm_read_frac_of_8TBs": 8.0 * NSAMPLES * group_items / cov_s / 1e9 / HBM_PEAK_GBS if cov_s > 0 else 0.0,
                    "note": ("v_mfma_f64_4x4x4_4b blocks: X0X0^T, X0X1^T, X1X1^T + one transposed duplicate per pair of "
                             "instructions; dwordx4 input stream; HBM-read bound while it streams") if x4 else
                            ("16x16x4 fp64 MFMA tiles hold 2 items block-diagonally at m=4 (half the issued flops are "
                             "structural zeros); the kernel is HBM-read bound")}
        per_round = [t / args.steps * 1e3 for t in rounds]
        line = {
            "metric": "MUSIC-DoA snapshots/s (4 ant, 1024 samp, 3600 bins)",
            "value": value, "unit": "snapshots/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": t_med / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "rounds": {"count": len(rounds), "steps_each": args.steps, "statistic": "median",
                       "ms_per_step_min": min(per_round), "ms_per_step_max": max(per_round),
                       "timed_seconds_total": timed_total},
            "kernel_sources_sha": src_sha,
            "config": {"workload": "cfg2 (BASELINE.json configs[1]): m=4 n=2 nsamples=1024 (K=256) resolution=3600, "
                                   "spectrum port wired, %d streams x %d items per GPU per step, device-resident%s"
                                   % (len(mine), ITEMS_PER_STREAM,
                                      " (configs[3]: the same %d streams at every GPU count, dealt s mod N)" % STRONG_STREAMS if strong else ""),
                       "items_per_gpu_per_step": batch, "items_per_step_all_gpus": int(total_items / args.steps + 0.5),
                       "streams_total": n_streams, "launch_sequences_per_step": n_groups, "items_per_launch_sequence": group_items,
                       "parallelism": "independent streams, s mod %d, no collective" % world,
                       "collective_backend_for_barrier_and_clock": backend,
                       "collective_backend_requested": binfo["requested"], "collective_backend_fell_back": binfo["fell_back"],
                       "collective_backend_fallback_reason": binfo["fallback_reason"], "ranks": ranks,
                       "algorithmic_bytes_per_item": bpi,
                       "headline_attempt": int(os.environ.get("BAZ_BENCH_ATTEMPT", "1")),
                       "headline_previous_failure": os.environ.get("BAZ_BENCH_PREVIOUS_FAILURE") or None,
                       # every rank checked a sample of what its last timed step wrote against the CPU oracle (outside the timed region)
                       "verified_items": sum(r["verified_items"] for r in ranks),
                       "verified_max_rel_err": max(r["verified_max_rel_err"] for r in ranks),
                       "verified_bins_identical": all(r["verified_bins_identical"] for r in ranks),
                       "verified_ok": all(r["verified_ok"] for r in ranks),
                       "verified_tolerance": VERIFY_TOL, "verified_against": verified["verified_against"],
                       "pipeline_hbm_fraction_of_8TBs": value / world * bpi / 8e12,
                       "stage_ms_per_launch_separate_pass": {nm: stage[s][0] / max(stage[s][1], 1)
                                               for s, nm in enumerate(("cov (+ evd when fused)", "evd_proj", "scan_mfma", "topn_merge"))},
                       # the one dense contraction (north_star): useful fp64 flops 8*m*N per item against the fp64
                       # matrix peak; rocprofv3 MFMA-busy for the same kernel is in profiles/r04_bench_pmc_summary.txt
                       "covariance_mfma": cov_mfma},
            "roofline": {"bound": "hbm", "kernel": scan_name.split("::")[-1].split("<")[0], "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_stale": traffic_stale, "traffic_profile": traffic_src,
                         "algorithmic_bytes_per_launch": scan_bytes, "avg_launch_ms": scan_avg_s * 1e3,
                         "launches": scan_launches,
                         # the same bytes as a plain fill on this box in this run (context, not a peak): see write_ceiling above
                         "plain_fill_of_the_same_bytes_GBs": write_ceiling_gbs,
                         "achieved_over_plain_fill": (achieved / write_ceiling_gbs) if write_ceiling_gbs else None},
        }
        cfgd = line["config"]
        if world == 1 and not args.no_cpu_baseline:
            note("cpu baseline (oracle on the host cores)")
            try:
                line["cpu_baseline"] = cpu_baseline(table)
            except Exception as e:               # (never seen; the line still goes out, and says why the object is missing)
                line["cpu_baseline"] = {"value": None, "unit": "snapshots/s", "cores": 0, "kind": "port", "sample": "failed", "error": repr(e)}
        legs = []
        if world == 1 and not args.no_extras:
            legs = [n for n in (args.legs.split(",") if args.legs else LEG_ORDER) if n]
            unknown = [n for n in legs if n not in LEG_ORDER]
            if unknown:
                raise SystemExit("unknown leg(s) %s (known: %s)" % (unknown, ", ".join(LEG_ORDER)))
        cfgd["extras_pending"] = len(legs)
        # THE LINE, complete (metric, roofline, cpu_baseline, verified_*), before any secondary leg runs: whatever happens below cannot take it
        # back.  Without legs it is the one line of the contract.  With legs to run, the same line follows once more, enriched with their figures,
        # and exactly ONE of the two reaches the caller's stdout: supervised (plain `python bench.py`, see supervise()) both go to the parent, which
        # prints the last one it has got; unsupervised the early copy goes to stderr only.
        if not legs or os.environ.get("BAZ_BENCH_CHILD") == "1":
            print(json.dumps(line), flush=True)
        if legs:
            print("bench.py: [%.1f s] complete line before the legs: %s" % (time.perf_counter() - T_START, json.dumps(line)), file=sys.stderr, flush=True)
            if os.environ.get("BAZ_BENCH_INJECT_FAULT", "") == "abort:after_first_line":      # test hook
                note("INJECTED FAULT: the measuring process abort()s after its complete line, before the legs")
                os.abort()
        if legs:
            # this process keeps nothing on the device while the legs run (they are sized for a whole GPU)
            del x, spec, ang, lvl
            torch.cuda.empty_cache()
            extra, failed_legs = {}, []
            for name in legs:
                note("leg %s (own process)" % name)
                extra[name] = run_leg_subprocess(name)
                if "error" in extra[name]:
                    failed_legs.append(name)
                    note("leg %s FAILED: %s" % (name, extra[name]["error"]))
            # the secondary claims as SCALAR keys of config (a driver that keeps only scalar config keys still carries them;
            # the dicts they come from follow under "extra")
            def pick(name, *path):
                v = extra.get(name, {})
                for k in path:
                    v = v.get(k) if isinstance(v, dict) else None
                return v
            cfgd["retune_ms"] = pick("cfg2_retune_in_flight", "retune_ms")
            cfgd["retune_ms_worst"] = pick("cfg2_retune_in_flight", "retune_ms_worst")
            cfgd["retune_lock_ms_worst"] = pick("cfg2_retune_in_flight", "retune_lock_ms_worst")
            cfgd["default_wiring_snapshots_per_s"] = pick("cfg2_without_spectrum_port", "snapshots_per_s")
            cfgd["default_wiring_hbm_read_frac"] = pick("cfg2_without_spectrum_port", "hbm_read_fraction_of_8TBs")
            cfgd["incoherent_snapshots_per_s"] = pick("cfg2_incoherent_scene", "snapshots_per_s")
            cfgd["incoherent_default_wiring_snapshots_per_s"] = pick("cfg2_incoherent_scene_without_spectrum_port", "snapshots_per_s")
            cfgd["snr60_snapshots_per_s"] = pick("cfg2_snr60", "snapshots_per_s")
            cfgd["incoherent_snr60_snapshots_per_s"] = pick("cfg2_incoherent_snr60", "snapshots_per_s")
            cfgd["incoherent_snr60_default_wiring_snapshots_per_s"] = pick("cfg2_incoherent_snr60_without_spectrum_port", "snapshots_per_s")
            cfgd["cfg3_snapshots_per_s"] = pick("cfg3", "snapshots_per_s")
            cfgd["cfg3_verified_items"] = pick("cfg3", "verified_items")
            cfgd["cfg3_verified_max_rel_err"] = pick("cfg3", "verified_max_rel_err")
            cfgd["cfg3_verified_bins_identical"] = pick("cfg3", "verified_bins_identical")
            cfgd["cfg3_retune_ms"] = pick("cfg3", "retune_ms")
            cfgd["cfg3_scan_int8_tops_first_tier"] = pick("cfg3", "scan_int8_tops_first_tier")
            cfgd["cfg3_scan_frac_of_int8_matrix_peak"] = pick("cfg3", "scan_frac_of_int8_matrix_peak_5000TOPS")
            cfgd["cfg3_scan_ms"] = pick("cfg3", "stage_ms_per_launch", "scan")
            cfgd["cfg3_scan_frac_of_hbm_8TBs"] = pick("cfg3", "scan_frac_of_hbm_8TBs")
            cfgd["cfg3_pipeline_hbm_frac"] = pick("cfg3", "pipeline_hbm_fraction_of_8TBs")
            cfgd["cfg3_default_wiring_snapshots_per_s"] = pick("cfg3_without_spectrum_port", "snapshots_per_s")
            cfgd["wide_m32_snapshots_per_s"] = pick("wide_m32_n2", "snapshots_per_s")
            cfgd["wide_m64_snapshots_per_s"] = pick("wide_m64_n2", "snapshots_per_s")
            cfgd["cfg5_chain_snapshots_per_s"] = pick("cfg5_chain", "snapshots_per_s")
            cfgd["cfg5_chain_hbm_frac"] = pick("cfg5_chain", "chain_hbm_fraction_of_8TBs")
            cfgd["cfg5_chain_music_scan_ms"] = pick("cfg5_chain", "music_stage_ms", "scan")
            cfgd["host_fed_pinned_with_port2_items_per_s"] = pick("cfg2_host_fed_gr37_model", "runs", "with_spectrum_port_page_locked_buffers", "items_per_s")
            cfgd["extras_all_verified_ok"] = all(v.get("verified_ok", True) for v in extra.values() if isinstance(v, dict))
            cfgd["extras_run"] = len(legs)
            cfgd["extras_failed"] = len(failed_legs)
            cfgd["extras_failed_legs"] = ",".join(failed_legs)
            cfgd["extras_pending"] = 0
            cfgd["extra"] = extra
            print(json.dumps(line), flush=True)
        failed = not cfgd["verified_ok"]           # the exit code speaks for the HEADLINE only (legs report through extras_failed / extras_all_verified_ok)
    else:
        failed = False
    if active:
        import torch.distributed as dist
        dist.destroy_process_group()
    if failed:
        raise SystemExit("bench.py: the timed outputs do NOT agree with the CPU oracle within %g (see verified_* in the line)" % VERIFY_TOL)


if __name__ == "__main__":
    main()
