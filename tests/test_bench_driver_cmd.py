"""What is graded, run as graded: the driver's exact command line

    python bench.py --gpus 1 --steps 20 --warmup 5

must exit 0 and leave on stdout exactly ONE JSON object (the contract) with BASELINE.json's metric, a `roofline` whose fraction lies in
(0, 1), a `cpu_baseline` with a positive value, the in-run verification green and no failed secondary leg.  Round 5's driver run died
with a GPU memory fault inside a secondary leg before anything had been printed; since round 6 the measuring process hands its complete
line to a supervising parent (and to stderr) BEFORE the secondary legs start -- each of which runs in its own process -- and the same
line, enriched, after them; the parent prints the last one it has got.  The second half of this file injects faults into legs, into the
measuring process before its first line and after it, and checks that a complete line comes out every time."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASELINE = json.load(open(os.path.join(ROOT, "BASELINE.json")))
DRIVER_ARGV = ["--gpus", "1", "--steps", "20", "--warmup", "5"]


def _run(argv, env=None, timeout=900):
    e = dict(os.environ, **(env or {}))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py"] + argv, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=e)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    return r, lines


def _early_line(r):
    """The complete line the measuring process put out before the legs (stderr copy)."""
    tag = "complete line before the legs: "
    got = [l.split(tag, 1)[1] for l in r.stderr.splitlines() if tag in l]
    assert len(got) == 1, "expected one early line on stderr, found %d" % len(got)
    return json.loads(got[0])


def _check_complete(line, steps=20, warmup=5):
    assert BASELINE["metric"].startswith(line["metric"]), line["metric"]          # "MUSIC-DoA snapshots/s (4 ant, 1024 samp, 3600 bins)"
    assert line["unit"] == "snapshots/s" and line["n_gpus"] == 1 and line["steps"] == steps and line["warmup"] == warmup
    assert line["higher_is_better"] is True and line["vs_baseline"] is None and line["dtype"] == "f64" and line["data"] == "synthetic"
    assert line["value"] > 1e7                                                    # north_star's floor; the measured rate is ~2.3e8
    assert abs(line["value"] - line["config"]["items_per_step_all_gpus"] / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    assert "cfg2" in line["config"]["workload"] and "model" not in line["config"]
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert 0.0 < rf["frac"] < 1.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert rf["launches"] >= steps and rf["avg_launch_ms"] > 0
    cb = line["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("reference", "port") and cb["unit"] == "snapshots/s" and cb["sample"]
    c = line["config"]
    assert c["verified_ok"] is True and c["verified_items"] >= 256 and c["verified_max_rel_err"] <= 1e-5 and c["verified_bins_identical"]
    assert c["headline_attempt"] in (1, 2)


@pytest.mark.gpu
@pytest.mark.timeout(1200)
def test_the_drivers_exact_command(gpu_device):
    r, lines = _run(DRIVER_ARGV)
    assert r.returncode == 0, "rc %d\n%s" % (r.returncode, r.stderr[-3000:])
    assert len(lines) == 1 and len(r.stdout.strip().splitlines()) == 1, "stdout must hold ONE JSON line, got %d" % len(lines)
    first, last = _early_line(r), lines[0]
    assert first["config"]["headline_attempt"] == 1 and first["config"]["headline_previous_failure"] is None
    _check_complete(first)
    _check_complete(last)
    assert first["value"] == last["value"] and first["roofline"] == last["roofline"]
    assert first["config"]["extras_pending"] > 0 and "extra" not in first["config"]
    c = last["config"]
    assert c["extras_pending"] == 0 and c["extras_run"] == len(c["extra"]) >= 13
    failed = {k: v["error"] for k, v in c["extra"].items() if "error" in v}
    assert not failed and c["extras_failed"] == 0 and c["extras_failed_legs"] == "", failed
    assert c["extras_all_verified_ok"] is True
    for k in ("default_wiring_snapshots_per_s", "incoherent_snapshots_per_s", "cfg3_snapshots_per_s", "cfg5_chain_snapshots_per_s",
              "wide_m64_snapshots_per_s", "retune_ms", "host_fed_pinned_with_port2_items_per_s"):
        assert c[k] and c[k] > 0, k
    print("\ndriver command: value %.4g snapshots/s, %.3f ms/step, roofline.frac %.3f, cpu_baseline %.3g on %d cores (%s); %d legs, %.0f s of legs"
          % (last["value"], last["ms_per_step"], last["roofline"]["frac"], last["cpu_baseline"]["value"], last["cpu_baseline"]["cores"],
             last["cpu_baseline"]["kind"], c["extras_run"], sum(v.get("leg_wall_s", 0) for v in c["extra"].values())))


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("kind", ["abort", "raise", "hang", "gpufault"])
def test_a_fault_in_a_leg_cannot_take_the_line(gpu_device, kind):
    """abort = what the HIP runtime does to the process after a GPU memory fault (SIGABRT); gpufault = a real one (the covariance kernel
    of a leg's child process is handed an unmapped address); hang = a leg that never returns (cut off by its time limit); raise = a
    Python exception.  Every time: rc 0, both lines complete, the failed leg named, the legs after it still measured."""
    if kind == "gpufault" and os.environ.get("BAZ_TEST_REAL_GPU_FAULT") != "1":
        pytest.skip("a real GPU memory fault is provoked only on request (BAZ_TEST_REAL_GPU_FAULT=1; profiles/r06_fault_injection.txt has a run)")
    legs = "cfg2_without_spectrum_port,wide_m64_n2,cfg2_snr60"
    env = {"BAZ_BENCH_INJECT_FAULT": "%s:wide_m64_n2" % kind}
    if kind == "hang":
        env["BAZ_BENCH_LEG_TIMEOUT_S"] = "45"      # (also the limit of the two healthy legs: 3 - 6 s each once the interpreter start is warm)
    r, lines = _run(DRIVER_ARGV + ["--legs", legs], env=env)
    assert r.returncode == 0, "rc %d\n%s" % (r.returncode, r.stderr[-3000:])
    assert len(lines) == 1
    _check_complete(_early_line(r))
    _check_complete(lines[0])
    c = lines[0]["config"]
    assert c["extras_run"] == 3 and c["extras_failed"] == 1 and c["extras_failed_legs"] == "wide_m64_n2"
    assert "error" in c["extra"]["wide_m64_n2"]
    if kind == "hang":
        assert "timed out" in c["extra"]["wide_m64_n2"]["error"]
    if kind == "gpufault":
        assert "rc -6" in c["extra"]["wide_m64_n2"]["error"] or "Memory access fault" in c["extra"]["wide_m64_n2"]["error"], c["extra"]["wide_m64_n2"]
    for ok_leg in ("cfg2_without_spectrum_port", "cfg2_snr60"):                    # the leg before and the leg AFTER the fault
        assert "error" not in c["extra"][ok_leg] and c["extra"][ok_leg]["snapshots_per_s"] > 1e7 and c["extra"][ok_leg]["verified_ok"]
    assert c["default_wiring_snapshots_per_s"] > 1e7 and c["snr60_snapshots_per_s"] > 1e7 and c["wide_m64_snapshots_per_s"] is None


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_a_fault_in_the_measuring_process_itself_costs_one_restart(gpu_device):
    """`python bench.py` measures in a child of itself.  The child of the first attempt abort()s before its line (what a GPU memory fault does to a
    process): it is started once more, rc 0, the line is complete and says that it is the second attempt's."""
    r, lines = _run(DRIVER_ARGV + ["--legs", "cfg2_snr60"], env={"BAZ_BENCH_INJECT_FAULT": "abort:headline@1"})
    assert r.returncode == 0, "rc %d\n%s" % (r.returncode, r.stderr[-3000:])
    assert len(lines) == 1
    _check_complete(lines[0])
    assert lines[0]["config"]["headline_attempt"] == 2 and "signal 6" in lines[0]["config"]["headline_previous_failure"]
    assert lines[0]["config"]["extras_run"] == 1 and lines[0]["config"]["extras_failed"] == 0
    assert "starting it once more" in r.stderr
    # the measuring process dies AFTER its complete line and before the legs' figures: that line is the one line of stdout (and the exit code tells)
    r, lines = _run(DRIVER_ARGV + ["--legs", "cfg2_snr60"], env={"BAZ_BENCH_INJECT_FAULT": "abort:after_first_line"})
    assert r.returncode == 134 and len(lines) == 1, (r.returncode, len(lines))
    _check_complete(lines[0])
    assert lines[0]["config"]["extras_pending"] == 1 and "extra" not in lines[0]["config"] and lines[0] == _early_line(r)
    # ... and an undisturbed run says attempt 1 (asserted on the driver's command above through _check_complete's caller); a fault in BOTH attempts is an error
    r, lines = _run(DRIVER_ARGV + ["--no-extras", "--no-cpu-baseline"], env={"BAZ_BENCH_INJECT_FAULT": "abort:headline@1", "BAZ_BENCH_SUPERVISE": "0"})
    assert r.returncode != 0 and not lines


def test_leg_table_and_driver_flags_on_cpu():
    """No GPU: the leg names are unique and known to the child mode, --help works, and a run without a GPU fails loudly (no CPU path)."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    assert len(set(bench.LEG_ORDER)) == len(bench.LEG_ORDER) >= 13
    assert set(bench.MUSIC_LEGS) <= set(bench.LEG_ORDER)
    for must in ("cfg2_retune_in_flight", "cfg3", "wide_m32_n2", "wide_m64_n2", "cfg5_chain", "cfg2_host_fed_gr37_model"):
        assert must in bench.LEG_ORDER
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, "bench.py"] + DRIVER_ARGV, capture_output=True, text=True, cwd=ROOT, timeout=300)
        assert r.returncode != 0 and "needs a GPU" in (r.stderr + r.stdout)
        r = subprocess.run([sys.executable, "bench.py", "--extra-leg", "cfg3"], capture_output=True, text=True, cwd=ROOT, timeout=300)
        assert r.returncode != 0 and "needs a GPU" in (r.stderr + r.stdout)
