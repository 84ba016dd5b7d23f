"""The kept measurement evidence names the code it was taken on.  CPU only."""
import json
import os

import pytest

from conftest import ROOT


# the newest round whose evidence is kept under profiles/ (scripts/gpu/rNN_final.sh)
EV = next(r for r in ("r06", "r05") if os.path.exists(os.path.join(ROOT, "profiles", "%s_bench_line.json" % r))
          and os.path.exists(os.path.join(ROOT, "profiles", "%s_scan_pmc_traffic.json" % r)))


def _bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_profiles_carry_the_sha_of_the_sources_they_were_taken_on():
    bench = _bench()
    here = bench.kernel_sources_sha()
    traffic = json.load(open(os.path.join(ROOT, "profiles", EV + "_scan_pmc_traffic.json")))
    line = json.load(open(os.path.join(ROOT, "profiles", EV + "_bench_line.json")))
    assert len(traffic["kernel_sources_sha"]) == 16 and traffic["git_head"]
    assert line["kernel_sources_sha"] == traffic["kernel_sources_sha"], "bench line and PMC traffic come from different sources"
    assert line["roofline"]["traffic"] == traffic["scan_hbm_bytes_per_launch"] and line["roofline"]["traffic_stale"] is False
    assert traffic["items_per_launch"] == line["config"]["items_per_gpu_per_step"]
    if traffic["kernel_sources_sha"] != here:
        # legitimate while kernels are being changed; at the end of a round scripts/gpu/profile_final.sh re-takes the evidence
        pytest.xfail("profiles/" + EV + "_* were taken on kernel sources %s, the tree holds %s: bench.py will report traffic_stale"
                     % (traffic["kernel_sources_sha"], here))


def test_roofline_fields_of_the_kept_bench_line_are_consistent():
    line = json.load(open(os.path.join(ROOT, "profiles", EV + "_bench_line.json")))
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # achieved = algorithmic bytes per launch / average launch duration measured in the same run
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["algorithmic_bytes_per_launch"] == (4 * 3600 + 8 * 2) * line["config"]["items_per_gpu_per_step"]
    assert 1.0 <= r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.2          # counter traffic close to the algorithmic bytes
    assert line["value"] == pytest.approx(line["config"]["items_per_gpu_per_step"] / (line["ms_per_step"] * 1e-3), rel=1e-9)
    assert line["vs_baseline"] is None and line["dtype"] == "f64" and line["higher_is_better"] is True
    cb = line["cpu_baseline"]
    assert cb["kind"] == "reference" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb


def test_every_secondary_leg_of_the_bench_stays_in_its_own_process():
    """bench.py runs each secondary leg as `python bench.py --extra-leg NAME` in a subprocess with a time limit: without a GPU the leg
    dies ("needs a GPU") and the parent gets an error ENTRY instead of an exception -- so a leg can never take the headline down."""
    for leg in ("cfg2_host_fed_gr37_model", "wide_m64_n2"):
        out = _bench().run_leg_subprocess(leg, timeout_s=120)
        assert isinstance(out, dict) and ("error" in out or "runs" in out or "snapshots_per_s" in out)
        assert "leg_wall_s" in out


def test_pmc_traffic_follows_from_the_kept_raw_counter_rows():
    """profiles/r05_pmc_raw_{write,fetch}.csv are the rocprofv3 --pmc rows of the product's kernels (one pass per counter);
    the traffic bench.py reports is their mean per scan launch, WRITE_SIZE + 2 x FETCH_SIZE in units of 1,024 B
    (MI355X_MICROARCH.md: gfx950 tallies the 128-B requests of wide reads at 64 B)."""
    import csv

    def mean(path, counter):
        v = [float(r["Counter_Value"]) for r in csv.DictReader(open(os.path.join(ROOT, "profiles", path)))
             if r["Counter_Name"] == counter and "scan_mfma_kernel" in r["Kernel_Name"]]
        assert len(v) >= 3
        return sum(v) / len(v), len(v)

    t = json.load(open(os.path.join(ROOT, "profiles", EV + "_scan_pmc_traffic.json")))
    w, nw = mean(EV + "_pmc_raw_write.csv", "WRITE_SIZE")
    f, nf = mean(EV + "_pmc_raw_fetch.csv", "FETCH_SIZE")
    assert (nw, nf) == (t["dispatches_averaged"]["write_pass"], t["dispatches_averaged"]["fetch_pass"])
    assert w == pytest.approx(t["WRITE_SIZE_KiB"], rel=1e-12) and f == pytest.approx(t["FETCH_SIZE_KiB_raw"], rel=1e-12)
    assert abs(int(round(w * 1024.0 + 2.0 * f * 1024.0)) - t["scan_hbm_bytes_per_launch"]) <= 1   # (a mean of 7 dispatches: .5 may round either way)
    assert t["writes_over_algorithmic"] == pytest.approx(w * 1024.0 / t["algorithmic_bytes_per_launch"], rel=1e-12)
    # the launch the counters saw is the bench's launch: 262,144 items = 65,536 workgroups-worth of 16-item tiles x ranges
    rows = [r for r in csv.DictReader(open(os.path.join(ROOT, "profiles", EV + "_pmc_raw_write.csv"))) if "scan_mfma_kernel" in r["Kernel_Name"]]
    assert len({(r["Grid_Size"], r["Workgroup_Size"]) for r in rows}) == 1
