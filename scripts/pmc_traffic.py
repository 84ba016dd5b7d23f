"""Builds profiles/<round>_scan_pmc_traffic.json from two rocprofv3 --pmc passes (WRITE_SIZE, FETCH_SIZE: separate runs,
MI355X_MICROARCH.md 'rocprofv3 PMC slots') of bench.py: HBM bytes per launch of the dominant kernel (the scan), tied to
the kernel sources it was measured on (bench.kernel_sources_sha) so that bench.py only reports it for the same code.
argv: write_csv fetch_csv out_json [git_head]"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (only for kernel_sources_sha and the workload constants; nothing runs)


def mean_per_dispatch(path, counter, needle):
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and needle in r["Kernel_Name"]:
            vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit("no %s rows for %s in %s" % (counter, needle, path))
    name = max(vals, key=lambda k: len(vals[k]))
    v = vals[name]
    return name, sum(v) / len(v), len(v)


wcsv, fcsv, out = sys.argv[1:4]
head = sys.argv[4] if len(sys.argv) > 4 else None
kname, w_kib, nw = mean_per_dispatch(wcsv, "WRITE_SIZE", "scan_mfma_kernel")
_, f_kib, nf = mean_per_dispatch(fcsv, "FETCH_SIZE", "scan_mfma_kernel")
batch = bench.STREAMS_PER_GPU * bench.ITEMS_PER_STREAM
alg = (4 * bench.RES + 8 * bench.N_EMIT) * batch
# units: WRITE_SIZE / FETCH_SIZE are KiB-like units of 1,024 B (calibrated exact for writes on a 944 MB fill, r01e);
# gfx950 FETCH_SIZE tallies the 128-B requests of wide streaming reads at 64 B (MI355X_MICROARCH.md, HBM): x2, an
# upper bound for this kernel, whose reads are the projector coefficients and the steering-table image
write_b = w_kib * 1024.0
fetch_b = 2.0 * f_kib * 1024.0
info = {
    "source": "rocprofv3 --pmc WRITE_SIZE / --pmc FETCH_SIZE (separate passes, --kernel-trace off) of "
              "`python bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 1 --ramp-seconds 0 --min-seconds 0`",
    "kernel": kname.replace("void bazmusic::", ""),
    "items_per_launch": batch,
    "dispatches_averaged": {"write_pass": nw, "fetch_pass": nf},
    "WRITE_SIZE_KiB": w_kib, "FETCH_SIZE_KiB_raw": f_kib,
    "fetch_correction": "x2 (gfx950 FETCH_SIZE counts 128-B requests of wide coalesced reads as 64 B; upper bound here)",
    "scan_hbm_bytes_per_launch": int(write_b + fetch_b),
    "algorithmic_bytes_per_launch": alg,
    "traffic_over_algorithmic": (write_b + fetch_b) / alg,
    "writes_over_algorithmic": write_b / alg,
    "kernel_sources_sha": bench.kernel_sources_sha(),
    "git_head": head,
}
json.dump(info, open(out, "w"), indent=1)
print(json.dumps(info, indent=1))
