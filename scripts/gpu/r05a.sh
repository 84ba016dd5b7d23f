#!/bin/bash
# Round 5, first GPU call: the retune tests (device-built table images against the host checker, the concurrent retune), the
# whole GPU suite on the new table path, then a driver-style bench line (self-verifying since this round).
set -u
TAG=${1:-r05a}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
cd $R
timeout 300 python -m pytest tests/test_retune.py -x -q -m gpu -s 2>&1 | tail -25 | tee $O/tests_retune.txt
echo "t=$(( $(date +%s) - T0 )) s after the retune tests"
timeout 400 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 | tee $O/tests_full_gpu_suite.txt
echo "t=$(( $(date +%s) - T0 )) s after the whole GPU suite"
timeout 330 python bench.py --steps 20 --warmup 3 > $O/bench_line.json 2> $O/bench_err.txt
echo "bench rc $?"; tail -5 $O/bench_err.txt
python - <<PY
import json
d = json.load(open("$O/bench_line.json"))
c = d["config"]
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"])
for k in sorted(c):
    if not isinstance(c[k], (dict, list)):
        print("  ", k, c[k])
for k, v in c.get("extra", {}).items():
    if isinstance(v, dict):
        print(k, {kk: v[kk] for kk in ("snapshots_per_s", "verified_max_rel_err", "verified_bins_identical", "verified_ok", "retune_ms", "scan_kernel_launched", "error") if kk in v})
print(d.get("cpu_baseline"))
PY
echo "t=$(( $(date +%s) - T0 )) s total"
