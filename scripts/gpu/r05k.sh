#!/bin/bash
# Round 5: the whole GPU suite on the tree, then a driver-style bench line.
set -u
TAG=${1:-r05k}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
cd $R
timeout 700 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 | tee $O/tests_full_gpu_suite.txt
echo "t=$(( $(date +%s) - T0 )) s after the whole GPU suite"
timeout 330 python bench.py --steps 20 --warmup 3 > $O/bench_line.json 2> $O/bench_err.txt
echo "bench rc $?"; tail -3 $O/bench_err.txt
python - <<PY
import json
d = json.load(open("$O/bench_line.json"))
c = d["config"]
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"])
for k in sorted(c):
    if not isinstance(c[k], (dict, list)):
        print("  ", k, c[k])
PY
echo "t=$(( $(date +%s) - T0 )) s total"
