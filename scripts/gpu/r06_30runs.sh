#!/bin/bash
# Round 6 ("30 consecutive clean runs of the driver command recorded in profiles/"): the graded command, verbatim, 30 times on one lease.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06runs30; mkdir -p $O; cd $R
T0=$(date +%s)
for i in $(seq 1 ${1:-30}); do
  timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/out.txt 2> $O/err.txt; rc=$?
  python3 - $i $rc $O/out.txt $(( $(date +%s) - T0 )) <<'PY' | tee -a $O/summary.txt
import json, sys
i, rc, path, t = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
ls = [json.loads(l) for l in open(path) if l.startswith("{")]
if not ls:
    print("run %s rc=%s t=%s s: NO LINE" % (i, rc, t))
else:
    d = ls[-1]; c = d["config"]
    print("run %s rc=%s t=%s s: lines %d value %.4g ms/step %.4f roofline.frac %.3f cpu_baseline %.3g verified %s attempt %s legs run %s failed %s %s"
          % (i, rc, t, len(ls), d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"], c["verified_ok"], c["headline_attempt"],
             c.get("extras_run"), c.get("extras_failed"), c.get("extras_failed_legs")))
PY
  [ $rc -ne 0 ] && cp $O/err.txt $O/err_run$i.txt
done
echo "clean runs: $(grep -c 'rc=0 .* failed 0 ' $O/summary.txt) of $(grep -c . $O/summary.txt)"
