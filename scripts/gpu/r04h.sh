#!/bin/bash
# round 4: the literal-form refinement with batched operand loads (m <= 4): parity at high SNR, then the bench's cfg2 rates
set -u
R=$(pwd); O=$R/gpurun_out/r04h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_coarse_scan.py -m gpu -x -q > $O/tests.txt 2>&1
echo "tests rc $?" >> $O/tests.txt
tail -3 $O/tests.txt
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err
echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04h/bench.json').read().strip().splitlines()[-1])
print("headline %.4g  ms %.4f  scan frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]))
for k,v in d["config"].items():
    if isinstance(v,(int,float)) and ("snapshots" in k or "frac" in k): print("  %-50s %.4g" % (k, v))
PY
