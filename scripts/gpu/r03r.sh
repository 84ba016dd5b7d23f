#!/bin/bash
# full GPU suite + the default bench line
set -u
O=gpurun_out/r03r; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $O/tests.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03r/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('metric','value','ms_per_step','n_gpus')}, d['roofline']['frac'])
for k,v in d['config']['extra'].items():
    if isinstance(v,dict): print(k, v.get('snapshots_per_s', v.get('error', '')), v.get('ms_per_step',''), v.get('stage_ms_per_launch',''))
print(d.get('cpu_baseline'))
PY
