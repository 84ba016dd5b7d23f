#!/bin/bash
set -u
O=gpurun_out/r03k; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/test_gpu.txt 2>&1; echo "rc=$?" >> $O/test_gpu.txt; tail -6 $O/test_gpu.txt
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; echo "bench rc=$?"; python - <<'PY'
import json
d = json.load(open("gpurun_out/r03k/bench_line.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["rounds"])
for k, v in d["config"]["extra"].items():
    if isinstance(v, dict) and "ms_per_step" in v:
        print(k, round(v["ms_per_step"], 4), "%.3e" % v["snapshots_per_s"], v.get("stage_ms_per_launch") or v.get("engine_ms"))
    else:
        print(k, str(v)[:400])
print(d.get("cpu_baseline"))
PY
