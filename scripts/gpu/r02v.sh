#!/bin/bash
B="python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 3"
pick='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "%.4g"%d["value"], "%.4f"%d["ms_per_step"], "%.4f"%d["rounds"]["ms_per_step_min"], d["config"]["stage_ms_per_launch_separate_pass"])'
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for v in 1 0 1 0; do
BAZ_MUSIC_SUB_EVD=$v timeout 120 $B 2>/dev/null | python -c "$pick" "sub_evd=$v"
done
