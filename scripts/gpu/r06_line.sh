#!/bin/bash
# Round 6: the evidence bench line once more with the final bench.py (supervised measuring process), the driver's exact command; kernels unchanged
# since scripts/gpu/r06_final.sh took the profiles (same kernel_sources_sha, so roofline.traffic still applies).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06line; mkdir -p $O; cd $R
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_stdout.txt 2> $O/bench_err.txt; echo "driver command rc=$?" | tee $O/bench_rc.txt
grep '^{' $O/bench_stdout.txt | tail -1 > $O/bench_line.json; grep -o 'complete line before the legs: {.*' $O/bench_err.txt | sed 's/^complete line before the legs: //' | head -1 > $O/bench_line_first.json   # (stdout holds ONE line; the early copy is on stderr)
python3 -c "
import json; d=json.load(open('$O/bench_line.json')); c=d['config']; print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_stale'], d['kernel_sources_sha']); print(d['cpu_baseline']['value'], d['cpu_baseline']['kind']); print('attempt', c['headline_attempt'], 'extras', c['extras_run'], 'failed', c['extras_failed'], 'verified', c['verified_ok'], c['extras_all_verified_ok'])"
grep -c '^{' $O/bench_stdout.txt; tail -3 $O/bench_err.txt
