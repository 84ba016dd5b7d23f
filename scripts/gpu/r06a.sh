#!/bin/bash
# Round 6, session A on the new tree: (1) the graded command as graded, 3 times; (2) the new tests (driver command, fault injection, retunes in
# flight on every scan path, soak); (3) 45-s two-thread retune soaks of the wide / cfg5 / gated shapes; (4) the bench's legs and the retune /
# wide / parity tests under the guard-zone allocator (lab library, BAZ_MUSIC_GUARD=1).
set -u
TAG=${1:-r06a}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
T0=$(date +%s)
for i in 1 2 3; do
  timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench$i.out 2> $O/bench$i.err; echo "bench $i rc=$? t=$(( $(date +%s) - T0 ))" | tee -a $O/summary.txt
done
python3 - $O <<'PY' | tee -a $O/summary.txt
import json, sys
for i in (1, 2, 3):
    try:
        ls = [json.loads(l) for l in open("%s/bench%d.out" % (sys.argv[1], i)) if l.startswith("{")]
        d = ls[-1]; c = d["config"]
        print("bench %d: lines %d value %.4g ms %.4f frac %.3f cpu %.3g verified %s extras_failed %s (%s) legs_s %.0f" % (i, len(ls), d["value"], d["ms_per_step"], d["roofline"]["frac"],
              d["cpu_baseline"]["value"], c["verified_ok"], c.get("extras_failed"), c.get("extras_failed_legs"), sum(v.get("leg_wall_s", 0) for v in c.get("extra", {}).values())))
    except Exception as e:
        print("bench %d: unreadable: %r" % (i, e))
PY
timeout 1500 python -m pytest tests/test_bench_driver_cmd.py tests/test_retune.py tests/test_abi.py tests/test_host_block.py -q -m gpu -x 2>&1 | tail -15 | tee $O/tests_new.txt
echo "t=$(( $(date +%s) - T0 )) s after the new tests" | tee -a $O/summary.txt
timeout 600 python tests/lab/soak_retune.py --seconds 45 wide64 wide32 wide64n8 wide24n8 cfg5 cfg3ns cfg2ns cfg3 > $O/soak.jsonl 2> $O/soak.err; echo "soak rc=$? t=$(( $(date +%s) - T0 ))" | tee -a $O/summary.txt
python3 -c "
import json,sys
for l in open('$O/soak.jsonl'):
    d=json.loads(l); print(d['shape'], 'ok' if d['ok'] else 'FAILED', 'calls', d['calls'], 'retunes', d['retunes'], 'seen', d['tables_seen'], 'worst %.3g' % d['worst_rel_err'], 'retune ms med %.3f worst %.2f' % (d['retune_ms_median'], d['retune_ms_worst']), d['errors'])
" | tee -a $O/summary.txt
# guard zones: the bench's legs in ONE process (the round-5 layout) on the lab library, then the retune / wide / parity tests
export BAZ_MUSIC_LAB_LIB=lab BAZ_MUSIC_GUARD=1
timeout 600 python3 bench.py --extra-leg ALL > $O/guard_legs.out 2> $O/guard_legs.err; echo "guarded legs rc=$? t=$(( $(date +%s) - T0 ))" | tee -a $O/summary.txt
grep -c "GUARD ZONE DAMAGED" $O/guard_legs.err | sed 's/^/guarded legs: damaged-zone reports: /' | tee -a $O/summary.txt
timeout 1500 python -m pytest tests/test_retune.py tests/test_gpu_parity.py tests/test_i8_scan.py tests/test_coarse_scan.py -q -m gpu -x 2>&1 | tail -8 | tee $O/tests_guarded.txt
echo "t=$(( $(date +%s) - T0 )) s total" | tee -a $O/summary.txt
