#!/bin/bash
# Round 5: the scan's store drain (a compiler-inserted s_waitcnt vmcnt(0) in front of the step's last store) removed: same-box A/B.
set -u
TAG=${1:-r05l}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
cd $R
for K in 1 2; do
  for L in quick lab; do
    echo "== library $L (quick = before the fix, lab = after), pass $K" | tee -a $O/walk_ab.txt
    BAZ_MUSIC_LAB_LIB=$L timeout 200 python tests/lab/walk_ab.py 2>&1 | grep -v amdgpu.ids | grep "strided" | tee -a $O/walk_ab.txt
  done
done
echo "t=$(( $(date +%s) - T0 )) s after the A/B"
timeout 700 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 | tee $O/tests_full_gpu_suite.txt
echo "t=$(( $(date +%s) - T0 )) s after the whole GPU suite"
timeout 330 python bench.py --steps 20 --warmup 3 > $O/bench_line.json 2> $O/bench_err.txt
echo "bench rc $?"
python - <<PY
import json
d = json.load(open("$O/bench_line.json"))
c = d["config"]
print(d["value"], d["ms_per_step"], d["roofline"])
for k in ("default_wiring_snapshots_per_s", "incoherent_snapshots_per_s", "snr60_snapshots_per_s", "incoherent_snr60_snapshots_per_s", "cfg3_snapshots_per_s", "cfg5_chain_snapshots_per_s", "cfg5_chain_music_scan_ms", "verified_ok", "extras_all_verified_ok"):
    print("  ", k, c.get(k))
PY
echo "t=$(( $(date +%s) - T0 )) s total"
