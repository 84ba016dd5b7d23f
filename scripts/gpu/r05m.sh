#!/bin/bash
# Round 5: scan_mfma_kernel's rotating LDS-DMA loader: A/B on one box, then the whole GPU suite and a bench line.
set -u
TAG=${1:-r05m}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
cd $R
timeout 300 python tests/lab/loader_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/loader_ab.txt
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
