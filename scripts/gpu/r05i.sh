#!/bin/bash
# Round 5: host-fed calls with the covariance kernel's task size by batch (A/B: -64 = round 4's tasks), the retune test five times, host tests.
set -u
TAG=${1:-r05i}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
cd $R
timeout 400 python tests/lab/hostfed_calls.py -64 0 512 2>&1 | grep -v amdgpu.ids | tee $O/hostfed_calls.txt
echo "t=$(( $(date +%s) - T0 )) s after the call sweep"
for K in 1 2 3 4 5; do timeout 120 python -m pytest tests/test_retune.py -q -m gpu -s -k does_not_stall 2>&1 | grep -E "retune|passed|failed" | tee -a $O/retune5.txt; done
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_host_block.py tests/test_scheduler_model.py tests/test_retune.py -q -m gpu -x 2>&1 | tail -5 | tee $O/tests_host.txt
timeout 100 python scripts/hostfed_extra.py 2>&1 | tail -1 | tee $O/hostfed_extra.json
echo "t=$(( $(date +%s) - T0 )) s total"
