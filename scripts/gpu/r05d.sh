#!/bin/bash
# Round 5, fourth GPU call: the int8 scan's fifth form (two passes, float32 combination; 6 .. 16 antennas): its tests, its rates, its parts.
set -u
TAG=${1:-r05d}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
cd $R
timeout 600 python -m pytest tests/test_i8_scan.py -q -m gpu -x 2>&1 | tail -15 | tee $O/tests_i8.txt
echo "t=$(( $(date +%s) - T0 )) s after the int8 scan's tests"
timeout 300 python tests/lab/i8_rate.py 2>&1 | grep -v amdgpu.ids | tee $O/i8_rate.txt
echo "t=$(( $(date +%s) - T0 )) s after the rates"
timeout 300 python tests/lab/i8_ablate.py 2>&1 | grep -v amdgpu.ids | tee $O/i8_ablate.txt
echo "t=$(( $(date +%s) - T0 )) s after the parts"
