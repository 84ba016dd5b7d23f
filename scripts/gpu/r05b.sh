#!/bin/bash
# Round 5, second GPU call: the retune tests, the level-packed int8 scan (m <= 4): its tests, then its rate against the fp64 scan.
set -u
TAG=${1:-r05b}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
cd $R
timeout 300 python -m pytest tests/test_retune.py -x -q -m gpu -s 2>&1 | tail -15 | tee $O/tests_retune.txt
echo "t=$(( $(date +%s) - T0 )) s after the retune tests"
timeout 600 python -m pytest tests/test_i8p_scan.py -q -m gpu -x 2>&1 | tail -30 | tee $O/tests_i8p.txt
echo "t=$(( $(date +%s) - T0 )) s after the packed-scan tests"
timeout 300 python tests/lab/i8p_rate.py coherent incoherent 2>&1 | grep -v amdgpu.ids | tee $O/i8p_rate.txt
echo "t=$(( $(date +%s) - T0 )) s after the rates"
if [ "${FULL:-1}" = 1 ]; then
timeout 500 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 | tee $O/tests_full_gpu_suite.txt
echo "t=$(( $(date +%s) - T0 )) s after the whole GPU suite"
fi
