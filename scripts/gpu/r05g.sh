#!/bin/bash
# Round 5: the strided walk in scan_mfma_kernel (A/B on one box), then the whole GPU suite on the tree.
set -u
TAG=${1:-r05g}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
cd $R
timeout 300 python tests/lab/walk_ab.py 2>&1 | grep -v amdgpu.ids | tee $O/walk_ab.txt
echo "t=$(( $(date +%s) - T0 )) s after the walk A/B"
timeout 600 python -m pytest tests -q -m gpu -x 2>&1 | tail -12 | tee $O/tests_full_gpu_suite.txt
echo "t=$(( $(date +%s) - T0 )) s after the whole GPU suite"
