#!/bin/bash
# Round 5: host-fed calls on page-locked buffers: one lane against the two-lane pipeline, by call size; then the host-path tests.
set -u
TAG=${1:-r05h}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
cd $R
timeout 400 python tests/lab/hostfed_calls.py 0 256 512 1024 2>&1 | grep -v amdgpu.ids | tee $O/hostfed_calls.txt
echo "t=$(( $(date +%s) - T0 )) s after the call sweep"
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_host_block.py tests/test_scheduler_model.py tests/test_retune.py -q -m gpu -x 2>&1 | tail -8 | tee $O/tests_host.txt
echo "t=$(( $(date +%s) - T0 )) s total"
