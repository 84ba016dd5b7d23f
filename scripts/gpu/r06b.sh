#!/bin/bash
# Round 6, session B: (1) the split default-wiring pipeline: bit-identity tests, then rates by parts / covariance grid; (2) host-fed calls on
# page-locked buffers by chunk size (the deep copy pipeline); (3) the host-block tests that exercise the chunked path.
set -u
TAG=${1:-r06b}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
T0=$(date +%s)
timeout 900 python -m pytest tests/test_split.py tests/test_coarse_scan.py -q -m gpu -x 2>&1 | tail -12 | tee $O/tests_split.txt
echo "t=$(( $(date +%s) - T0 )) s after the split tests"
timeout 600 python tests/lab/split_rate.py 262144 65536 32768 16384 2>&1 | tee $O/split_rate.txt
echo "t=$(( $(date +%s) - T0 )) s after the split rates"
timeout 900 python tests/lab/hostfed_chunk_sweep.py 4 6 8 12 16 24 2>&1 | tee $O/hostfed_chunk_sweep.txt
echo "t=$(( $(date +%s) - T0 )) s after the chunk sweep"
timeout 900 python -m pytest tests/test_host_block.py tests/test_gpu_parity.py -q -m gpu -x -k "host or process or chunk or pinned or page or finite or stream" 2>&1 | tail -8 | tee $O/tests_host.txt
echo "t=$(( $(date +%s) - T0 )) s total"
