#!/bin/bash
# Round 6, session C on the repaired bench.py: the graded command via its test (+ fault injections), the N > 1 tests incl. RCCL with one rank, the
# whole GPU suite, differential fuzz on this tree.
set -u
TAG=${1:-r06c2}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
T0=$(date +%s)
timeout 900 python -m pytest tests/test_bench_driver_cmd.py tests/test_bench_multi.py -q -m gpu -x -s 2>&1 | grep -v "^bench.py\|amdgpu.ids" | tail -15 | tee $O/tests_bench.txt
echo "t=$(( $(date +%s) - T0 )) s after the bench tests"
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 | tee $O/tests_full_gpu_suite.txt
echo "t=$(( $(date +%s) - T0 )) s after the whole GPU suite"
timeout 300 python tests/lab/fuzz.py 3000 60601 2>&1 | grep "^fuzz\|FAIL" | tee $O/fuzz.txt
BAZ_MUSIC_LAB_LIB=lab timeout 120 python tests/lab/fuzz_wide.py 300 60602 2>&1 | grep "^fuzz\|FAIL" | tee $O/fuzz_wide.txt
timeout 90 python tests/lab/fuzz_frontend.py 250 60603 2>&1 | grep "^fuzz\|FAIL" | tee $O/fuzz_frontend.txt
timeout 120 python tests/lab/fuzz_host.py 120 60604 2>&1 | grep "^fuzz\|FAIL" | tee $O/fuzz_host.txt
echo "t=$(( $(date +%s) - T0 )) s total"
