#!/bin/bash
# Round 6, after the stream-ordered fills: the whole GPU suite on the release library, then on the lab library under the guard-zone allocator, then the fuzzers.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06f; mkdir -p $O; cd $R
T0=$(date +%s)
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 | tee $O/tests_full_gpu_suite.txt
echo "t=$(( $(date +%s) - T0 )) s after the release suite"
BAZ_MUSIC_LAB_LIB=lab BAZ_MUSIC_GUARD=1 timeout 2400 python -m pytest tests -q -m gpu --deselect tests/test_bench_driver_cmd.py --deselect tests/test_bench_multi.py 2>&1 | tail -12 | tee $O/tests_guarded_full.txt
echo "t=$(( $(date +%s) - T0 )) s after the guarded suite"
timeout 300 python tests/lab/fuzz.py 2000 60621 2>&1 | grep "^fuzz\|FAIL" | tee $O/fuzz.txt
timeout 200 python tests/lab/fuzz_host.py 300 60622 2>&1 | grep "^fuzz\|FAIL" | tee $O/fuzz_host.txt
BAZ_MUSIC_LAB_LIB=lab BAZ_MUSIC_GUARD=1 timeout 300 python tests/lab/fuzz_host.py 300 60623 2>&1 | grep "^fuzz\|FAIL" | tee $O/fuzz_host_guarded.txt
timeout 90 python tests/lab/fuzz_frontend.py 250 60624 2>&1 | grep "^fuzz\|FAIL" | tee $O/fuzz_frontend.txt
echo "t=$(( $(date +%s) - T0 )) s total"
