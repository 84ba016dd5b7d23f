#!/bin/bash
# Round 6: the WHOLE GPU suite on the lab library under the guard-zone allocator (BAZ_MUSIC_GUARD=1): zones checked after every test.
# (tests that are about the release library itself -- the knobs it reads, the release-only taps -- are deselected.)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06guard; mkdir -p $O; cd $R
export BAZ_MUSIC_LAB_LIB=lab BAZ_MUSIC_GUARD=1
timeout 2400 python -m pytest tests -q -m gpu --deselect tests/test_bench_driver_cmd.py --deselect tests/test_bench_multi.py 2>&1 | tail -25 | tee $O/tests_guarded_full.txt
