#!/bin/bash
# Round 5: items sorted by their nulls in front of the gated scan: tests, rates, kernel times.
set -u
TAG=${1:-r05j}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
cd $R
timeout 400 python -m pytest tests/test_sort.py tests/test_coarse_scan.py -q -m gpu -x -s 2>&1 | grep -v "^$" | tail -25 | tee $O/tests_sort.txt
echo "t=$(( $(date +%s) - T0 )) s after the tests"
timeout 300 python tests/lab/sort_rate.py 2>&1 | grep -v amdgpu.ids | tee $O/sort_rate.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o sort -- python $R/tests/lab/sort_rate.py > /dev/null 2> $O/stats_err.txt
K=$(find $O/stats -name '*kernel_stats.csv' | head -1); [ -n "$K" ] && cut -d, -f1-5 "$K" | head -14 | tee $O/sort_kernel_stats.txt
echo "t=$(( $(date +%s) - T0 )) s total"
