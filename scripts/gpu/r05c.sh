#!/bin/bash
# Round 5, third GPU call: the packed scan's first pass alone (the floor of the level-packed form), and the retune gap under the three
# priorities of the builders' side stream.
set -u
TAG=${1:-r05c}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
cd $R
timeout 300 python tests/lab/i8p_rate.py coherent incoherent 2>&1 | grep -v amdgpu.ids | tee $O/i8p_rate.txt
echo "t=$(( $(date +%s) - T0 )) s after the rates"
for P in 1 0 -1; do
  for K in 1 2; do
    BAZ_MUSIC_LAB_LIB=lab BAZ_MUSIC_TAB_PRIORITY=$P timeout 120 python -m pytest tests/test_retune.py -q -m gpu -s -k does_not_stall 2>&1 | grep -E "retune at|passed|failed|assert" | sed "s/^/priority $P: /" | tee -a $O/retune_priority.txt
  done
done
echo "t=$(( $(date +%s) - T0 )) s total"
