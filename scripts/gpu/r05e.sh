#!/bin/bash
# Round 5: same-box A/B of the int8 scan: the round-4 kernel (built from the previous commit into the "quick" library slot) against the fifth form.
set -u
TAG=${1:-r05e}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
cd $R
timeout 300 python -m pytest tests/test_i8_scan.py -q -m gpu -x 2>&1 | tail -5 | tee $O/tests_i8.txt
for K in 1 2; do
  for L in quick lab; do
    echo "== library $L (quick = the round-4 kernel, lab = the fifth form), pass $K" | tee -a $O/i8_ab.txt
    BAZ_MUSIC_LAB_LIB=$L timeout 200 python tests/lab/i8_rate.py ab 2>&1 | grep "int8 scan" | tee -a $O/i8_ab.txt
  done
done
BAZ_MUSIC_LAB_LIB=lab timeout 200 python tests/lab/i8_ablate.py 2>&1 | grep -v amdgpu.ids | tee $O/i8_ablate.txt
echo "t=$(( $(date +%s) - T0 )) s total"
