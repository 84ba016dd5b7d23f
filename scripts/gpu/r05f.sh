#!/bin/bash
# Round 5: same-box A/B of the int8 scan: round 4's kernel ("quick" library slot) against round 5's (float32 combination, strided walk; 256 = left to right).
set -u
TAG=${1:-r05f}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
T0=$(date +%s)
cd $R
timeout 300 python -m pytest tests/test_i8_scan.py -q -m gpu -x 2>&1 | tail -5 | tee $O/tests_i8.txt
for K in 1 2; do
  echo "== round 4's kernel, pass $K" | tee -a $O/i8_ab.txt
  BAZ_MUSIC_LAB_LIB=quick timeout 200 python tests/lab/i8_ablate.py 0 1 2>&1 | grep -v amdgpu.ids | tee -a $O/i8_ab.txt
  echo "== round 5's kernel, pass $K" | tee -a $O/i8_ab.txt
  BAZ_MUSIC_LAB_LIB=lab timeout 200 python tests/lab/i8_ablate.py 0 1 256 257 32 2>&1 | grep -v amdgpu.ids | tee -a $O/i8_ab.txt
done
for L in quick lab; do
  echo "== library $L" | tee -a $O/i8_rate_ab.txt
  BAZ_MUSIC_LAB_LIB=$L timeout 200 python tests/lab/i8_rate.py ab 2>&1 | grep "int8 scan" | tee -a $O/i8_rate_ab.txt
done
echo "t=$(( $(date +%s) - T0 )) s total"
