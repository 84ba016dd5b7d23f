#!/bin/bash
# Round 6: all secondary legs of bench.py in ONE process (round 5's layout) on the lab library under the guard-zone allocator, N times in a row.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06g; mkdir -p $O; cd $R
export BAZ_MUSIC_LAB_LIB=lab BAZ_MUSIC_GUARD=1
T0=$(date +%s)
for i in $(seq 1 ${1:-25}); do
  timeout 300 python3 bench.py --extra-leg ALL > $O/out.txt 2> $O/err.txt; rc=$?
  echo "run $i rc=$rc t=$(( $(date +%s) - T0 )) s legs $(grep -c '"leg"' $O/out.txt) errors $(grep -c '"error"' $O/out.txt) damaged-zone reports $(grep -c 'GUARD ZONE DAMAGED' $O/err.txt) | $(grep 'guard zones damaged' $O/err.txt | tail -1)" | tee -a $O/summary.txt
  [ $rc -ne 0 ] && cp $O/err.txt $O/err_run$i.txt
done
echo "clean: $(grep -c 'rc=0 .* errors 0 damaged-zone reports 0' $O/summary.txt) of $(grep -c . $O/summary.txt)"
