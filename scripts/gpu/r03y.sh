#!/bin/bash
set -u
O=$(pwd)/gpurun_out/r03y; mkdir -p $O
R=$(pwd)
for mode in 0 1; do
  echo "BAZ_MUSIC_ROLES_MODE=$mode" | tee -a $O/out.txt
  BAZ_MUSIC_ROLES_MODE=$mode timeout 200 python tests/lab/roles_rate.py 262144 2>&1 | grep -v amdgpu.ids | tee -a $O/out.txt
done
cd /tmp && export TMPDIR=/tmp
BAZ_MUSIC_ROLES_MODE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats0 -o roles -- python $R/tests/lab/roles_rate.py 262144 > /dev/null 2>&1
BAZ_MUSIC_ROLES_MODE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o roles -- python $R/tests/lab/roles_rate.py 262144 > /dev/null 2>&1
for m in 0 1; do echo "mode $m"; K=$(find $O/stats$m -name '*kernel_stats.csv' | head -1); grep "bazmusic" "$K" | cut -c1-60,400- | sed 's/([^"]*"/"/' | head -6; grep "bazmusic" "$K" | awk -F'",' '{print substr($1,1,50), $2,$3,$4}' | head -6; done | tee -a $O/out.txt
