#!/bin/bash
# resident vs staged scan at equal numbers of bin ranges
set -u
O=gpurun_out/r03u; mkdir -p $O
for ns in 1 2 4 8; do
  echo "BAZ_MUSIC_NSPLIT=$ns" | tee -a $O/rate.txt
  BAZ_MUSIC_NSPLIT=$ns timeout 300 python tests/lab/res_scan_rate.py 262144 coherent 2>&1 | grep -v amdgpu.ids | head -2 | tee -a $O/rate.txt
done
