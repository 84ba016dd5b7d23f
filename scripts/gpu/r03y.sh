#!/bin/bash
# resident scan: the ranges of a slot on ONE XCD (BAZ_MUSIC_RES_LAB=100) against the plain order; nsplit 4 and 8
set -u
O=gpurun_out/r03y; mkdir -p $O
for lab in 0 100; do for ns in 0 8; do
  echo "BAZ_MUSIC_RES_LAB=$lab BAZ_MUSIC_NSPLIT=$ns" | tee -a $O/rate.txt
  BAZ_MUSIC_RES_LAB=$lab BAZ_MUSIC_NSPLIT=$ns timeout 300 python tests/lab/res_scan_rate.py 262144 coherent 2>&1 | grep -v amdgpu.ids | head -2 | tee -a $O/rate.txt
done; done
