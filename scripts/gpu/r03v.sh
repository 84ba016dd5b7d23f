#!/bin/bash
# resident scan: what does a store drain (s_waitcnt vmcnt(0)) every k steps cost?
set -u
O=gpurun_out/r03v; mkdir -p $O
for k in 0 8 4 2 1; do
  echo "BAZ_MUSIC_RES_LAB=$k (drain every k steps)" | tee -a $O/rate.txt
  BAZ_MUSIC_RES_LAB=$k timeout 300 python tests/lab/res_scan_rate.py 262144 coherent 2>&1 | grep -v amdgpu.ids | head -2 | tail -1 | tee -a $O/rate.txt
done
