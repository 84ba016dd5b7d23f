#!/bin/bash
# config 5 (16 antennas, short-form scan at 2 waves per SIMD): bin ranges per row
set -u
O=gpurun_out/r03u; mkdir -p $O
for ns in 0 2 3 4 6 8 12; do
  echo "BAZ_MUSIC_NSPLIT=$ns" | tee -a $O/cfg5.txt
  BAZ_MUSIC_NSPLIT=$ns timeout 300 python scripts/cfg5_pipeline.py 16384 20 2>&1 | grep "stages" | tee -a $O/cfg5.txt
done
