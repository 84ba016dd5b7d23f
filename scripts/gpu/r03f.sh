#!/bin/bash
set -u
O=gpurun_out/r03f; mkdir -p $O
scripts/ubench_f16mfma | tee $O/ubench_f16mfma.txt
for lazy in 0 1; do
  BAZ_MUSIC_COARSE_LAZY=$lazy timeout 900 python -m pytest tests/test_coarse_scan.py -q -m gpu -k "equals_the_full_scan or poisoned" > $O/test_lazy$lazy.txt 2>&1; echo "lazy=$lazy rc=$?"; tail -3 $O/test_lazy$lazy.txt
done
