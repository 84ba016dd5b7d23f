#!/bin/bash
# wide arrays on the matrix core: tests, then rate A/B
set -u
O=gpurun_out/r03m; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide" 2>&1 | tail -15 | tee $O/tests.txt
for mode in "1 1" "0 0"; do
  set -- $mode
  echo "BAZ_MUSIC_WIDE_MFMA=$1 BAZ_MUSIC_WIDE_COV_MFMA=$2" | tee -a $O/rate.txt
  BAZ_MUSIC_WIDE_MFMA=$1 BAZ_MUSIC_WIDE_COV_MFMA=$2 timeout 600 python tests/lab/wide_rate.py 2>&1 | grep "^m=" | tee -a $O/rate.txt
done
