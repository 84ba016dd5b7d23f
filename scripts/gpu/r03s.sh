#!/bin/bash
# AGC fast path: tests, then A/B timing (agc_bench + cfg5 chain)
set -u
O=gpurun_out/r03s; mkdir -p $O
timeout 900 python -m pytest tests/test_agc.py tests/test_frontend.py -x -q -m gpu 2>&1 | tail -12 | tee $O/tests.txt
for f in 1 0; do
  echo "BAZ_AGC_FAST=$f" | tee -a $O/rate.txt
  BAZ_AGC_FAST=$f timeout 300 python tests/lab/agc_bench.py 2>&1 | grep "^agc" | tee -a $O/rate.txt
  BAZ_AGC_FAST=$f timeout 300 python scripts/cfg5_pipeline.py 16384 20 2>&1 | grep -v amdgpu.ids | tail -6 | tee -a $O/rate.txt
done
