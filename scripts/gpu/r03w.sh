#!/bin/bash
# wide arrays after a change to the run-time-m kernels: parity tests (incl. goldens), fuzz, rates
set -u
O=gpurun_out/r03w9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -x -q -m gpu -k "wide or fuzz or golden" 2>&1 | tail -4 | tee $O/tests.txt
timeout 600 python tests/lab/fuzz_wide.py 80 808 2>&1 | grep "^fuzz\|FAIL" | tee $O/fuzz_wide.txt
timeout 300 python tests/lab/wide_rate.py 2>&1 | grep -v amdgpu.ids | tee $O/wide_rate.txt
