#!/bin/bash
# wide arrays: five to eight emitters through the subspace iteration; tests, fuzz, rates
set -u
O=gpurun_out/r03w7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -x -q -m gpu -k "wide or fuzz" 2>&1 | tail -5 | tee $O/tests.txt
timeout 600 python tests/lab/fuzz_wide.py 100 4321 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/fuzz_wide.txt
timeout 300 python tests/lab/wide_rate.py 2>&1 | grep -v amdgpu.ids | grep "n=8\|n=4" | tee $O/wide_rate.txt
