#!/bin/bash
# wide arrays on the matrix core (33..64 antennas; three and four emitters): parity tests, fuzz, rates
set -u
O=gpurun_out/r03w5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz.py -x -q -m gpu -k "wide or fuzz" 2>&1 | tail -5 | tee $O/tests.txt
timeout 600 python tests/lab/fuzz_wide.py 90 991 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/fuzz_wide.txt
timeout 300 python tests/lab/wide_rate.py 2>&1 | grep -v amdgpu.ids | tee $O/wide_rate.txt
for v in 0; do BAZ_MUSIC_WIDE_MFMA=0 timeout 300 python tests/lab/wide_rate.py 2>&1 | grep -v amdgpu.ids | grep "n=3\|n=4\|n=1" | sed 's/^/vector-unit scan: /' | tee -a $O/wide_rate.txt; done
