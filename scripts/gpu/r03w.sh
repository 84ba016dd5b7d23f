#!/bin/bash
# 33..64 antennas on the matrix core: parity tests, then rates
set -u
O=gpurun_out/r03w4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide" 2>&1 | tail -5 | tee $O/tests.txt
timeout 300 python tests/lab/wide_rate.py 2>&1 | grep -v amdgpu.ids | tee $O/wide_rate.txt
