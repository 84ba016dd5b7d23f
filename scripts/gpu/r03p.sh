#!/bin/bash
# coarse-gated scan generalised to 5..8 antennas
set -u
O=gpurun_out/r03p; mkdir -p $O
timeout 1200 python -m pytest tests/test_coarse_scan.py -x -q -m gpu 2>&1 | tail -25 | tee $O/tests.txt
timeout 800 python tests/lab/coarse_rate_m8.py 2>&1 | grep -v amdgpu.ids | tee $O/rates.txt
