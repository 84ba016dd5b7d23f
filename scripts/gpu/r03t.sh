#!/bin/bash
# table-resident scan: parity tests, then A/B timing against the staged kernel
set -u
O=gpurun_out/r03t; mkdir -p $O
timeout 600 python -m pytest tests/test_res_scan.py -x -q -m gpu 2>&1 | tail -15 | tee $O/tests.txt
timeout 300 python tests/lab/res_scan_rate.py 262144 coherent 2>&1 | grep -v amdgpu.ids | tee $O/rate.txt
timeout 300 python tests/lab/res_scan_rate.py 262144 incoherent 2>&1 | grep -v amdgpu.ids | tee -a $O/rate.txt
