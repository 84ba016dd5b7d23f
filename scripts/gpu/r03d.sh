#!/bin/bash
set -u
O=gpurun_out/r03e; mkdir -p $O

timeout 900 python -m pytest tests/test_coarse_scan.py -q -m gpu > $O/test_coarse.txt 2>&1; echo "rc=$?" >> $O/test_coarse.txt; tail -8 $O/test_coarse.txt
timeout 300 python tests/lab/coarse_rate.py > $O/coarse_rate.txt 2>&1; grep -v amdgpu.ids $O/coarse_rate.txt
