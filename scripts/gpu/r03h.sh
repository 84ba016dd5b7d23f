#!/bin/bash
set -u
O=gpurun_out/r03j; mkdir -p $O
python tests/lab/coarse_dump.py 2>&1 | grep -v amdgpu.ids | tee $O/coarse_dump.txt
python tests/lab/coarse_diff.py 2>&1 | grep -v amdgpu.ids | tee $O/coarse_diff.txt
timeout 900 python -m pytest tests/test_coarse_scan.py -q -m gpu > $O/test_coarse.txt 2>&1; echo "rc=$?" >> $O/test_coarse.txt; tail -8 $O/test_coarse.txt
timeout 300 python tests/lab/coarse_rate.py > $O/coarse_rate.txt 2>&1; grep -v amdgpu.ids $O/coarse_rate.txt | cut -c1-330
