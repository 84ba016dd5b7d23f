#!/bin/bash
# round 3, call b: the coarse-gated scan (tests + rates), the new N=2 bench test, the partly-page-locked test
set -u
O=gpurun_out/r03b; mkdir -p $O
timeout 900 python -m pytest tests/test_coarse_scan.py -q -m gpu > $O/test_coarse.txt 2>&1; echo "rc=$?" >> $O/test_coarse.txt; tail -15 $O/test_coarse.txt
timeout 300 python tests/lab/coarse_rate.py > $O/coarse_rate.txt 2>&1; grep -v amdgpu.ids $O/coarse_rate.txt
timeout 900 python -m pytest tests/test_host_block.py tests/test_scheduler_model.py tests/test_agc.py tests/test_frontend.py -q -m gpu > $O/test_misc.txt 2>&1; echo "rc=$?" >> $O/test_misc.txt; tail -15 $O/test_misc.txt
