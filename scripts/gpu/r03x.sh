#!/bin/bash
# resident scan v2 (next group's q fetched a group ahead, lane coordinates re-derived per group)
set -u
O=gpurun_out/r03x; mkdir -p $O
timeout 600 python -m pytest tests/test_res_scan.py -x -q -m gpu 2>&1 | tail -4 | tee $O/tests.txt
timeout 300 python tests/lab/res_scan_rate.py 262144 coherent 2>&1 | grep -v amdgpu.ids | tee $O/rate.txt
BAZ_MUSIC_NSPLIT=8 timeout 300 python tests/lab/res_scan_rate.py 262144 coherent 2>&1 | grep -v amdgpu.ids | head -2 | tee -a $O/rate.txt
