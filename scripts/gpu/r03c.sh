#!/bin/bash
# round 3, call c: coarse-gated scan after the inline-asm hazard fix
set -u
O=gpurun_out/r03c; mkdir -p $O
export BAZ_MUSIC_DEBUG_MARGIN=1
timeout 900 python -m pytest tests/test_coarse_scan.py -q -m gpu -s > $O/test_coarse.txt 2>&1; echo "rc=$?" >> $O/test_coarse.txt; grep "coarse margin" $O/test_coarse.txt | sort | uniq -c | sort -rn | head -20; tail -8 $O/test_coarse.txt
timeout 300 python tests/lab/coarse_rate.py > $O/coarse_rate.txt 2>&1; grep -v amdgpu.ids $O/coarse_rate.txt
timeout 900 python -m pytest tests/test_host_block.py tests/test_scheduler_model.py tests/test_agc.py tests/test_frontend.py -q -m gpu > $O/test_misc.txt 2>&1; echo "rc=$?" >> $O/test_misc.txt; tail -12 $O/test_misc.txt
