#!/bin/bash
# differential fuzz against the C oracle on the round-4 tree (int8 scan for 6 .. 16 antennas, batched literal form); seeds new this round
set -u
O=gpurun_out/r04fuzz; mkdir -p $O
timeout 150 python tests/lab/fuzz.py 1200 40922 2>&1 | grep "^fuzz\|FAIL" | tee $O/fuzz.txt
BAZ_MUSIC_LAB_LIB=lab timeout 60 python tests/lab/fuzz_wide.py 120 5252 2>&1 | grep "^fuzz\|FAIL" | tee $O/fuzz_wide.txt
timeout 40 python tests/lab/fuzz_frontend.py 100 78 2>&1 | grep "^fuzz\|FAIL" | tee $O/fuzz_frontend.txt
timeout 40 python tests/lab/fuzz_host.py 40 10 2>&1 | grep "^fuzz\|FAIL" | tee $O/fuzz_host.txt
