#!/bin/bash
# differential fuzz on the round's final tree (lab harnesses only; nothing under csrc/ or include/ changes)
set -u
O=gpurun_out/r03fuzz; mkdir -p $O
timeout 60 python tests/lab/fuzz.py 500 30922 2>&1 | grep "^fuzz\|FAIL" | tee $O/fuzz.txt
timeout 60 python tests/lab/fuzz_wide.py 160 4242 2>&1 | grep "^fuzz\|FAIL" | tee $O/fuzz_wide.txt
timeout 40 python tests/lab/fuzz_frontend.py 100 77 2>&1 | grep "^fuzz\|FAIL" | tee $O/fuzz_frontend.txt
timeout 40 python tests/lab/fuzz_host.py 40 9 2>&1 | grep "^fuzz\|FAIL" | tee $O/fuzz_host.txt
