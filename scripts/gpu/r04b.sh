#!/bin/bash
# round 4, second call: int8 scan with the seven-digit refinement -- parity tests, rates, ablation of the bulk loop
set -u
mkdir -p gpurun_out/r04b
timeout 600 python -m pytest tests/test_i8_scan.py -m gpu -x -q -s > gpurun_out/r04b/tests.txt 2>&1
echo "tests rc $?" >> gpurun_out/r04b/tests.txt
grep -v "^\.*$" gpurun_out/r04b/tests.txt | tail -12
timeout 300 python tests/lab/i8_rate.py > gpurun_out/r04b/rates.txt 2>&1
echo "rates rc $?" >> gpurun_out/r04b/rates.txt
cat gpurun_out/r04b/rates.txt
timeout 200 python tests/lab/i8_ablate.py > gpurun_out/r04b/ablate.txt 2>&1
echo "ablate rc $?" >> gpurun_out/r04b/ablate.txt
cat gpurun_out/r04b/ablate.txt
