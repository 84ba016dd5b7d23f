#!/bin/bash
# round 4, first call: the int8 scan -- parity tests (stop at the first failure), then rates
set -u
mkdir -p gpurun_out/r04a
timeout 600 python -m pytest tests/test_i8_scan.py -m gpu -x -q -s > gpurun_out/r04a/tests.txt 2>&1
echo "tests rc $?" >> gpurun_out/r04a/tests.txt
tail -25 gpurun_out/r04a/tests.txt
timeout 300 python tests/lab/i8_rate.py > gpurun_out/r04a/rates.txt 2>&1
echo "rates rc $?" >> gpurun_out/r04a/rates.txt
cat gpurun_out/r04a/rates.txt
