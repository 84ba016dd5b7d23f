#!/bin/bash
# round-2 GPU call D: dwordx4 / 4x4x4 covariance kernel
set -u
O=gpurun_out/r02e; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
timeout 900 python tests/lab/r02_scan_gate.py 262144 0,8,7 > $O/scan_gate.txt 2>&1; echo "gate rc=$?" >> $O/scan_gate.txt
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; echo "bench rc=$?" >> $O/bench_err.txt
tail -n 15 $O/pytest_gpu.txt; cat $O/scan_gate.txt; tail -c 1200 $O/bench_line.json
