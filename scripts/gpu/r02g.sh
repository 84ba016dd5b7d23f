#!/bin/bash
# round-2 GPU call G: upper-triangle Jacobi + fast reciprocal roots
set -u
O=gpurun_out/r02g; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
timeout 900 python tests/lab/r02_scan_gate.py 262144 0,8 > $O/scan_gate.txt 2>&1; echo "gate rc=$?" >> $O/scan_gate.txt
timeout 900 python tests/lab/fuzz.py 600 > $O/fuzz.txt 2>&1; echo "fuzz rc=$?" >> $O/fuzz.txt
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; echo "bench rc=$?" >> $O/bench_err.txt
tail -n 15 $O/pytest_gpu.txt; cat $O/scan_gate.txt; tail -n 8 $O/fuzz.txt; tail -c 1200 $O/bench_line.json
