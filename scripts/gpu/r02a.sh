#!/bin/bash
# round-2 GPU call A: MFMA 4x4x4_4b layout probe, GPU suite on the gated scan, gate A/B, bench line
set -u
O=gpurun_out/r02a; mkdir -p $O
./scripts/probe_mfma4 $O/probe_mfma4.json > $O/probe_mfma4.txt 2>&1; echo "probe rc=$?" >> $O/probe_mfma4.txt
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
timeout 600 python tests/lab/r02_scan_gate.py 262144 > $O/scan_gate.txt 2>&1; echo "gate rc=$?" >> $O/scan_gate.txt
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_err.txt; echo "bench rc=$?" >> $O/bench_err.txt
tail -5 $O/probe_mfma4.txt $O/pytest_gpu.txt; cat $O/scan_gate.txt; tail -c 1500 $O/bench_line.json
