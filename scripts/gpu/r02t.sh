#!/bin/bash
# whole GPU suite on the final library
set -u
O=gpurun_out/r02t3; mkdir -p $O
timeout 400 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -n 8 $O/pytest_gpu.txt
