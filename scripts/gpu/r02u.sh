#!/bin/bash
# zero-copy small calls on page-locked stream buffers: the whole GPU suite on this library, then rates with the path off / on
set -u
O=gpurun_out/r02u2; mkdir -p $O
timeout 300 python -m pytest tests -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -n 5 $O/pytest_gpu.txt
export FLOWGRAPH_RATE_PINNED_ONLY=1
for zc in 0 1; do
  BAZ_MUSIC_ZERO_COPY=$zc timeout 100 python tests/lab/flowgraph_rate.py 16384 1,64,256,1024 >> $O/flowgraph_rate.txt 2>> $O/flowgraph_err.txt
done
cat $O/flowgraph_rate.txt | cut -c1-250; grep -v "MUSIC DOA: M" $O/flowgraph_err.txt | tail -n 5
