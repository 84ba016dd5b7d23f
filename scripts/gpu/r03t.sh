#!/bin/bash
# zero-copy calls cut for both link directions: parity test, direct call rates, the host block under the scheduler model
set -u
O=gpurun_out/r03t; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "zero_copy_calls_cut or page_locked or registrations or host_block" 2>&1 | tail -6 | tee $O/tests.txt
timeout 300 python tests/lab/duplex_rate.py 2>&1 | grep -v amdgpu.ids | tee $O/duplex_rate.txt
for d in 0 1; do
  echo "BAZ_MUSIC_DUPLEX=$d" | tee -a $O/flowgraph.txt
  BAZ_MUSIC_DUPLEX=$d FLOWGRAPH_RATE_PINNED_ONLY=1 timeout 300 python tests/lab/flowgraph_rate.py 16384 64 2>&1 | grep -v amdgpu.ids | tee -a $O/flowgraph.txt
done
