#!/bin/bash
# the scheduler model and page-locked stream buffers: tests, then rates by output multiple
set -u
O=gpurun_out/r02s2; mkdir -p $O
timeout 200 python -m pytest tests/test_scheduler_model.py tests/test_gpu_parity.py -q -m gpu -k "scheduler_model or page_locking or several_chunks" -x > $O/pytest.txt 2>&1; tail -n 15 $O/pytest.txt
timeout 200 python tests/lab/flowgraph_rate.py 16384 > $O/flowgraph_rate.txt 2> $O/flowgraph_err.txt; cat $O/flowgraph_rate.txt; tail -n 5 $O/flowgraph_err.txt
