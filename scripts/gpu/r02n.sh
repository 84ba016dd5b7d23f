#!/bin/bash
# N > 1 code path of bench.py on the 1-GPU box: two ranks sharing the device (gloo for the barrier / clock)
set -u
O=gpurun_out/r02n; mkdir -p $O
BAZ_BENCH_SHARE_DEVICES=1 timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 > $O/bench_n2.json 2> $O/bench_n2_err.txt; echo "rc=$?" >> $O/bench_n2_err.txt
tail -n 4 $O/bench_n2_err.txt; python -c "
import json; d=json.load(open('$O/bench_n2.json')); print(d['n_gpus'], d['value'], d['ms_per_step'], d['scaling']); print(json.dumps(d['config']['ranks'])); print(d['config']['collective_backend_for_barrier_and_clock'], d['config']['parallelism'], 'extra' in d['config'], 'cpu_baseline' in d)"
