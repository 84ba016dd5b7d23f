#!/bin/bash
# round-2 GPU call H: new tests (ordering, cfg4, dealing, scheduler hints) + rewritten bench.py
set -u
O=gpurun_out/r02h; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.txt
( time timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench_line.json 2> $O/bench_err.txt ) 2> $O/bench_time.txt; echo "bench rc=$?" >> $O/bench_err.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc=$?" >> $O/smoke.txt
tail -n 12 $O/pytest_gpu.txt; tail -n 3 $O/bench_err.txt $O/bench_time.txt $O/smoke.txt; python -c "
import json; d=json.load(open('$O/bench_line.json'))
print({k:d[k] for k in ('value','ms_per_step','rounds')}); print(d['roofline']); print(json.dumps(d['config'].get('extra'),indent=1)[:3000]); print(d.get('cpu_baseline',{}).get('value'))"
