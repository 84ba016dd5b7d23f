#!/bin/bash
# Evidence for the shipped code (run LAST, on the commit that is judged): rocprofv3 kernel stats and the PMC passes of
# bench.py, the traffic file bench.py reads back, a driver-style bench line, refine rate by SNR.
# usage (from the repo root on the GPU box): scripts/gpu/profile_final.sh <tag> <git head>
set -u
TAG=${1:-r02z}; HEAD=${2:-unknown}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B --steps 20 --warmup 3 > $O/stats_bench_line.json 2> $O/stats_err.txt
PM="--steps 3 --warmup 1 --ramp-seconds 0 --min-seconds 0"
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_write -o bench -- $B $PM > /dev/null 2> $O/pmc_write_err.txt
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_fetch -o bench -- $B $PM > /dev/null 2> $O/pmc_fetch_err.txt
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $O/pmc_sq -o bench -- $B $PM > /dev/null 2> $O/pmc_sq_err.txt
cd $R
W=$(find $O/pmc_write -name '*counter_collection.csv' | head -1); F=$(find $O/pmc_fetch -name '*counter_collection.csv' | head -1); S=$(find $O/pmc_sq -name '*counter_collection.csv' | head -1)
K=$(find $O/stats -name '*kernel_stats.csv' | head -1)
cp "$K" $O/bench_kernel_stats.csv 2>/dev/null
python scripts/pmc_summary.py "$W" "$F" "$S" > $O/bench_pmc_summary.txt 2>&1
python scripts/pmc_traffic.py "$W" "$F" $O/scan_pmc_traffic.json $HEAD > $O/pmc_traffic_out.txt 2>&1
cp $O/scan_pmc_traffic.json profiles/r02_scan_pmc_traffic.json      # so that the bench run below reports it
python bench.py --steps 20 --warmup 3 > $O/bench_line.json 2> $O/bench_err.txt
# raw counter rows of the product's kernels, kept beside the summaries (tests/test_evidence.py recomputes the traffic from them)
python - "$W" "$F" "$S" <<'PY'
import csv, sys
for src, name in zip(sys.argv[1:4], ("write", "fetch", "sq")):
    rows = list(csv.DictReader(open(src)))
    with open("profiles/r02_pmc_raw_%s.csv" % name, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(r for r in rows if "bazmusic" in r["Kernel_Name"])
PY
mkdir -p $O/raw && cp profiles/r02_pmc_raw_*.csv $O/raw/
python tests/lab/refine_rate.py > $O/refine_rate.txt 2>&1
python scripts/cfg5_pipeline.py 16384 20 > $O/cfg5_pipeline.txt 2>&1
head -12 $O/bench_kernel_stats.csv; cat $O/pmc_traffic_out.txt | tail -22; python -c "
import json; d=json.load(open('$O/bench_line.json')); print(d['value'], d['ms_per_step'], d['rounds']); print(d['roofline'])"
