#!/bin/bash
# Evidence for the shipped code (run LAST, on the commit that is judged): rocprofv3 kernel stats and the PMC passes of
# bench.py, the traffic file bench.py reads back, a driver-style bench line, refine rate by SNR.
# usage (from the repo root on the GPU box): scripts/gpu/profile_final.sh <tag> <git head> [profile prefix = r03]
set -u
TAG=${1:-r03z}; HEAD=${2:-unknown}; P=${3:-r03}
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
cp $O/scan_pmc_traffic.json profiles/${P}_scan_pmc_traffic.json      # so that the bench run below reports it
python bench.py --steps 20 --warmup 3 > $O/bench_line.json 2> $O/bench_err.txt
# raw counter rows of the product's kernels, kept beside the summaries (tests/test_evidence.py recomputes the traffic from them)
P=$P python - "$W" "$F" "$S" <<'PY'
import csv, os, sys
P = os.environ["P"]
for src, name in zip(sys.argv[1:4], ("write", "fetch", "sq")):
    rows = list(csv.DictReader(open(src)))
    with open("profiles/%s_pmc_raw_%s.csv" % (P, name), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(r for r in rows if "bazmusic" in r["Kernel_Name"])
PY
mkdir -p $O/raw && cp profiles/${P}_pmc_raw_*.csv $O/raw/
# the default wiring (no spectrum port): kernel stats and SQ counters of cov4_evd + scan_coarse + merge
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_nospec -o nospec -- python $R/tests/lab/coarse_prof.py 40 > /dev/null 2> $O/stats_nospec_err.txt)
(cd /tmp && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $O/pmc_nospec -o nospec -- python $R/tests/lab/coarse_prof.py 6 > /dev/null 2> $O/pmc_nospec_err.txt)
cp "$(find $O/stats_nospec -name '*kernel_stats.csv' | head -1)" profiles/${P}_nospec_kernel_stats.csv 2>/dev/null
python scripts/pmc_summary.py "$(find $O/pmc_nospec -name '*counter_collection.csv' | head -1)" > profiles/${P}_nospec_pmc_summary.txt 2>&1
python tests/lab/coarse_rate.py > $O/coarse_rate.txt 2>&1; grep -v amdgpu.ids $O/coarse_rate.txt > profiles/${P}_coarse_scan_rates.txt
python tests/lab/refine_rate.py > $O/refine_rate.txt 2>&1
python scripts/cfg5_pipeline.py 16384 20 > $O/cfg5_pipeline.txt 2>&1
cp $O/bench_kernel_stats.csv profiles/${P}_bench_kernel_stats.csv; cp $O/bench_pmc_summary.txt profiles/${P}_bench_pmc_summary.txt
cp $O/bench_line.json profiles/${P}_bench_line.json; cp $O/stats_bench_line.json profiles/${P}_bench_line_under_rocprofv3.json
grep -v amdgpu.ids $O/refine_rate.txt > profiles/${P}_refine_rate_by_snr.txt; grep -v amdgpu.ids $O/cfg5_pipeline.txt > profiles/${P}_cfg5_pipeline.txt
head -12 $O/bench_kernel_stats.csv; cat $O/pmc_traffic_out.txt | tail -22; python -c "
import json; d=json.load(open('$O/bench_line.json')); print(d['value'], d['ms_per_step'], d['rounds']); print(d['roofline'])"
# everything this run put under profiles/ travels back through gpurun_out/ (only that directory is merged back):
#   locally afterwards:  cp gpurun_out/<tag>/profiles/* profiles/
mkdir -p $O/profiles && cp profiles/${P}_* $O/profiles/ 2>/dev/null
