#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lab in 0 1; do
BAZ_MUSIC_COARSE_LAB=$lab rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $O/pmc_sq$lab -o c -- python $R/tests/lab/coarse_prof.py 6 > /dev/null 2> $O/err$lab.txt
BAZ_MUSIC_COARSE_LAB=$lab rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/pmc_b$lab -o c -- python $R/tests/lab/coarse_prof.py 6 > /dev/null 2>> $O/err$lab.txt
done
cd $R
for lab in 0 1; do
python scripts/pmc_summary.py $(find $O/pmc_sq$lab -name '*counter_collection.csv' | head -1) $(find $O/pmc_b$lab -name '*counter_collection.csv' | head -1) 2>&1 | grep -A10 "scan_coarse" > $O/summary$lab.txt; echo "LAB=$lab"; cat $O/summary$lab.txt
done
