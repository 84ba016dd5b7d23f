#!/bin/bash
# round 4, fourth call: third form of the int8 scan (one comparison per value, level 4 folded, whole rounds of workgroup slots)
set -u
R=$(pwd); O=$R/gpurun_out/r04d; mkdir -p $O
timeout 600 python -m pytest tests/test_i8_scan.py -m gpu -x -q -s > $O/tests.txt 2>&1
echo "tests rc $?" >> $O/tests.txt
grep -v "^\.*$" $O/tests.txt | tail -8
timeout 300 python tests/lab/i8_rate.py > $O/rates.txt 2>&1
echo "rates rc $?" >> $O/rates.txt
cat $O/rates.txt
timeout 200 python tests/lab/i8_ablate.py > $O/ablate.txt 2>&1
echo "ablate rc $?" >> $O/ablate.txt
cat $O/ablate.txt
cd /tmp && export TMPDIR=/tmp
SETA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
SETB="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM"
for shape in "8 36000" "16 3600"; do
  tag=m$(echo $shape | cut -d' ' -f1)
  i=0
  for S in "$SETA" "$SETB"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $S --output-format csv -d $O/pmc_${tag}_$i -o p -- python $R/tests/lab/i8_prof.py $shape 16384 4 > $O/pmc_${tag}_$i.out 2> $O/pmc_${tag}_$i.err
  done
done
cd $R
python scripts/pmc_summary.py $(find $O -name '*counter_collection.csv' | sort) > $O/pmc_summary.txt 2>&1
grep -A9 "scan_i8" $O/pmc_summary.txt | cut -c1-100
