#!/bin/bash
# round 4, third call: where the int8 scan's wave cycles go (SQ counter passes, cfg3's shape and cfg5's MUSIC stage)
set -u
R=$(pwd); O=$R/gpurun_out/r04c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
SETA="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
SETB="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM"
SETC="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"
for shape in "8 36000" "16 3600"; do
  tag=m$(echo $shape | cut -d' ' -f1)
  i=0
  for S in "$SETA" "$SETB" "$SETC"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $S --output-format csv -d $O/pmc_${tag}_$i -o p -- python $R/tests/lab/i8_prof.py $shape 16384 4 > $O/pmc_${tag}_$i.out 2> $O/pmc_${tag}_$i.err
    echo "pmc $tag set $i rc $?"
  done
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_${tag} -o p -- python $R/tests/lab/i8_prof.py $shape 16384 8 > $O/stats_${tag}.out 2> $O/stats_${tag}.err
  echo "stats $tag rc $?"
done
cd $R
python scripts/pmc_summary.py $(find $O -name '*counter_collection.csv' | sort) > $O/pmc_summary.txt 2>&1
grep -A9 "scan_i8" $O/pmc_summary.txt | head -120
for t in m8 m16; do f=$(find $O/stats_$t -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -6 $f | cut -c1-160; done
