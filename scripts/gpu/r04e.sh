#!/bin/bash
# round 4, fifth call: one __shared__ object + raw barrier (no compiler vmcnt(0) in the step loop); stores spanning a wait (lab)
set -u
R=$(pwd); O=$R/gpurun_out/r04e; mkdir -p $O
timeout 600 python -m pytest tests/test_i8_scan.py -m gpu -x -q -s > $O/tests.txt 2>&1
echo "tests rc $?" >> $O/tests.txt
grep -v "^\.*$" $O/tests.txt | tail -4
timeout 200 python tests/lab/i8_ablate.py > $O/ablate.txt 2>&1
echo "ablate rc $?" >> $O/ablate.txt
cat $O/ablate.txt
timeout 300 python tests/lab/i8_rate.py > $O/rates.txt 2>&1
echo "rates rc $?" >> $O/rates.txt
cut -c1-215 $O/rates.txt
