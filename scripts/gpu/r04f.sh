#!/bin/bash
# round 4: the whole GPU suite on the fourth form of the int8 scan (two tiers, whole rounds of workgroup slots, one __shared__
# object + raw barrier), then its rates against the fp64 scan
set -u
R=$(pwd); O=$R/gpurun_out/r04f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/suite.txt 2>&1
echo "suite rc $?" >> $O/suite.txt
tail -5 $O/suite.txt
timeout 300 python -m pytest tests/test_i8_scan.py -m gpu -q -s -k "error_bounds" > $O/bounds.txt 2>&1
grep "worst" $O/bounds.txt | tail -40
timeout 300 python tests/lab/i8_rate.py > $O/rates.txt 2>&1
echo "rates rc $?" >> $O/rates.txt
cut -c1-215 $O/rates.txt
