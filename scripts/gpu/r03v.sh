#!/bin/bash
# bin ranges per row for the shapes whose scan runs at fewer than 4 waves per SIMD
set -u
O=gpurun_out/r03v; mkdir -p $O
echo "cfg3 (m8 N4096 res36000, 16384 items, spectrum)" | tee -a $O/sweep.txt
timeout 300 python tests/lab/nsplit_sweep.py 8 4096 36000 16384 1 0,1,2,3,4,8 2>&1 | grep -v amdgpu.ids | tee -a $O/sweep.txt
echo "m16 N4096 res3600, 16384 items, spectrum" | tee -a $O/sweep.txt
timeout 300 python tests/lab/nsplit_sweep.py 16 4096 3600 16384 1 0,1,2,4 2>&1 | grep -v amdgpu.ids | tee -a $O/sweep.txt
echo "m16 N4096 res3600, 4096 items, spectrum" | tee -a $O/sweep.txt
timeout 300 python tests/lab/nsplit_sweep.py 16 4096 3600 4096 1 0,4,8,16 2>&1 | grep -v amdgpu.ids | tee -a $O/sweep.txt
echo "m16 N4096 res3600, 65536 items, spectrum" | tee -a $O/sweep.txt
timeout 300 python tests/lab/nsplit_sweep.py 16 4096 3600 65536 1 0,1,2 2>&1 | grep -v amdgpu.ids | tee -a $O/sweep.txt
echo "m12 n2 N3072 res3600, 16384 items" | tee -a $O/sweep.txt
timeout 300 python tests/lab/nsplit_sweep.py 12 3072 3600 16384 1 0,1,2,4 2>&1 | grep -v amdgpu.ids | tee -a $O/sweep.txt
echo "m6 N1536 res3600, 65536 items, spectrum" | tee -a $O/sweep.txt
timeout 300 python tests/lab/nsplit_sweep.py 6 1536 3600 65536 1 0,1,2,4 2>&1 | grep -v amdgpu.ids | tee -a $O/sweep.txt
