#!/bin/bash
# Round 6: A/B of the Jacobi sweep order at m = 4 (release library in the tree = row-cyclic, quick lab library = round-robin): stage times at cfg2, then parity of the new order.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06e; mkdir -p $O
for rep in 1 2 3; do
  python tests/lab/scan_ablate.py one 2>/dev/null | tail -1 | sed 's/^/release (row-cyclic)   /'
  BAZ_MUSIC_LAB_LIB=quick BAZ_MUSIC_SCAN_VARIANT=1 python tests/lab/scan_ablate.py one 2>/dev/null | tail -1 | sed 's/^/quick   (round-robin)  /'
done | tee $O/ab.txt
BAZ_MUSIC_LAB_LIB=quick timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_coarse_scan.py -q -m gpu -x -k "cfg2 or cfg1 or m4 or golden or stage or tap" 2>&1 | tail -4 | tee $O/parity_quick.txt
