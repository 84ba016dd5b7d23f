#!/bin/bash
set -u
O=gpurun_out/r03g; mkdir -p $O
timeout 900 python -m pytest tests/test_coarse_scan.py -q -m gpu -k "equals_the_full_scan or poisoned" > $O/test_prefetch1.txt 2>&1; echo "prefetch=1 rc=$?"; tail -2 $O/test_prefetch1.txt
cp scripts/lib_prefetch0.so gr_baz_amd/csrc/libbaz_music_hip.so
timeout 900 python -m pytest tests/test_coarse_scan.py -q -m gpu -k "equals_the_full_scan or poisoned" > $O/test_prefetch0.txt 2>&1; echo "prefetch=0 rc=$?"; tail -2 $O/test_prefetch0.txt
