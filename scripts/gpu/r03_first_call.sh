#!/bin/bash
# first call of the next round: the config-5 front-end with the resampler inside both AGC passes (scripts/frontend_fused_lab.hip:
# bit-for-bit check against the three-engine chain, then timings of the chain, the lane-local and the LDS-staged fused forms)
set -u
O=gpurun_out/r03a; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scripts/frontend_fused_lab scripts/frontend_fused_lab.hip 2> $O/build_err.txt || { tail $O/build_err.txt; exit 1; }
timeout 120 scripts/frontend_fused_lab 16384 20 > $O/frontend_fused.txt 2>&1; echo "rc=$?" >> $O/frontend_fused.txt; cat $O/frontend_fused.txt
# and: does the default wiring (no spectrum port) gain from contexts on separate streams?
timeout 120 python tests/lab/nospec_two_ctx.py 1,2,4 262144 > $O/nospec_two_ctx.txt 2>&1; cat $O/nospec_two_ctx.txt | grep -v amdgpu.ids
# and: does the order in which the XCDs visit the spectrum move the store rate? (scripts/ubench_hbm.hip, write_rows_class MAP 1 / 2)
hipcc --offload-arch=gfx950 -O3 -o scripts/ubench_hbm scripts/ubench_hbm.hip 2>> $O/build_err.txt && timeout 120 scripts/ubench_hbm > $O/ubench_hbm.txt 2>&1; grep "write_rows_class\|pitch 14336 plain" $O/ubench_hbm.txt
# and: where do the ~175 us of a small host-fed call go? (scripts/hostfed_call_lab.hip)
hipcc --offload-arch=gfx950 -O3 -o scripts/hostfed_call_lab scripts/hostfed_call_lab.hip 2>> $O/build_err.txt && timeout 60 scripts/hostfed_call_lab > $O/hostfed_call_lab.txt 2>&1; cat $O/hostfed_call_lab.txt
