#!/bin/bash
# First GPU call of the next round: four experiments prepared at the end of round 2, when the GPU minutes were spent.
# The harnesses are built where this runs unless an up-to-date binary travelled with the snapshot (build them beforehand:
# `bash scripts/gpu/r03_first_call.sh build`).
set -u
build() {   # build <binary> <extra flags...>
    local bin=$1; shift
    [ "$bin" -nt "$bin.hip" ] || hipcc --offload-arch=gfx950 -O3 "$@" -o "$bin" "$bin.hip"
}
build scripts/frontend_fused_lab -std=c++17 || exit 1
build scripts/ubench_hbm || exit 1
build scripts/hostfed_call_lab || exit 1
[ "${1:-}" = build ] && exit 0
O=gpurun_out/r03a; mkdir -p $O
# 1. the config-5 front-end with the resampler inside both AGC passes: bit-for-bit check against the three-engine chain,
#    then timings of the chain, the lane-local and the LDS-staged fused forms
timeout 120 scripts/frontend_fused_lab 16384 20 > $O/frontend_fused.txt 2>&1; echo "rc=$?" >> $O/frontend_fused.txt; cat $O/frontend_fused.txt
# 2. does the default wiring (no spectrum port) gain from contexts on separate streams?
timeout 120 python tests/lab/nospec_two_ctx.py 1,2,4 262144 > $O/nospec_two_ctx.txt 2>&1; grep -v amdgpu.ids $O/nospec_two_ctx.txt
# 3. does the order in which the XCDs visit the spectrum move the store rate? (write_rows_class MAP 1 / 2)
timeout 120 scripts/ubench_hbm > $O/ubench_hbm.txt 2>&1; grep "write_rows_class\|pitch 14336 plain" $O/ubench_hbm.txt
# 4. where do the ~175 us of a small host-fed call go?
timeout 60 scripts/hostfed_call_lab > $O/hostfed_call_lab.txt 2>&1; cat $O/hostfed_call_lab.txt
