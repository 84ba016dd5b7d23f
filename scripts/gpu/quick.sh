#!/bin/bash
# lab iterations: runs "$@" with the quick lab library (python -m gr_baz_amd.build --quick), output under gpurun_out/quick/
set -u
mkdir -p gpurun_out/quick
export BAZ_MUSIC_LAB_LIB=quick
"$@" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/quick/last.txt
