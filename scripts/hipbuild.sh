#!/bin/bash
# dev helper: rebuild libbaz_music_hip.so with resource-usage remarks and print a filtered table
cd "$(dirname "$0")/.." || exit 1
t0=$(date +%s)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -mllvm -amdgpu-mfma-vgpr-form -Rpass-analysis=kernel-resource-usage \
  -I include -o gr_baz_amd/csrc/libbaz_music_hip.so gr_baz_amd/csrc/baz_music_hip.hip 2> /tmp/build.log
rc=$?
echo "hipcc rc=$rc in $(( $(date +%s) - t0 )) s"
grep -E "error" -A6 /tmp/build.log | head -40
if [ -n "$1" ]; then python scripts/resusage.py /tmp/build.log | grep -E "$1"; fi
exit $rc
