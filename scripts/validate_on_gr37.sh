#!/bin/bash
# Validation kit for a host that HAS GNU Radio 3.7 (or later), SWIG, ROCm and an MI355X -- none of which coexist in this
# repository's build image, so this script has never run here; it is the one-command check INTEGRATION.md 4 refers to.
#
#   1. configures this tree with find_package(Gnuradio) (no API stand-in), builds the three kernel libraries, the host
#      blocks and the SWIG module from swig/baz_music.i (the reference's stanza, /root/reference/swig/baz_swig.i:560-574);
#   2. runs scripts/gr37/validate_flowgraph.py: vector_source -> baz.music_doa -> vector_sink over tests/golden/cfg{1,2}*.npz
#      under the real scheduler, compares at 1e-5, checks finite streams of 1 / 7 / 2,500 items and prints the call sizes the
#      real flat_flowgraph granted (the look-back buffer request of SURVEY.md 8f row 1 rests on a restatement of it);
#   3. dumps gnuradio-filter's MMSE tap table into tests/golden/mmse_taps_gr37.npz (scripts/dump_gr_mmse_taps.py) and
#      runs the resampler test that then pins the default table bit for bit.
#
# usage: scripts/validate_on_gr37.sh [build dir]        (GNU Radio's prefix on CMAKE_PREFIX_PATH / PYTHONPATH as usual)
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
BUILD=${1:-$ROOT/build_gr}
PY=${PYTHON:-python}
command -v gnuradio-config-info > /dev/null || { echo "gnuradio-config-info not found: this script needs a GNU Radio host"; exit 2; }
echo "GNU Radio $(gnuradio-config-info --version), prefix $(gnuradio-config-info --prefix)"
cmake -S "$ROOT" -B "$BUILD" -DBAZ_MUSIC_WITH_GR_SHIM=OFF -DBAZ_MUSIC_ENABLE_SWIG=ON -DCMAKE_BUILD_TYPE=Release
cmake --build "$BUILD" -j
MOD=$(dirname "$(find "$BUILD" -name 'baz_music_swig.py' | head -1)")
[ -n "$MOD" ] || { echo "the SWIG module was not built (SWIG / PythonLibs missing?)"; exit 3; }
export LD_LIBRARY_PATH="$BUILD:${LD_LIBRARY_PATH:-}"
rc=0
$PY "$ROOT/scripts/gr37/validate_flowgraph.py" "$MOD" "$ROOT" || rc=$?
$PY "$ROOT/scripts/dump_gr_mmse_taps.py" && (cd "$ROOT" && $PY -m pytest tests/test_resamp.py -q -k "gnuradio_filters or tap_table") || rc=$?
exit $rc
