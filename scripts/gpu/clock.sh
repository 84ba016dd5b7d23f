#!/bin/bash
# lab: effective shader clock of the int8 scan = GRBM_GUI_ACTIVE / duration, for ablation masks "$@" (quick lab library)
set -u
R=$(pwd); O=$R/gpurun_out/quick/clock; rm -rf $O; mkdir -p $O
export BAZ_MUSIC_LAB_LIB=quick
cd /tmp && export TMPDIR=/tmp
for abl in "$@"; do
  timeout 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/a$abl -o p -- python $R/tests/lab/i8_prof.py 8 36000 16384 4 BAZ_MUSIC_I8_ABL=$abl > $O/a$abl.out 2> $O/a$abl.err
  python - $O/a$abl $abl <<'PY'
import csv, glob, sys, collections
d, abl = sys.argv[1], sys.argv[2]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
agg = collections.defaultdict(list)
for r in csv.DictReader(open(cc[0])):
    if "scan_i8" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = [ (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) for r in csv.DictReader(open(kt[0])) if "scan_i8" in r["Kernel_Name"]] if kt else []
m = {k: sum(v) / len(v) for k, v in agg.items()}
ns = sum(dur) / len(dur) if dur else float("nan")
print("ABL %4s: %.3f ms  GRBM_GUI_ACTIVE %.3e -> %.2f GHz | wave quad-cycles %.3e: parked %.0f %%, issue-stalled %.0f %%, issuing %.0f %% | VALU insts %.3e" % (
    abl, ns * 1e-6, m.get("GRBM_GUI_ACTIVE", 0), m.get("GRBM_GUI_ACTIVE", 0) / ns if ns == ns else 0, m.get("SQ_WAVE_CYCLES", 0),
    100 * m.get("SQ_WAIT_ANY", 0) / max(1, m.get("SQ_WAVE_CYCLES", 1)), 100 * m.get("SQ_WAIT_INST_ANY", 0) / max(1, m.get("SQ_WAVE_CYCLES", 1)),
    100 * m.get("SQ_ACTIVE_INST_ANY", 0) / max(1, m.get("SQ_WAVE_CYCLES", 1)), m.get("SQ_INSTS_VALU", 0)))
PY
done
