#!/usr/bin/env python3
"""Which kernels of libbaz_music_hip.so keep registers in scratch memory?  Extracts the gfx950 code object from the library's fat binary and
prints, per kernel, the private (scratch) segment, VGPRs, spilled VGPRs and static LDS from its metadata notes.
usage: python scripts/kernel_scratch_report.py [path/to/lib.so]     (needs objcopy and /opt/rocm/lib/llvm/bin/{clang-offload-bundler,llvm-readelf})"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gr_baz_amd", "csrc", "libbaz_music_hip.so")
llvm = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as d:
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "co.hsaco")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    subprocess.check_call([os.path.join(llvm, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
rows = []
for k in re.split(r"\n\s*- \.agpr_count", notes)[1:]:
    g = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, k).group(1))
    rows.append([g("private_segment_fixed_size"), g("vgpr_count"), g("vgpr_spill_count"), g("group_segment_fixed_size"), re.search(r"\.name:\s+(\S+)", k).group(1)])
names = subprocess.run(["c++filt"], input="\n".join(r[4] for r in rows), capture_output=True, text=True).stdout.splitlines()
print("# %s: %d kernels, %d with a scratch segment" % (os.path.relpath(lib, ROOT), len(rows), sum(1 for r in rows if r[0])))
print("# scratch B/lane  VGPRs  spilled VGPRs  static LDS B  kernel")
for r, n in sorted(zip(rows, names), key=lambda t: (-t[0][0], t[1])):
    if r[0]:
        print("%6d %5d %5d %7d  %s" % (r[0], r[1], r[2], r[3], re.sub(r"\(.*", "", n)))
