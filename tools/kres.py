#!/usr/bin/env python3
"""kernel resource usage (VGPRs, SGPRs, scratch, occupancy, LDS) of the HIP sources, from hipcc's
-Rpass-analysis=kernel-resource-usage remarks.  Usage: tools/kres.py [file.hip ...]  (default: all of csrc/)"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "hdl_deflate_amd/csrc/*.hip")))
for f in files:
    p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", f, "-o", "/dev/null",
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    cur = None
    rows = []
    for ln in p.stderr.splitlines():
        m = re.search(r"remark: (?:Function )?Name: (\S+)", ln)
        if m:
            cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+(\w[\w ]*\w)(?: \[bytes/lane\]| \[waves/SIMD\]| \[bytes/block\])?: (\d+)", ln)
        if m and cur is not None:
            cur[m.group(1)] = int(m.group(2))
    if p.returncode != 0:
        print(p.stderr[-3000:])
    for r in rows:
        print("%-58s VGPR %3d  AGPR %3d  SGPR %3d  scratch %4d  occ %d  LDS %6d" % (
            r["name"][:58], r.get("VGPRs", -1), r.get("AGPRs", 0), r.get("TotalSGPRs", -1), r.get("ScratchSize", -1),
            r.get("Occupancy", -1), r.get("LDS Size", -1)))
