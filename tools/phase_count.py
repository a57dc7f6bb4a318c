#!/usr/bin/env python3
"""Static per-phase instruction count of k_compress's tile body: hipcc -S, then count the instructions between the
`; @@PHASE name` marks (HDLZ_MARK in hdlz_compress.hip).  The tile body is straight-line code executed once per tile,
so static counts = dynamic counts per wave-tile; issue cycles are priced with the measured table of
tools/ubench/valu_cycles*.hip (1.96 shader cycles for v_add/sub/and/or/xor/lshr/ashr/mov/min_u16/bitop3, 3.25 for every other VALU op).
Usage: tools/phase_count.py [-DNAME ...] [--kernel MANGLED_SUBSTR] [--src FILE.hip]
(k_inflate_tok: --src hdlz_inflate_tok.hip -DHDLZ_TOK_MARKS --kernel k_inflate_tokILb1E -- parts of the round loop, branches counted once)"""
import collections
import re
import subprocess
import sys

FAST = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_mov_b32", "v_min_u16",
        "v_add_co_u32", "v_not_b32", "v_ashrrev_i32", "v_bitop3_b32", "v_max_i16")      # 1.96 cycles (profiles/r04_ubench/ubench_valu_cycles*.txt)
defs = [a for a in sys.argv[1:] if a.startswith("-D")]
kern = "k_compressILi1ELb1ELb1E"
srcname = "hdlz_compress.hip"
for i, a in enumerate(sys.argv):
    if a == "--kernel":
        kern = sys.argv[i + 1]
    if a == "--src":
        srcname = sys.argv[i + 1]
import os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "hdl_deflate_amd/csrc", srcname)
asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", "-", src] + defs,
                     capture_output=True, text=True).stdout
m = re.search(r"^(_ZN4hdlz\w*%s\w*):[^\n]*\n(.*?)^\.Lfunc_end" % re.escape(kern), asm, re.S | re.M)
body = m.group(2)
phase = "prologue"
cnt = collections.OrderedDict()
for ln in body.splitlines():
    t = ln.strip()
    pm = re.match(r"; @@PHASE (\w+)", t)
    if pm:
        phase = pm.group(1)
        continue
    op = t.split()[0] if t and not t.startswith((";", ".", "//")) and not t.endswith(":") else None
    if not op:
        continue
    c = cnt.setdefault(phase, collections.Counter())
    base = re.sub(r"_e32$|_e64$|_sdwa$|_dpp$", "", op)
    if op.startswith("v_"):
        if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            c["valu_slow"] += 1
        elif base in FAST and not op.endswith(("_sdwa", "_dpp")):     # (every SDWA / DPP form runs at the slow rate)
            c["valu_fast"] += 1
        else:
            c["valu_slow"] += 1
    elif op.startswith("ds_"):
        c["lds"] += 1
    elif op.startswith(("global_", "scratch_", "buffer_", "flat_")):
        c["vmem"] += 1
    elif op.startswith("s_"):
        c["salu"] += 1
print("kernel %s %s" % (m.group(1), " ".join(defs)))
print("%-10s %6s %6s %6s %5s %5s %5s %9s" % ("phase", "VALU", "fast", "slow", "SALU", "LDS", "VMEM", "VALU cyc"))
tot = collections.Counter()
tile = collections.Counter()
for ph, c in cnt.items():
    cyc = 1.96 * c["valu_fast"] + 3.25 * c["valu_slow"]
    print("%-10s %6d %6d %6d %5d %5d %5d %9.0f" % (ph, c["valu_fast"] + c["valu_slow"], c["valu_fast"], c["valu_slow"], c["salu"], c["lds"], c["vmem"], cyc))
    if ph != "prologue":
        tile.update(c)
        tile["cyc"] += cyc
print("%-10s %6d %6d %6d %5d %5d %5d %9.0f   (per wave-tile of 2048 bytes: %.2f VALU instructions per input byte)" % (
    "tile", tile["valu_fast"] + tile["valu_slow"], tile["valu_fast"], tile["valu_slow"], tile["salu"], tile["lds"], tile["vmem"], tile["cyc"],
    (tile["valu_fast"] + tile["valu_slow"]) / 2048.0))
