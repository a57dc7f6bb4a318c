#!/usr/bin/env python3
"""Share of a kernel's VALU instructions that issue at the full rate (1.96 shader cycles per wave64 instruction: v_add/sub/and/or/xor/
lshrrev/ashrrev/mov/not/min_u16/max_i16/bitop3 in their plain encodings) -- everything else, every SDWA / DPP form and the lane
moves cost 3.25 (profiles/r04_ubench/ubench_valu_cycles*.txt).  STATIC counts from the ISA hipcc emits (hipcc -S): exact for the
straight-line compress tiles, an approximation for the inflate kernels (loops and wave-uniform branches weigh every instruction once).
usage: tools/valu_mix.py SRC.hip MANGLED_SUBSTRING [-DNAME ...]      (also imported by tools/update_traffic.py)"""
import os
import re
import subprocess
import sys

FAST = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshrrev_b32", "v_mov_b32", "v_min_u16",
        "v_add_co_u32", "v_not_b32", "v_ashrrev_i32", "v_bitop3_b32", "v_max_i16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_asm = {}


def mix(srcname, kern, defs=()):
    """-> (fast, slow) VALU instruction counts of the first kernel whose mangled name contains `kern`"""
    key = (srcname, tuple(defs))
    if key not in _asm:
        src = os.path.join(ROOT, "hdl_deflate_amd/csrc", srcname)
        _asm[key] = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", "-", src] +
                                   list(defs), capture_output=True, text=True).stdout
    m = re.search(r"^(_ZN4hdlz\w*%s\w*):[^\n]*\n(.*?)^\.Lfunc_end" % re.escape(kern), _asm[key], re.S | re.M)
    if not m:
        raise SystemExit("no kernel matching %s in %s" % (kern, srcname))
    fast = slow = 0
    for ln in m.group(2).splitlines():
        t = ln.strip()
        op = t.split()[0] if t and not t.startswith((";", ".", "//")) and not t.endswith(":") else None
        if not op or not op.startswith("v_"):
            continue
        base = re.sub(r"_e32$|_e64$|_sdwa$|_dpp$", "", op)
        if base in FAST and not op.endswith(("_sdwa", "_dpp")):
            fast += 1
        else:
            slow += 1
    return fast, slow


if __name__ == "__main__":
    f, s = mix(sys.argv[1], sys.argv[2], [a for a in sys.argv[3:] if a.startswith("-D")])
    print("%s %s: %d VALU instructions, %d full rate (%.1f %%), %d at 3.25 cycles -> %.3f cycles per instruction" % (
        sys.argv[1], sys.argv[2], f + s, f, 100.0 * f / (f + s), s, (1.96 * f + 3.25 * s) / (f + s)))
