#!/usr/bin/env python3
"""Throughput of the batch paths on inputs the bench families do not contain: zeros, a period-3 pattern, random bytes, 2 KiB blocks
(and the same as 64 KiB blocks).  Round trip checked.  usage: tools/probe_special_inputs.py [blocks]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hdl_deflate_amd import Engine
e = Engine()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144


def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        r = f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, r


for n, nb in ((2048, B), (65536, B // 32)):
    total = n * nb
    gens = {"zeros": lambda: torch.zeros(total, dtype=torch.uint8, device="cuda"),
            "period3": lambda: (torch.arange(total, device="cuda") % 3 + 65).to(torch.uint8),
            "random": lambda: torch.randint(0, 256, (total,), dtype=torch.uint8, device="cuda"),
            "ramp": lambda: (torch.arange(total, device="cuda") % 251).to(torch.uint8)}
    for name, g in gens.items():
        d = g().reshape(nb, n)
        ms_c, (zo, zl, st) = timed(lambda: e.compress_batch(d, cwindow=32, maxmatch=10))
        assert int(st.max().item()) == 0
        line = "%6d x %5d B %-8s compress %7.3f ms %6.1f GB/s ratio %.3f |" % (nb, n, name, ms_c, total / ms_c / 1e6, float(zl.sum().item()) / total)
        for label, fl in (("auto", 0), ("lane", 2), ("wave", 4), ("group", 64)):
            if label in ("wave", "group") and nb > 65536:
                continue
            ms_i, (back, bl, bs) = timed(lambda: e.inflate_batch(zo, out_pitch=n, flags=fl), reps=3)
            ok = int(bs.max().item()) == 0 and torch.equal(back[:, :n], d)
            line += " %s %7.3f ms %6.1f GB/s%s" % (label, ms_i, total / ms_i / 1e6, "" if ok else " MISMATCH")
        print(line, flush=True)
        del d, zo
        torch.cuda.empty_cache()
