#!/usr/bin/env python3
"""ONE stream through hdlz_compress_stream (k_stream_*) and back through hdlz_inflate_batch (k_par_*), over sizes that are not the bench's
16 MiB and both windows: looks for sizes that fall off the curve.  Round trip checked.  usage: tools/probe_stream_sizes.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hdl_deflate_amd import Engine
from hdl_deflate_amd.data import make_blocks
e = Engine()


def timed(f, reps=4):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        r = f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, r


src = make_blocks((260 << 20) // 2048, 2048, "cuda", seed=4).reshape(-1)
for n in (5, 100, 2047, 5000, 65536, 100000, 300001, 1 << 20, 3000000, 5 << 20, 16 << 20, 50000000, 100 << 20, (256 << 20) - 64):
    d = src[: n + 16].clone(); d[n:] = 0
    line = "%10d B |" % n
    for cw in (32, 256):
        ms_c, (zo, zl, st) = timed(lambda: e.compress_stream(d, n, cwindow=cw))
        assert int(st.item()) == 0, (n, cw, int(st.item()))
        zn = int(zl.item())
        zin = zo[:zn].reshape(1, zn).contiguous()
        cap = (n + 64 + 15) // 16 * 16
        ms_i, (back, bl, bs) = timed(lambda: e.inflate_batch(zin, out_pitch=cap))
        ok = int(bs[0].item()) == 0 and int(bl[0].item()) == n and torch.equal(back[0, :n], d[:n])
        line += " cw %3d: STARTC %7.3f ms %6.1f GB/s  STARTD %7.3f ms %6.1f GB/s%s |" % (cw, ms_c, n / ms_c / 1e6, ms_i, n / ms_i / 1e6, "" if ok else " MISMATCH")
    print(line, flush=True)
