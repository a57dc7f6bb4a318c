#!/usr/bin/env python3
"""compress / inflate throughput over block sizes that are NOT the bench's powers of two (256 MiB per shape, families data, CWINDOW 32 and 256):
looks for shapes that fall off the curve.  Round trip checked.  usage: tools/probe_block_sizes.py"""
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


src = make_blocks((272 << 20) // 2048, 2048, "cuda", seed=2).reshape(-1)
for n in (5, 17, 64, 100, 255, 256, 257, 300, 777, 1000, 2047, 2048, 2049, 3000, 4096, 5000, 8191, 10000, 20000, 40000, 65535, 65536, 100000, 1000000):
    nb = max(1, (256 << 20) // n)
    total = nb * n
    pitch = (n + 15) // 16 * 16
    d = torch.zeros((nb, pitch), dtype=torch.uint8, device="cuda")
    d[:, :n] = src[:total].reshape(nb, n)
    line = "%8d x %7d B |" % (nb, n)
    for cw in (32, 256):
        ms_c, (zo, zl, st) = timed(lambda: e.compress_batch(d, in_len=n, cwindow=cw, maxmatch=10))
        assert int(st.max().item()) == 0, (n, cw, int(st.max().item()))
        ms_i, (back, bl, bs) = timed(lambda: e.inflate_batch(zo, out_pitch=pitch))
        ok = int(bs.max().item()) == 0 and int(bl.min().item()) == n and torch.equal(back[:, :n], d[:, :n])
        line += " cw %3d: compress %7.3f ms %6.1f GB/s  inflate %7.3f ms %6.1f GB/s%s |" % (cw, ms_c, total / ms_c / 1e6, ms_i, total / ms_i / 1e6, "" if ok else " MISMATCH")
        del zo, zl, st, back
    print(line, flush=True)
    del d
    torch.cuda.empty_cache()
