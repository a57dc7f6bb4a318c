#!/usr/bin/env python3
"""compress throughput of 2^18 x 2 KiB and 4096 x 64 KiB blocks (families data) over CWINDOW and MAXMATCH values off the bench's: looks for
windows that fall off the curve (FULLWIN / non-FULLWIN template variants, the finder of the wide windows).  Round trip of one block checked."""
import sys, os, zlib
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


for n, nb in ((2048, 1 << 18), (65536, 4096), (256, 1 << 20)):
    d = make_blocks(nb * n // 2048 if n >= 2048 else nb, 2048, "cuda", seed=3)
    d = d.reshape(nb, n) if n >= 2048 else d[:, :n].contiguous()
    total = nb * n
    line = "%7d x %6d B |" % (nb, n)
    for cw, mm in ((32, 10), (32, 5), (31, 10), (16, 10), (5, 10), (1, 10), (33, 10), (48, 10), (64, 10), (65, 10), (100, 10), (128, 10), (200, 10), (255, 10), (256, 10), (256, 5)):
        ms, (zo, zl, st) = timed(lambda: e.compress_batch(d, cwindow=cw, maxmatch=mm))
        assert int(st.max().item()) == 0
        k = nb - 1
        ok = zlib.decompress(zo[k, : int(zl[k].item())].cpu().numpy().tobytes()) == d[k].cpu().numpy().tobytes()
        line += " %d/%d: %5.1f%s" % (cw, mm, total / ms / 1e6, "" if ok else "!")
    print(line, flush=True)
