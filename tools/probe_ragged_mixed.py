#!/usr/bin/env python3
"""RAGGED batches of mixed block sizes (in_off form), 256 MiB per case: log-uniform 5 B .. 200 KB, 5 B .. 1000 B, 1 .. 8 KiB, 40 .. 64 KiB
-- through hdlz_compress_batch and back through hdlz_inflate_batch (ragged input, fixed output pitch).  Round trip checked."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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


rng = np.random.default_rng(3)
TOTAL = 256 << 20
src = make_blocks((TOTAL + (1 << 20)) // 2048, 2048, "cuda", seed=8).reshape(-1)


def lens_of(kind):
    out, tot = [], 0
    while tot < TOTAL:
        if kind == "log 5 B .. 200 KB":
            n = int(np.exp(rng.uniform(np.log(5), np.log(200000))))
        elif kind == "5 .. 1000 B":
            n = int(rng.integers(5, 1001))
        elif kind == "1 .. 8 KiB":
            n = int(rng.integers(1024, 8193))
        elif kind == "40 .. 64 KiB":
            n = int(rng.integers(40960, 65537))
        else:
            raise ValueError(kind)
        out.append(n); tot += n
    return np.array(out, dtype=np.int64)


for kind in ("log 5 B .. 200 KB", "5 .. 1000 B", "1 .. 8 KiB", "40 .. 64 KiB"):
    lens = lens_of(kind)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    total = int(off[-1])
    d_in = src[: total + 64].contiguous()
    d_off = torch.from_numpy(off).cuda()
    mx = int(lens.max())
    for cw in (32, 256):
        ms_c, (zo, zl, st) = timed(lambda: e.compress_batch(d_in, in_off=d_off, cwindow=cw, max_len=mx), reps=3)
        assert int(st.max().item()) == 0
        # the compressed rows back as a ragged batch: an archive + offsets
        arch, aoff = e.archive(zo, zl)
        cap = (mx + 15) // 16 * 16
        if len(lens) * cap > (12 << 30):
            print("%-30s cw %3d: %8d blocks, compress %8.3f ms %6.1f GB/s (inflate skipped: %d x %d B of output rows)" % (kind, cw, len(lens), ms_c, total / ms_c / 1e6, len(lens), cap), flush=True)
            continue
        ms_i, (back, bl, bs) = timed(lambda: e.inflate_batch(arch, in_off=aoff, out_pitch=cap), reps=3)
        hl = bl.cpu().numpy()
        ok = int(bs.max().item()) == 0 and (hl == lens).all()
        k = int(np.argmax(lens))
        ok = ok and torch.equal(back[k, :lens[k]], d_in[off[k]: off[k + 1]]) and torch.equal(back[0, :lens[0]], d_in[: lens[0]])
        print("%-30s cw %3d: %8d blocks, compress %8.3f ms %6.1f GB/s  inflate %8.3f ms %6.1f GB/s%s" % (kind, cw, len(lens), ms_c, total / ms_c / 1e6, ms_i, total / ms_i / 1e6, "" if ok else " MISMATCH"), flush=True)
        del zo, zl, st, arch, aoff
        torch.cuda.empty_cache()
