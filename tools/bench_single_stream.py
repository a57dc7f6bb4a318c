#!/usr/bin/env python3
"""STARTD on ONE large stream: a stream written by compress_stream (one fixed block, as STARTC's) of 1 / 16 / 256 MiB through
hdlz_inflate_batch(nstreams = 1) -- the parallel path of hdlz_inflate_par.hip -- and, for the small ones, through the forced
wave-per-stream decoder (what every single stream got before).  Usage: tools/bench_single_stream.py [MiB ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hdl_deflate_amd
from hdl_deflate_amd.data import make_blocks

e = hdl_deflate_amd.Engine()
for nmb in [int(x) for x in sys.argv[1:]] or [1, 16, 256]:
    n = nmb << 20
    d = make_blocks(n // 2048, 2048, "cuda", seed=5).reshape(-1)
    out, ol, st = e.compress_stream(d, n)
    torch.cuda.synchronize()
    z = out[:int(ol.item())].clone()
    zin = torch.cat([z, torch.zeros(64, dtype=torch.uint8, device="cuda")]).reshape(1, -1)
    back = torch.empty((1, n), dtype=torch.uint8, device="cuda")
    line = "%4d MiB stream (%d bytes compressed):" % (nmb, z.numel())
    for name, flags in (("parallel", 1), ("one wave", 1 | 4)):
        if flags & 4 and nmb > 16:
            continue
        fn = lambda: e.inflate_batch(zin, in_len=z.numel(), out_pitch=n, flags=flags, out=back)
        _, bl, bs = fn()
        torch.cuda.synchronize()
        assert int(bs.item()) == 0 and int(bl.item()) == n and torch.equal(back.reshape(-1), d[:n])
        reps = 5 if not flags & 4 else 1
        t0 = time.time()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / reps
        line += "  %s %9.3f ms = %8.3f GB/s" % (name, dt * 1e3, n / dt / 1e9)
    print(line, flush=True)
