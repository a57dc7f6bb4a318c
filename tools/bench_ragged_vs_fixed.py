import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from hdl_deflate_amd import Engine
from hdl_deflate_amd.data import make_blocks
e = Engine()
for B, n in ((1 << 19, 2048), (1 << 15, 65536)):
    d = make_blocks(B, n, "cuda", seed=1)
    off = torch.arange(B + 1, dtype=torch.int64, device="cuda") * n
    flat = d.reshape(-1)
    out = torch.empty((B, (6 + (9 * n + 17) // 8 + 15) // 16 * 16), dtype=torch.uint8, device="cuda")
    for name, fn in (("fixed pitch", lambda: e.compress_batch(d, out=out, out_pitch=out.shape[1])),
                     ("ragged in_off", lambda: e.compress_batch(flat, in_off=off, max_len=n, out=out, out_pitch=out.shape[1]))):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): r = fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("%d x %d %-14s %.3f ms %.1f GB/s" % (B, n, name, ms, B * n / ms / 1e6))
