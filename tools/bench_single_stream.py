import sys, time, zlib, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hdl_deflate_amd
from hdl_deflate_amd.data import make_blocks
e = hdl_deflate_amd.Engine()
for nmb in (1, 16):
    n = nmb << 20
    d = make_blocks(n // 2048, 2048, "cuda", seed=5).reshape(-1)
    out, ol, st = e.compress_stream(d, n)
    torch.cuda.synchronize()
    z = out[:int(ol.item())].clone()
    zin = z.reshape(1, -1)
    for flags in (0,):
        fn = lambda: e.inflate_batch(zin, in_len=z.numel(), out_pitch=n, flags=flags | 1)
        back, bl, bs = fn(); torch.cuda.synchronize()
        assert int(bs.item()) == 0 and torch.equal(back.reshape(-1)[:n], d[:n])
        t0 = time.time(); back, bl, bs = fn(); torch.cuda.synchronize(); dt = time.time() - t0
        print("%d MiB single fixed stream: inflate %.1f ms = %.3f GB/s (compress_stream out %d)" % (nmb, dt * 1e3, n / dt / 1e9, z.numel()))
