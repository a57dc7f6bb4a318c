#!/usr/bin/env python3
"""inflate throughput on the engine's OWN compressed output (CWINDOW-limited distances: no far copies) next to
stock zlib Z_FIXED streams of the same blocks (distances up to the block size) -- isolates the far-copy cost"""
import sys, os, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hdl_deflate_amd import Engine, INFLATE_ASSUME_FIXED
from hdl_deflate_amd.data import make_blocks
e = Engine()
B, n = 1 << 19, 2048
d = make_blocks(B, n, "cuda", seed=4, families=(1, 2, 4))

def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): r = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 3, r

for cw in (32, 256):
    out, ol, st = e.compress_batch(d, cwindow=cw)
    arch, offs = e.compact(out, ol)
    in_off = torch.cat([offs, (offs[-1:] + ol[-1:].to(torch.int64))])
    arch = torch.cat([arch, torch.zeros(64, dtype=torch.uint8, device="cuda")])
    ms, (back, bl, bs) = timeit(lambda: e.inflate_batch(arch, in_off=in_off, out_pitch=n, flags=INFLATE_ASSUME_FIXED))
    assert os.environ.get("HDLZ_NOCHECK") or (int((bs != 0).sum()) == 0 and torch.equal(back, d))
    print("own CW%-3d streams: %.3f ms  %.1f GB/s out  (ratio %.3f)" % (cw, ms, B * n / ms / 1e6, float(ol.sum()) / (B * n)))

# stock zlib Z_FIXED of a sample, tiled
h = d[:4096].cpu().numpy()
zs = []
for k in range(4096):
    c = zlib.compressobj(6, zlib.DEFLATED, 15, 9, zlib.Z_FIXED)
    zs.append(c.compress(h[k].tobytes()) + c.flush())
reps = B // 4096
lens = np.array([len(z) for z in zs] * reps, dtype=np.int64)
off = np.zeros(B + 1, np.int64); np.cumsum(lens, out=off[1:])
flat = np.frombuffer(b"".join(zs) * reps + bytes(64), dtype=np.uint8).copy()
zin, zoff = torch.from_numpy(flat).cuda(), torch.from_numpy(off).cuda()
ms, (back, bl, bs) = timeit(lambda: e.inflate_batch(zin, in_off=zoff, out_pitch=n, flags=INFLATE_ASSUME_FIXED))
assert int((bs != 0).sum()) == 0
print("zlib Z_FIXED streams: %.3f ms  %.1f GB/s out  (ratio %.3f)" % (ms, B * n / ms / 1e6, float(lens.sum()) / (B * n)))
