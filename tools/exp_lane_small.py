#!/usr/bin/env python3
"""k_inflate_tok on a SMALL batch (latency regime): usage tools/exp_lane_small.py <streams> [block] [reps]   (Z_FIXED streams, lane mapping)"""
import sys, os, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hdl_deflate_amd import Engine
from hdl_deflate_amd.data import make_blocks
e = Engine()
B = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
fam = tuple(int(x) for x in os.environ.get("FAM", "1,2,4").split(","))
h = make_blocks(min(B, 4096), n, "cuda", seed=4, families=fam).cpu().numpy()
zs = []
for k in range(h.shape[0]):
    c = zlib.compressobj(6, zlib.DEFLATED, 15, 9, zlib.Z_FIXED)
    zs.append(c.compress(h[k].tobytes()) + c.flush())
sel = (zs * ((B + len(zs) - 1) // len(zs)))[:B]
lens = np.array([len(z) for z in sel], dtype=np.int64)
off = np.zeros(B + 1, np.int64); np.cumsum(lens, out=off[1:])
zin = torch.from_numpy(np.frombuffer(b"".join(sel) + bytes(64), dtype=np.uint8).copy()).cuda()
zoff = torch.from_numpy(off).cuda()
out = torch.empty((B, n), dtype=torch.uint8, device="cuda")
fn = lambda: e.inflate_batch(zin, in_off=zoff, out_pitch=n, flags=1 | 2, out=out)
fn(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): back, bl, bs = fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
assert int((bs != 0).sum()) == 0
print("%d streams of %d: %.3f ms  %.1f GB/s" % (B, n, ms, B * n / ms / 1e6))
