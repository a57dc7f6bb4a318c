#!/usr/bin/env python3
"""Where a step of k_inflate_grp goes (a -DHDLZ_GRP_TIMING build: s_memtime per part; group g of every wave reports part g).
usage: HDLZ_LIB=.../libhdlz_grptime.so tools/exp_grp_timing.py <streams> [fixed|own]"""
import sys, os, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hdl_deflate_amd import Engine, INFLATE_GROUP_PER_STREAM
from hdl_deflate_amd.data import make_blocks
e = Engine()
B = int(sys.argv[1]); kind = sys.argv[2] if len(sys.argv) > 2 else "fixed"; n = 2048
h = make_blocks(min(B, 4096), n, "cuda", seed=4, families=(1, 2, 4)).cpu().numpy()
if kind == "own":
    zo, zl, _ = e.compress_batch(torch.from_numpy(h).cuda()); zo, zl = zo.cpu().numpy(), zl.cpu().numpy()
    zs = [zo[k, :zl[k]].tobytes() for k in range(len(h))]
else:
    zs = []
    for k in range(len(h)):
        c = zlib.compressobj(6, zlib.DEFLATED, 15, 9, zlib.Z_FIXED); zs.append(c.compress(h[k].tobytes()) + c.flush())
sel = (zs * ((B + len(zs) - 1) // len(zs)))[:B]
off = np.zeros(B + 1, np.int64); np.cumsum([len(z) for z in sel], out=off[1:])
zin = torch.from_numpy(np.frombuffer(b"".join(sel) + bytes(64), dtype=np.uint8).copy()).cuda()
for _ in range(2):
    back, bl, bs = e.inflate_batch(zin, in_off=torch.from_numpy(off).cuda(), out_pitch=n, flags=INFLATE_GROUP_PER_STREAM | 1)
torch.cuda.synchronize()
v = bl.cpu().numpy().astype(np.float64).reshape(-1, 4); w = bs.cpu().numpy().astype(np.float64).reshape(-1, 4)
steps = w[:, 0]
names = ["input + refill", "look-ups + decode", "slow path", "move", "flush + loop end"]
parts = [v[:, 0], v[:, 1], v[:, 2], v[:, 3], w[:, 1]]
tot = sum(p.sum() for p in parts)
print("%d %s streams: %d waves, steps per wave mean %.0f max %.0f" % (B, kind, len(steps), steps.mean(), steps.max()))
for nme, p in zip(names, parts):
    print("  %-20s %8.1f cycles per step  %5.1f %%" % (nme, (p / steps).mean(), 100 * p.sum() / tot))
print("  total %.0f cycles per step" % (sum((p / steps).mean() for p in parts)))
