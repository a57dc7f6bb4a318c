#!/usr/bin/env python3
"""Where a round of k_inflate_tok spends its time (a -DHDLZ_TOK_TIMING build: s_memtime per part, reported in out_len / status of lanes 0..7
of every wave).  usage: HDLZ_LIB=.../libhdlz_timing.so tools/exp_tok_timing.py <streams> [block] [own|zfixed|zdefault]"""
import sys, os, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hdl_deflate_amd import Engine
from hdl_deflate_amd.data import make_blocks
e = Engine()
B = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048; kind = sys.argv[3] if len(sys.argv) > 3 else "zfixed"
fam = tuple(int(x) for x in os.environ.get("FAM", "1,2,3,4").split(","))
nb = min(B, 4096)
d = make_blocks(nb, n, "cuda", seed=4, families=fam)
h = d.cpu().numpy()
if kind == "own":
    zo, zl, st = e.compress_batch(d, cwindow=32, maxmatch=10)
    zo, zl = zo.cpu().numpy(), zl.cpu().numpy()
    zs = [zo[k, :zl[k]].tobytes() for k in range(nb)]
else:
    zs = []
    for k in range(nb):
        c = zlib.compressobj(6, zlib.DEFLATED, 15, 9, zlib.Z_FIXED if kind == "zfixed" else zlib.Z_DEFAULT_STRATEGY)
        zs.append(c.compress(h[k].tobytes()) + c.flush())
sel = (zs * ((B + len(zs) - 1) // len(zs)))[:B]
lens = np.array([len(z) for z in sel], dtype=np.int64)
off = np.zeros(B + 1, np.int64); np.cumsum(lens, out=off[1:])
zin = torch.from_numpy(np.frombuffer(b"".join(sel) + bytes(64), dtype=np.uint8).copy()).cuda()
zoff = torch.from_numpy(off).cuda()
out = torch.empty((B, n), dtype=torch.uint8, device="cuda")
flags = 1 | 2 if kind != "zdefault" else 2
for rep in range(2):
    back, bl, bs = e.inflate_batch(zin, in_off=zoff, out_pitch=n, flags=flags, out=out)
torch.cuda.synchronize()
v = (bl.cpu().numpy().astype(np.uint64) | (bs.cpu().numpy().astype(np.uint64) << np.uint64(32))).reshape(-1, 64)[:, :8].astype(np.float64)
names = ["move loop exit", "refill", "decode", "slow", "epilogue", "loop top", "move iterations + flush", "rounds"]
rounds = v[:, 7]
print("%d streams of %d (%s, families %s): %d waves, rounds per wave mean %.0f max %.0f" % (B, n, kind, fam, v.shape[0], rounds.mean(), rounds.max()))
tot = v[:, :7].sum(axis=1)
for k in (5, 6, 0, 1, 2, 3, 4):
    print("  %-24s %10.0f cycles per wave  %6.1f per round  %5.1f %%" % (names[k], v[:, k].mean(), (v[:, k] / rounds).mean(), 100 * v[:, k].sum() / tot.sum()))
print("  total %.0f cycles per wave = %.0f per round" % (tot.mean(), (tot / rounds).mean()))
