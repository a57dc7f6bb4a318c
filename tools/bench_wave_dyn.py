#!/usr/bin/env python3
"""wave-per-stream inflate (k_inflate_dyn) on B stock-zlib default-strategy streams of n bytes: ms and GB/s of output.
Usage: tools/bench_wave_dyn.py [B=131072] [n=2048] [check=1]   (HDLZ_LIB selects an A/B build; check=0 for timing builds that
produce wrong bytes on purpose)"""
import sys, os, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hdl_deflate_amd import Engine, INFLATE_WAVE_PER_STREAM
from hdl_deflate_amd.data import make_blocks
e = Engine()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
check = int(sys.argv[3]) if len(sys.argv) > 3 else 1
h = make_blocks(4096, n, "cuda", seed=4, families=(1, 2, 4)).cpu().numpy()
zs = []
for k in range(4096):
    c = zlib.compressobj(6, zlib.DEFLATED, 15, 9, zlib.Z_DEFAULT_STRATEGY)
    zs.append(c.compress(h[k].tobytes()) + c.flush())
sel = (zs * ((B + 4095) // 4096))[:B]
off = np.zeros(B + 1, np.int64); np.cumsum([len(z) for z in sel], out=off[1:])
zin = torch.from_numpy(np.frombuffer(b"".join(sel) + bytes(64), dtype=np.uint8).copy()).cuda()
zoff = torch.from_numpy(off).cuda()
fn = lambda: e.inflate_batch(zin, in_off=zoff, out_pitch=n, flags=INFLATE_WAVE_PER_STREAM)
back, bl, bs = fn(); torch.cuda.synchronize()
if check:
    assert int((bs != 0).sum()) == 0
    hb = back.cpu().numpy()
    for k in range(0, B, max(1, B // 512)):
        assert hb[k].tobytes() == h[k % 4096].tobytes(), k
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print("%s: %d streams x %d B, wave per stream: %.3f ms %.1f GB/s%s" % (os.environ.get("HDLZ_LIB", "libhdlz.so").split("/")[-1], B, n, ms, B * n / ms / 1e6, "" if check else " (unchecked)"))
