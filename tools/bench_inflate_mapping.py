#!/usr/bin/env python3
"""lane-per-stream vs wave-per-stream vs 16-lanes-per-stream inflate as a function of the batch size (stock-zlib streams of 2 KiB, Z_FIXED or -- argument
`default` -- dynamic trees): where is the crossover that HDLZ_INFLATE_WAVE_THRESHOLD encodes?  Optional 2nd argument: block size."""
import sys, os, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hdl_deflate_amd import Engine, INFLATE_LANE_PER_STREAM, INFLATE_WAVE_PER_STREAM, INFLATE_GROUP_PER_STREAM
from hdl_deflate_amd.data import make_blocks
e = Engine()
dyn = len(sys.argv) > 1 and sys.argv[1] == "default"
own = len(sys.argv) > 1 and sys.argv[1] == "own"      # the streams STARTC writes (CWINDOW 32): near matches only
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
h = make_blocks(4096, n, "cuda", seed=4, families=(1, 2, 4)).cpu().numpy()
zs = []
if own:
    zo, zl, _ = e.compress_batch(torch.from_numpy(h).cuda())
    zo, zl = zo.cpu().numpy(), zl.cpu().numpy()
    zs = [zo[k, :zl[k]].tobytes() for k in range(4096)]
else:
    for k in range(4096):
        c = zlib.compressobj(6, zlib.DEFLATED, 15, 9, zlib.Z_DEFAULT_STRATEGY if dyn else zlib.Z_FIXED)
        zs.append(c.compress(h[k].tobytes()) + c.flush())
sizes = [int(x) for x in os.environ.get("SIZES", "256,1024,2048,4096,8192,16384,24576,32768,49152,65536,131072,262144").split(",")]
for B in sizes:
    reps = (B + 4095) // 4096
    sel = (zs * reps)[:B]
    lens = np.array([len(z) for z in sel], dtype=np.int64)
    off = np.zeros(B + 1, np.int64); np.cumsum(lens, out=off[1:])
    zin = torch.from_numpy(np.frombuffer(b"".join(sel) + bytes(64), dtype=np.uint8).copy()).cuda()
    zoff = torch.from_numpy(off).cuda()
    line = "%7d streams:" % B
    for name, fl in (("lane", INFLATE_LANE_PER_STREAM), ("wave", INFLATE_WAVE_PER_STREAM), ("group", INFLATE_GROUP_PER_STREAM), ("auto", 0)):
        fn = lambda: e.inflate_batch(zin, in_off=zoff, out_pitch=n, flags=fl)
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): back, bl, bs = fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        assert int((bs != 0).sum()) == 0 and torch.equal(back[:4096 if B >= 4096 else B], torch.from_numpy(h[:min(B, 4096)]).cuda())
        line += "  %s %8.3f ms %7.1f GB/s" % (name, ms, B * n / ms / 1e6)
    print(line, flush=True)
