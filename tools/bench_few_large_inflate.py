#!/usr/bin/env python3
"""hdlz_inflate_batch on a FEW LARGE streams (fixed pitch): the whole-GPU path over all of them (k_par_*, blockIdx.y = the stream) against the
batch kernels (every stream a serial chain).  SHAPES=512x16,2048x64 (streams x KiB) overrides the list; the limit of the path:
HDLZ_LIB=... variants built with -DHDLZ_PAR_BATCH_MAX=n."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hdl_deflate_amd import Engine
from hdl_deflate_amd.data import make_blocks
e = Engine()
SHAPES = [tuple(int(v) for v in x.split('x')) for x in os.environ['SHAPES'].split(',')] if os.environ.get('SHAPES') else None
for nb, kib in SHAPES or ((16, 64), (32, 64), (64, 64), (128, 64), (256, 64), (64, 256), (256, 256), (512, 256), (1024, 256), (256, 1024), (1024, 1024)):
    n = kib << 10
    d = make_blocks(nb * (n // 2048), 2048, "cuda", seed=2).reshape(nb, n)
    if os.environ.get("ZLIB"):            # stock zlib streams (dynamic trees) in rows of one pitch instead of our own
        import zlib, numpy as np
        zs = [zlib.compress(r.tobytes(), 6) for r in d.cpu().numpy()]
        pitch = (max(len(z) for z in zs) + 16 + 15) // 16 * 16
        h = np.zeros((nb, pitch), np.uint8)
        for k, z in enumerate(zs):
            h[k, : len(z)] = np.frombuffer(z, np.uint8)
        zo = torch.from_numpy(h).cuda()
    else:
        zo, zl, st = e.compress_batch(d)
        assert int(st.max().item()) == 0
    res = []
    ragged = os.environ.get("RAGGED") and not os.environ.get("ZLIB")      # the streams back to back (an archive) + the caller's bound on their lengths
    if ragged:
        arc, aoff = e.archive(zo, zl)
        bound = zo.shape[1]
    for label, fl in (("auto", 0), ("lane", 2), ("wave", 4), ("group", 64)):
        f = (lambda: e.inflate_batch(arc, in_off=aoff, in_len=bound, out_pitch=n, flags=fl)) if ragged else (lambda: e.inflate_batch(zo, out_pitch=n, flags=fl))
        back, bl, bs = f(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            back, bl, bs = f()
        b.record(); torch.cuda.synchronize()
        ok = int(bs.max().item()) == 0 and torch.equal(back[:, :n], d)
        res.append("%s %8.3f ms%s" % (label, a.elapsed_time(b) / 3, "" if ok else " MISMATCH"))
    print("%5d x %5d KiB (pitch %d):  %s" % (nb, kib, zo.shape[1], "  ".join(res)), flush=True)
    del d, zo, back
    torch.cuda.empty_cache()
