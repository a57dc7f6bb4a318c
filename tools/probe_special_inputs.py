#!/usr/bin/env python3
"""Throughput of the batch paths on inputs the bench families do not contain: zeros, a period-3 pattern, random bytes, 2 KiB blocks
(and the same as 64 KiB blocks).  Round trip checked.  usage: tools/probe_special_inputs.py [blocks]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hdl_deflate_amd import Engine
e = Engine()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144


def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        r = f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, r


for n, nb in ((2048, B), (65536, B // 32)):
    total = n * nb
    gens = {"zeros": lambda: torch.zeros(total, dtype=torch.uint8, device="cuda"),
            "period3": lambda: (torch.arange(total, device="cuda") % 3 + 65).to(torch.uint8),
            "random": lambda: torch.randint(0, 256, (total,), dtype=torch.uint8, device="cuda"),
            "ramp": lambda: (torch.arange(total, device="cuda") % 251).to(torch.uint8)}
    if len(sys.argv) > 2 and sys.argv[2] == "kinds":
        # kinds of real data (16 MiB made on the host, tiled: CWINDOW = 32 does not see the repeat)
        import random, base64
        import numpy as np
        r = random.Random(77)
        m = 16 << 20
        def logs():
            out = bytearray(); t = 1700000000
            hosts = ["web-%02d" % i for i in range(12)]; paths = ["/api/v1/items/%d" % i for i in range(40)] + ["/index.html", "/static/app.js", "/health"]
            while len(out) < m:
                t += r.randint(0, 3)
                out += ("%d %s GET %s %d %d \"Mozilla/5.0 (X11; Linux x86_64)\" rt=%.3f\n" % (t, r.choice(hosts), r.choice(paths), r.choice([200, 200, 200, 304, 404, 500]), r.randint(100, 90000), r.random())).encode()
            return bytes(out[:m])
        def words():
            ws = [bytes(r.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(r.randint(2, 9))) for _ in range(3000)]
            out = bytearray()
            while len(out) < m:
                out += r.choice(ws) + b" "
            return bytes(out[:m])
        host = {"logs": logs, "words": words,
                "dna": lambda: np.random.default_rng(3).choice(np.frombuffer(b"ACGT", dtype=np.uint8), m).tobytes(),
                "sparse": lambda: (np.random.default_rng(4).integers(1, 256, m, dtype=np.uint8) * (np.random.default_rng(5).random(m) < 1 / 64)).astype(np.uint8).tobytes(),
                "float32": lambda: np.cumsum(np.random.default_rng(6).normal(size=m // 4)).astype(np.float32).tobytes(),
                "base64": lambda: base64.b64encode(np.random.default_rng(8).integers(0, 256, m, dtype=np.uint8).tobytes())[:m],
                "hex": lambda: np.random.default_rng(9).integers(0, 256, m // 2, dtype=np.uint8).tobytes().hex().encode()[:m]}
        def tiled(f):
            def g():
                t = torch.frombuffer(bytearray(f()), dtype=torch.uint8).cuda()
                return t.repeat((total + m - 1) // m)[:total].contiguous()
            return g
        gens = {k: tiled(f) for k, f in host.items()}
    for name, g in gens.items():
        d = g().reshape(nb, n)
        ms_c, (zo, zl, st) = timed(lambda: e.compress_batch(d, cwindow=32, maxmatch=10))
        assert int(st.max().item()) == 0
        line = "%6d x %5d B %-8s compress %7.3f ms %6.1f GB/s ratio %.3f |" % (nb, n, name, ms_c, total / ms_c / 1e6, float(zl.sum().item()) / total)
        for label, fl in (("auto", 0), ("lane", 2), ("wave", 4), ("group", 64)):
            if label in ("wave", "group") and nb > 65536:
                continue
            ms_i, (back, bl, bs) = timed(lambda: e.inflate_batch(zo, out_pitch=n, flags=fl), reps=3)
            ok = int(bs.max().item()) == 0 and torch.equal(back[:, :n], d)
            line += " %s %7.3f ms %6.1f GB/s%s" % (label, ms_i, total / ms_i / 1e6, "" if ok else " MISMATCH")
        print(line, flush=True)
        del d, zo
        torch.cuda.empty_cache()
