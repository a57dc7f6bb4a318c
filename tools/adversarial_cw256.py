#!/usr/bin/env python3
"""worst cases for the hashed match finder (CWINDOW = 256, k_compress<8>): inputs whose positions all fall into a few hash classes
(zeros, short periods: 64 lanes on one table address) or into none twice (random bytes), next to text; every block against the oracle.
Usage: tools/adversarial_cw256.py [blocks]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, hdl_deflate_amd
from hdl_deflate_amd.data import make_text_blocks
from oracle import oracle as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = 65536
e = hdl_deflate_amd.Engine()
g = torch.Generator(device="cuda"); g.manual_seed(5)
cases = {"text": make_text_blocks(B, n, "cuda", seed=1),
         "zeros": torch.zeros((B, n), dtype=torch.uint8, device="cuda"),
         "period 2": torch.arange(n, device="cuda").remainder(2).to(torch.uint8).add(65).repeat(B, 1),
         "period 7": torch.arange(n, device="cuda").remainder(7).to(torch.uint8).add(65).repeat(B, 1),
         "period 300": torch.arange(n, device="cuda").remainder(300).remainder(251).to(torch.uint8).repeat(B, 1),
         "random bytes": torch.randint(0, 256, (B, n), generator=g, device="cuda", dtype=torch.uint8),
         "random 0/1": torch.randint(0, 2, (B, n), generator=g, device="cuda", dtype=torch.uint8) + 48}
for name, d in cases.items():
    for cw in (256, 32):
        out, ol, st = e.compress_batch(d, cwindow=cw)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(3):
            e.compress_batch(d, cwindow=cw, out=out)
        ev1.record(); torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / 3
        ok = True
        for b in (0, B // 2, B - 1):
            rc, ref = O.compress(d[b].cpu().numpy().tobytes(), cw, 10)
            ok &= rc == 0 and out[b, :int(ol[b])].cpu().numpy().tobytes() == ref
        print("%-14s CWINDOW=%-3d %8.3f ms  %7.1f GB/s  ratio %.4f  %s" % (name, cw, ms, B * n / ms / 1e6, float(ol.sum()) / (B * n), "= oracle" if ok else "MISMATCH"), flush=True)
