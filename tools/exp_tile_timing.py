#!/usr/bin/env python3
"""Where a tile of k_compress spends its wave time (a -DHDLZ_TILE_TIMING build: s_memtime at the phase boundaries; the j-th block a
wave processed reports the wave's total of part j in out_len).  Wave time is NOT VALU-pipe time: five waves share a SIMD, so a part
whose share of the wave time exceeds its share of the priced VALU cycles (tools/phase_count.py) is where waves WAIT.
usage: HDLZ_LIB=.../libhdlz_tiletime.so tools/exp_tile_timing.py [blocks] [block bytes] [cwindow]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hdl_deflate_amd import Engine
from hdl_deflate_amd.data import make_blocks
e = Engine()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
cw = int(sys.argv[3]) if len(sys.argv) > 3 else 32
d = make_blocks(B, n, "cuda", seed=1)
for rep in range(3):
    zo, zl, st = e.compress_batch(d, cwindow=cw, maxmatch=10)
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record(); zo, zl, st = e.compress_batch(d, cwindow=cw, maxmatch=10); ev1.record(); torch.cuda.synchronize()
ms = ev0.elapsed_time(ev1)
ncu = torch.cuda.get_device_properties(0).multi_processor_count
G = min(B, ncu * 256)
v = zl.cpu().numpy().astype(np.uint32)[: 15 * G].reshape(15, G).astype(np.float64)      # [part][wave]
names = ["block prologue", "stage (HBM wait, LDS fill)", "search", "adler", "extend", "parse", "chain", "codes", "scan", "scatter",
         "first prologue of the wave", "flush"]
tiles = v[12]
tot = v[:12].sum(axis=0)
print("%d blocks of %d bytes, CWINDOW %d: %d waves, %.0f tiles per wave, launch %.3f ms (timing build)" % (B, n, cw, G, tiles.mean(), ms))
for k in range(12):
    print("  %-28s %9.0f cycles per tile  %5.1f %%" % (names[k], (v[k] / tiles).mean(), 100 * v[k].sum() / tot.sum()))
print("  total %.0f wave cycles per tile; per wave %.3f M cycles (slowest %.3f M, fastest %.3f M)" % (
    (tot / tiles).mean(), tot.mean() / 1e6, tot.max() / 1e6, tot.min() / 1e6))
print("  wave lifetime %.3f M ticks (mean); sum of lifetimes / (SIMDs x launch ticks) = %.2f waves resident on average" % (
    v[13].mean() / 1e6, v[13].sum() / (4 * ncu * ms * 1e6 * (v[13] / v[14]).mean() * 0.1)))
print("  s_memtime ticks per wave lifetime / s_memrealtime ticks (100 MHz): the counter runs at %.3f GHz; launch %.3f ms = %.0f ticks per SIMD" % (
    (v[13] / v[14]).mean() * 0.1, ms, ms * 1e6 * (v[13] / v[14]).mean() * 0.1))
from hdl_deflate_amd.data import make_blocks as _mb
