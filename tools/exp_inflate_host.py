#!/usr/bin/env python3
"""Engine.inflate_host on the configs[3] job (2^20 Z_FIXED streams of 2 KiB blocks) for several chunk sizes; prints ms per job.
usage: tools/exp_inflate_host.py [streams]"""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import multiprocessing as mp
import hdl_deflate_amd
from hdl_deflate_amd.data import make_blocks
from bench import _zfixed_chunk, median

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
n = 2048
dev = torch.device("cuda", 0)
eng = hdl_deflate_amd.Engine(dev)
d_plain = make_blocks(B, n, dev, seed=4, families=(1, 2, 4))
host = d_plain.cpu().numpy()
nproc = min(os.cpu_count() or 1, 64)
per = (B + nproc * 4 - 1) // (nproc * 4)
with mp.get_context("fork").Pool(nproc) as pool:
    parts = pool.map(_zfixed_chunk, [(host[k:k + per].tobytes(), n, "fixed") for k in range(0, B, per)])
lens = np.fromiter((l for _, ls in parts for l in ls), dtype=np.int64, count=B)
off = np.zeros(B + 1, np.int64)
np.cumsum(lens, out=off[1:])
flat = np.frombuffer(b"".join(p for p, _ in parts) + bytes(64), dtype=np.uint8)
h_z = torch.empty(flat.size, dtype=torch.uint8, pin_memory=True)
h_z.copy_(torch.from_numpy(flat.copy()))
h_rows = torch.empty((B, n), dtype=torch.uint8, pin_memory=True)
h_l = torch.empty(B, dtype=torch.int32, pin_memory=True)
h_s = torch.empty(B, dtype=torch.int32, pin_memory=True)
h_off = torch.from_numpy(off)
plain_cpu = d_plain.cpu()
for mib in (128, 256, 512):
    C = (mib << 20) // n
    ts = []
    h_rows.zero_()
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.inflate_host(h_z, h_off, n, flags=1, chunk_streams=C, h_out=h_rows, h_len=h_l, h_status=h_s, d2h=os.environ.get("D2H", "copy"))
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    ok = int((h_s != 0).sum()) == 0 and torch.equal(h_rows, plain_cpu)
    print("chunk %4d MiB of rows (%7d streams): %s ms   median %.2f   ok %s" % (mib, C, " ".join("%.1f" % t for t in ts), median(ts[1:]), ok), flush=True)
    eng.release_host_buffers()
