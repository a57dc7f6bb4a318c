#!/usr/bin/env python3
"""one inflate_host job per chunk size given on the command line (for a rocprofv3 kernel trace)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, multiprocessing as mp
import hdl_deflate_amd
from hdl_deflate_amd.data import make_blocks
from bench import _zfixed_chunk
B, n = 1 << 19, 2048
dev = torch.device("cuda", 0)
eng = hdl_deflate_amd.Engine(dev)
d_plain = make_blocks(B, n, dev, seed=4, families=(1, 2, 4))
host = d_plain.cpu().numpy()
with mp.get_context("fork").Pool(32) as pool:
    per = B // 128
    parts = pool.map(_zfixed_chunk, [(host[k:k + per].tobytes(), n, "fixed") for k in range(0, B, per)])
lens = np.fromiter((l for _, ls in parts for l in ls), dtype=np.int64, count=B)
off = np.zeros(B + 1, np.int64); np.cumsum(lens, out=off[1:])
flat = np.frombuffer(b"".join(p for p, _ in parts) + bytes(64), dtype=np.uint8)
h_z = torch.empty(flat.size, dtype=torch.uint8, pin_memory=True); h_z.copy_(torch.from_numpy(flat.copy()))
h_rows = torch.empty((B, n), dtype=torch.uint8, pin_memory=True)
h_l = torch.empty(B, dtype=torch.int32, pin_memory=True); h_s = torch.empty(B, dtype=torch.int32, pin_memory=True)
C = (int(sys.argv[1]) << 20) // n
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.inflate_host(h_z, torch.from_numpy(off), n, flags=1, chunk_streams=C, h_out=h_rows, h_len=h_l, h_status=h_s, d2h=os.environ.get("D2H", "copy"))
    torch.cuda.synchronize(); print("job ms", (time.perf_counter() - t0) * 1e3, flush=True)
