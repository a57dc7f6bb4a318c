#!/usr/bin/env python3
"""D2H of inflated rows: hipMemcpyAsync (torch copy_) against a KERNEL that writes the rows into mapped pinned host memory
(hdlz_compact_batch with a host destination), alone and beside an H2D copy on another stream.  usage: tools/exp_d2h.py [MiB]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hdl_deflate_amd

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
eng = hdl_deflate_amd.Engine(dev)
pitch = 2048
B = (mib << 20) // pitch
d_rows = torch.randint(0, 255, (B, pitch), dtype=torch.uint8, device=dev)
lens = torch.full((B,), pitch, dtype=torch.int32, device=dev)
offs = torch.arange(B, dtype=torch.int64, device=dev) * pitch
h_out = torch.empty((B, pitch), dtype=torch.uint8, pin_memory=True)
h_in = torch.empty(mib << 18, dtype=torch.uint8, pin_memory=True)       # a quarter of the size, the other direction
d_in = torch.empty_like(h_in, device=dev)
s2 = torch.cuda.Stream(dev)
h_out.zero_(); torch.cuda.synchronize()


def timed(fn, reps=4):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts[1:])[len(ts[1:]) // 2]


def memcpy(chunk=None):
    if chunk is None:
        h_out.copy_(d_rows, non_blocking=True)
    else:
        for b0 in range(0, B, chunk):
            h_out[b0:b0 + chunk].copy_(d_rows[b0:b0 + chunk], non_blocking=True)


def kernel(chunk=None):
    c = chunk or B
    for b0 in range(0, B, c):
        nb = min(c, B - b0)
        rc = eng.lib.hdlz_compact_batch(d_rows[b0:].data_ptr(), pitch, lens[b0:].data_ptr(), offs[b0:].data_ptr(), nb, h_out.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream)
        assert rc == 0


def with_h2d(fn):
    def f():
        with torch.cuda.stream(s2):
            d_in.copy_(h_in, non_blocking=True)
        fn()
    return f


print("rows: %d MiB" % mib)
for name, fn in (("hipMemcpyAsync whole", memcpy), ("hipMemcpyAsync 32 MiB chunks", lambda: memcpy((32 << 20) // pitch)),
                 ("kernel -> mapped host, whole", kernel), ("kernel -> mapped host, 32 MiB chunks", lambda: kernel((32 << 20) // pitch))):
    h_out.zero_()
    t = timed(fn)
    ok = torch.equal(h_out, d_rows.cpu())
    t2 = timed(with_h2d(fn))
    print("%-40s %8.2f ms  %6.1f GB/s   beside an H2D of %d MiB: %8.2f ms   bytes ok: %s" % (name, t, (mib << 20) / t / 1e6, mib // 4, t2, ok))
t = timed(lambda: d_in.copy_(h_in, non_blocking=True))
print("%-40s %8.2f ms  %6.1f GB/s" % ("H2D alone (%d MiB)" % (mib // 4), t, (mib << 18) / t / 1e6))
