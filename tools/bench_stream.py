#!/usr/bin/env python3
"""time ONE large stream: multi-wave hdlz_compress_stream vs the batch kernel with a single block"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hdl_deflate_amd import Engine
from hdl_deflate_amd.data import make_text_blocks

e = Engine()
for mib in (1, 4, 16, 64, 256):
    n = mib << 20
    d = make_text_blocks(mib, 1 << 20, "cuda", seed=3).reshape(-1)
    d = torch.cat([d, torch.zeros(16, dtype=torch.uint8, device="cuda")])
    for cw in (32, 64):
        def run_stream():
            return e.compress_stream(d, n, cwindow=cw)
        def run_batch():
            return e.compress_batch(d[:n].view(1, n), cwindow=cw)
        for name, fn in (("stream", run_stream), ("batch1", run_batch)):
            if name == "batch1" and mib > 16:
                continue
            o, ol, st = fn()
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 5 if name == "stream" else 2
            ev0.record()
            for _ in range(reps):
                o, ol, st = fn()
            ev1.record()
            torch.cuda.synchronize()
            ms = ev0.elapsed_time(ev1) / reps
            print("%4d MiB cw=%-3d %-7s %9.3f ms  %8.2f GB/s  ratio %.3f st=%d" % (mib, cw, name, ms, n / ms / 1e6, int(ol.item()) / n, int(st.item())), flush=True)

# ---- the same calls replayed from a captured HIP graph (the C-ABI only enqueues async work on the caller's
# stream, so it can be captured; one eager call first: the library caches device properties on first use)
print("graph replay:")
for mib in (1, 4, 16):
    n = mib << 20
    d = make_text_blocks(mib, 1 << 20, "cuda", seed=3).reshape(-1)
    d = torch.cat([d, torch.zeros(16, dtype=torch.uint8, device="cuda")])
    out = torch.empty(((n * 9 + 10 + 7) // 8 + 6 + 15) // 16 * 16, dtype=torch.uint8, device="cuda")
    work = torch.empty((e.lib.hdlz_stream_work_bytes(n) + 7) // 8, dtype=torch.int64, device="cuda")
    ref = e.compress_stream(d, n, out=out, work=work)
    torch.cuda.synchronize()
    want = out[:int(ref[1].item())].clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            o, ol, st = e.compress_stream(d, n, out=out, work=work)
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert int(st.item()) == 0 and torch.equal(out[:int(ol.item())], want)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20):
        g.replay()
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / 20
    print("%4d MiB cw=32  graph   %9.3f ms  %8.2f GB/s" % (mib, ms, n / ms / 1e6), flush=True)
