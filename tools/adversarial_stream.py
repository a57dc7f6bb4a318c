import sys, os, zlib, time
sys.path.insert(0, ".")
import torch, numpy as np
from hdl_deflate_amd import Engine
from oracle import oracle as O
e = Engine()
n = 64 << 20
tests = {"zeros": torch.zeros(n + 16, dtype=torch.uint8, device="cuda"),
         "period7": (torch.arange(n + 16, device="cuda") % 7 + 65).to(torch.uint8),
         "period11": (torch.arange(n + 16, device="cuda") % 11 + 65).to(torch.uint8),
         "random": torch.randint(0, 256, (n + 16,), dtype=torch.uint8, device="cuda")}
for name, d in tests.items():
    d[n:] = 0
    out, ol, st = e.compress_stream(d, n)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out, ol, st = e.compress_stream(d, n); e1.record(); torch.cuda.synchronize()
    got = out[:int(ol.item())].cpu().numpy().tobytes()
    rc, ref = O.compress(d[:n].cpu().numpy().tobytes())
    ok = int(st.item()) == rc == 0 and got == ref and zlib.decompress(got) == d[:n].cpu().numpy().tobytes()
    print("%-9s 64 MiB: %.3f ms  %.1f GB/s  ratio %.3f  parity %s" % (name, e0.elapsed_time(e1), n / e0.elapsed_time(e1) / 1e6, len(got) / n, ok))
    assert ok
