import sys; sys.path.insert(0, ".")
import torch, numpy as np, time
from hdl_deflate_amd import Engine
e = Engine()
for name, n, gen in (("zeros", 64 << 20, lambda n: torch.zeros(n + 16, dtype=torch.uint8, device="cuda")),
                     ("period3", 64 << 20, lambda n: (torch.arange(n + 16, device="cuda") % 3 + 65).to(torch.uint8)),
                     ("zeros256M", 256 << 20, lambda n: torch.zeros(n + 16, dtype=torch.uint8, device="cuda"))):
    d = gen(n); d[n:] = 0
    out, ol, st = e.compress_stream(d, n)
    zn = int(ol.item()); assert int(st.item()) == 0
    z = out[:zn + 64].contiguous().clone(); z[zn:] = 0
    zin = z[:zn].reshape(1, zn)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        back, bl, bs = e.inflate_batch(zin, out_pitch=n + 64)
        torch.cuda.synchronize(); dt = time.time() - t0
    ok = int(bs[0].item()) == 0 and int(bl[0].item()) == n and torch.equal(back[0, :n], d[:n])
    print("%-10s %d MiB -> %d bytes: inflate %.3f ms, round trip %s" % (name, n >> 20, zn, dt * 1e3, ok))
    assert ok and dt < 0.1
