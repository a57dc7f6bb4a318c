import sys; sys.path.insert(0, ".")
import torch, time, numpy as np
from hdl_deflate_amd import Engine
from hdl_deflate_amd.data import make_blocks
e = Engine()
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 64) << 20
def text(n):
    rng = np.random.default_rng(5)
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 10)), dtype=np.uint8)) for _ in range(2000)]
    idx = rng.zipf(1.3, size=n // 4) % 2000
    b = b" ".join(words[i] for i in idx)[:n]
    return torch.frombuffer(bytearray(b.ljust(n + 16, b" ")), dtype=torch.uint8).cuda()
cases = {"random": lambda: torch.randint(0, 256, (n + 16,), dtype=torch.uint8, device="cuda"),
         "ramp": lambda: (torch.arange(n + 16, device="cuda") % 251).to(torch.uint8),
         "families": lambda: torch.cat([make_blocks(n // 2048, 2048, "cuda", seed=1).reshape(-1), torch.zeros(16, dtype=torch.uint8, device="cuda")]),
         "text": lambda: text(n)}
for name, g in cases.items():
    for cw in (32, 256):
        d = g(); d[n:] = 0
        out, ol, st = e.compress_stream(d, n, cwindow=cw)
        zn = int(ol.item()); assert int(st.item()) == 0
        zin = out[:zn].reshape(1, zn).contiguous()
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.time()
            back, bl, bs = e.inflate_batch(zin, out_pitch=n + 64)
            torch.cuda.synchronize(); dt = time.time() - t0
        ok = int(bs[0].item()) == 0 and int(bl[0].item()) == n and torch.equal(back[0, :n], d[:n])
        print("%-9s cw %3d: %d MiB -> %9d bytes, STARTD %.3f ms = %5.1f GB/s  round trip %s" % (name, cw, n >> 20, zn, dt * 1e3, n / dt / 1e9, ok), flush=True)
        del d, out, back, zin; torch.cuda.empty_cache()
