import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hdl_deflate_amd import Engine
from hdl_deflate_amd.data import make_blocks
e = Engine()
def timed(f, reps=3):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        r = f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, r
for n, nb in ((65536, 6144), (65536, 8192), (65536, 16384), (16384, 8192), (16384, 32768), (262144, 4096), (262144, 8192)):
    total = n * nb
    gens = {"zeros": lambda: torch.zeros(total, dtype=torch.uint8, device="cuda"),
            "random": lambda: torch.randint(0, 256, (total,), dtype=torch.uint8, device="cuda"),
            "families": lambda: make_blocks(total // 2048, 2048, "cuda", seed=1).reshape(-1)}
    for name, g in gens.items():
        d = g().reshape(nb, n)
        zo, zl, st = e.compress_batch(d, cwindow=32, maxmatch=10)
        line = "%6d x %6d B %-8s |" % (nb, n, name)
        for label, fl in (("auto", 0), ("lane", 2), ("group", 64)):
            ms_i, (back, bl, bs) = timed(lambda: e.inflate_batch(zo, out_pitch=n, flags=fl))
            ok = int(bs.max().item()) == 0 and torch.equal(back[:, :n], d)
            line += " %s %8.3f ms %6.1f GB/s%s" % (label, ms_i, total / ms_i / 1e6, "" if ok else " MISMATCH")
        print(line, flush=True)
        del d, zo
        torch.cuda.empty_cache()
