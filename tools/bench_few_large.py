import sys; sys.path.insert(0, ".")
import torch
from hdl_deflate_amd import Engine
from hdl_deflate_amd.data import make_text_blocks
e = Engine()
for nb, mib in ((4, 16), (64, 1), (256, 1), (512, 0.25), (1024, 0.25), (256, 0.0625), (1024, 0.0625), (1200, 0.0625)):
    n = int(mib * (1 << 20))
    d = make_text_blocks(nb, n, "cuda", seed=1)
    res = {}
    for name, few in (("stream-path", 1 << 30), ("one-wave-per-block", 0)):
        e.MANY_WAVES = few
        fn = lambda: e.compress_batch(d)
        o, ol, st = fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): o, ol, st = fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = (e0.elapsed_time(e1) / 3, o, ol, st)
    a, b = res["stream-path"], res["one-wave-per-block"]
    same = torch.equal(a[2], b[2]) and all(torch.equal(a[1][k, :int(a[2][k])], b[1][k, :int(b[2][k])]) for k in range(0, nb, max(1, nb // 8)))
    print("%5d x %5.2f MiB: stream path %7.3f ms, one wave per block %7.3f ms, same bytes %s" % (nb, mib, a[0], b[0], same))
