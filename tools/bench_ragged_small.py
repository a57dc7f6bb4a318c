import sys; sys.path.insert(0, ".")
import numpy as np, torch
from hdl_deflate_amd import Engine
from hdl_deflate_amd.constants import pitch_for
from hdl_deflate_amd.data import make_blocks
e = Engine()
rng = np.random.default_rng(1)
for maxlen in (128, 512, 1024):
    B = 1 << 19
    lens = rng.integers(maxlen // 2, maxlen + 1, size=B)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    total = int(off[-1])
    src = make_blocks(total // 2048 + 2, 2048, "cuda", seed=3).reshape(-1)[:total + 64].contiguous()
    d_off = torch.from_numpy(off).cuda()
    for name, kw in (("packed", dict(max_len=maxlen)), ("general", dict(out_pitch=pitch_for(maxlen)))):
        fn = lambda: e.compress_batch(src, in_off=d_off, **kw)
        o, ol, st = fn(); torch.cuda.synchronize()
        assert int((st != 0).sum()) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print("ragged %4d..%4d B x %d: %-7s %7.3f ms  %6.1f GB/s" % (maxlen // 2, maxlen, B, name, ms, total / ms / 1e6), flush=True)
