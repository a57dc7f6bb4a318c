import sys, zlib
sys.path.insert(0, '.')
import torch, numpy as np
import hdl_deflate_amd
from hdl_deflate_amd.data import make_blocks
from oracle import oracle as O
eng = hdl_deflate_amd.Engine()
B, n = 65536, 2048
d = make_blocks(B, n, "cuda", seed=3)
out, ol, st = eng.compress_batch(d)
back, bl, bs = eng.inflate_batch(out, out_pitch=n)
torch.cuda.synchronize()
badidx = torch.nonzero(bl != n).squeeze(1).cpu().tolist()
print("bad", len(badidx), badidx[:40])
h, ho, hl = d.cpu().numpy(), out.cpu().numpy(), ol.cpu().numpy()
hb, hbl = back.cpu().numpy(), bl.cpu().numpy()
for b in badidx[:10]:
    blk = h[b].tobytes(); z = ho[b, :hl[b]].tobytes()
    rc, ref = O.compress(blk)
    ok_c = (z == ref)
    try:
        zl_ok = zlib.decompress(z) == blk
    except Exception as e:
        zl_ok = repr(e)
    rc2, inf = O.inflate(ho[b].tobytes())
    print(b, "fam", 1 + b % 4, "clen", hl[b], "compress==oracle", ok_c, "zlib", zl_ok, "gpu_inflate_len", hbl[b],
          "oracle_inflate(padded row)", rc2, len(inf), "tail", ho[b, hl[b]-6:hl[b]+8].tobytes().hex())
# compress full check vs oracle (threaded)
flat = h.reshape(-1)
off = (np.arange(B + 1, dtype=np.uint64) * n)
ro, rl, rs = O.compress_batch(flat, off, nthreads=8, out_pitch=ho.shape[1])
neq = 0
for b in range(B):
    if hl[b] != rl[b] or not (ho[b, :hl[b]] == ro[b, :rl[b]]).all():
        neq += 1
print("compress blocks differing from oracle:", neq, "of", B)
