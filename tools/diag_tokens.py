import sys, zlib
sys.path.insert(0, '.')
import torch, numpy as np
import hdl_deflate_amd
from hdl_deflate_amd.data import make_blocks
from oracle import oracle as O

def rev(v, n):
    return int('{:0{w}b}'.format(v, w=n)[::-1], 2)

def parse_tokens(z):
    """fixed-Huffman single-block zlib stream -> [(pos, len, dist_or_lit)]"""
    bits = []
    for byte in z[2:]:
        for k in range(8):
            bits.append((byte >> k) & 1)
    p = 3
    toks = []
    pos = 0
    def take(n):
        nonlocal p
        v = 0
        for k in range(n):
            v |= bits[p + k] << k
        p += n
        return v
    while True:
        c7 = rev(take(7), 7)
        if c7 < 24:
            sym = 256 + c7
        else:
            c8 = (c7 << 1) | take(1)
            if 0x30 <= c8 < 0xC0: sym = c8 - 0x30
            elif 0xC0 <= c8 < 0xC8: sym = 280 + c8 - 0xC0
            else:
                c9 = (c8 << 1) | take(1)
                sym = c9 - 0x190 + 144
        if sym == 256: break
        if sym < 256:
            toks.append((pos, 0, sym)); pos += 1
        else:
            ln = sym - 254
            dc = rev(take(5), 5)
            eb = 0 if dc < 4 else (dc >> 1) - 1
            base = 1 + dc if dc < 4 else 1 + ((2 + (dc & 1)) << eb)
            d = base + take(eb)
            toks.append((pos, ln, d)); pos += ln
    return toks

eng = hdl_deflate_amd.Engine()
B, n = 65536, 2048
d = make_blocks(B, n, "cuda", seed=3)
out, ol, st = eng.compress_batch(d)
torch.cuda.synchronize()
h, ho, hl = d.cpu().numpy(), out.cpu().numpy(), ol.cpu().numpy()
for b in [191, 245, 4187, 10739]:
    blk = h[b].tobytes(); z = ho[b, :hl[b]].tobytes()
    gt = parse_tokens(z); rt = O.tokens(blk)
    k = next((i for i, (x, y) in enumerate(zip(gt, rt)) if x != y), None)
    print("block", b, "ntok gpu/ref", len(gt), len(rt), "first diff idx", k)
    if k is not None:
        print("  gpu:", gt[max(0,k-3):k+4]); print("  ref:", rt[max(0,k-3):k+4])
        p = rt[k][0]; print("  pos", p, "lane", p // 32, "i", p % 32, "bytes", blk[max(0,p-34):p+12])
