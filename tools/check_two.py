#!/usr/bin/env python3
"""two-phase inflate (HDLZ_INFLATE_TWO_PHASE) against the oracle and the one-pass kernel: stock-zlib streams of every strategy and
level over small blocks, stored blocks, multi-block streams, damaged / cut streams, capacities below the output size -- status,
length and bytes of every stream.  Usage: tools/check_two.py [rounds] [streams per round]"""
import sys, os, zlib, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hdl_deflate_amd import Engine, INFLATE_LANE_PER_STREAM, INFLATE_TWO_PHASE, INFLATE_ASSUME_FIXED
from hdl_deflate_amd.data import make_blocks
from oracle import oracle as O
e = Engine()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
r = random.Random(11)
bad = 0
for rd in range(rounds):
    pitch = r.choice((2048, 2048, 1024, 516, 64, 1536))
    n = r.choice((pitch, pitch, pitch // 2, min(2048, pitch + 40)))
    h = make_blocks(B, n, "cpu", seed=100 + rd, families=(1, 2, 3, 4)).numpy()
    zs = []
    for k in range(B):
        blk = h[k].tobytes()[: r.choice((n, n, n, r.randrange(0, n + 1)))]
        if r.random() < 0.1: blk = bytes(r.getrandbits(8) for _ in range(len(blk)))          # incompressible: literal-heavy / stored
        if r.random() < 0.05: blk = bytes([r.getrandbits(8)]) * len(blk)                     # runs: distance 1, length 258
        strat = r.choice((zlib.Z_FIXED, zlib.Z_FIXED, zlib.Z_FIXED, zlib.Z_DEFAULT_STRATEGY, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY))
        c = zlib.compressobj(r.choice((0, 1, 6, 9)), zlib.DEFLATED, r.choice((9, 12, 15)), 9, strat)
        z = c.compress(blk[: len(blk) // 2]) + (c.flush(zlib.Z_FULL_FLUSH) if r.random() < 0.3 else b"") + c.compress(blk[len(blk) // 2:]) + c.flush()
        q = r.random()
        if q < 0.08 and len(z) > 8:
            z = bytearray(z); z[r.randrange(2, len(z))] ^= 1 << r.randrange(8); z = bytes(z)
        elif q < 0.14:
            z = z[: r.randrange(0, len(z) + 1)]
        zs.append(z)
    lens = np.array([len(z) for z in zs], dtype=np.int64)
    off = np.zeros(B + 1, np.int64); np.cumsum(lens, out=off[1:])
    flat = np.frombuffer(b"".join(zs) + bytes(64), dtype=np.uint8).copy()
    zin = torch.from_numpy(flat).cuda(); zoff = torch.from_numpy(off).cuda()
    for fl in (0, INFLATE_ASSUME_FIXED):
        ref, rl, rs = O.inflate_batch(flat, off.astype(np.uint64), pitch, flags=fl, nthreads=32)
        o1, l1, s1 = e.inflate_batch(zin, in_off=zoff, out_pitch=pitch, flags=fl | INFLATE_LANE_PER_STREAM)
        o2, l2, s2 = e.inflate_batch(zin, in_off=zoff, out_pitch=pitch, flags=fl | INFLATE_LANE_PER_STREAM | INFLATE_TWO_PHASE)
        torch.cuda.synchronize()
        l2n, s2n, o2n = l2.cpu().numpy().astype(np.uint32), s2.cpu().numpy().astype(np.uint32), o2.cpu().numpy()
        ok = np.array_equal(s2n, rs) and np.array_equal(l2n, rl)
        if ok:
            m = np.arange(pitch)[None, :] < rl[:, None]
            ok = np.array_equal(o2n[m], ref[m])
        same1 = torch.equal(s1, s2) and torch.equal(l1, l2)
        print("round %d pitch %d n %d flags %d: %s (one-pass agrees: %s)  statuses %s" % (
            rd, pitch, n, fl, "OK" if ok else "MISMATCH", same1, np.bincount(rs, minlength=11).tolist()), flush=True)
        if not ok:
            bad += 1
            w = np.nonzero((s2n != rs) | (l2n != rl))[0]
            print("  first status/len diffs:", [(int(i), int(s2n[i]), int(rs[i]), int(l2n[i]), int(rl[i])) for i in w[:8]])
            if len(w) == 0:
                d = np.nonzero(((o2n != ref) & m).any(axis=1))[0]
                for i in d[:4]:
                    j = int(np.nonzero((o2n[i] != ref[i]) & m[i])[0][0])
                    print("  stream %d first byte diff at %d of %d: got %s want %s" % (i, j, rl[i], o2n[i, j:j + 8].tolist(), ref[i, j:j + 8].tolist()))
print("FAILED" if bad else "ALL OK")
sys.exit(1 if bad else 0)
