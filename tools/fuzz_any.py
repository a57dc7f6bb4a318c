#!/usr/bin/env python3
"""GPU soak of the whole-GPU inflate for streams of ANY block types (hdlz_inflate_any.hip; not part of pytest: runs for --seconds):
random stock-zlib streams -- data kind, level, strategy, window, size, streams glued from segments of different kinds at full flushes
(dynamic / fixed / stored blocks in one stream), occasionally damaged or cut -- one at a time and in small batches; status, length and
bytes of every stream against the CPU oracle (and stock zlib for the good ones).  Counts how often the chain was TAKEN (by time: the
serial decoder needs ~90 ms per MiB).  Exit code 1 on the first mismatch, with the seed and the case number."""
import argparse
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, ".")
import torch                                   # noqa: E402
import hdl_deflate_amd                         # noqa: E402
from oracle import oracle as O                 # noqa: E402


def gen(rng, n):
    kind = int(rng.integers(0, 7))
    if kind == 0:
        return rng.integers(0, int(rng.choice([2, 4, 16, 64, 256])), size=n, dtype=np.uint8).tobytes()
    if kind == 1:
        per = int(rng.integers(1, 300))
        d = np.tile(rng.integers(0, 256, size=per, dtype=np.uint8), n // per + 1)[:n].copy()
        m = rng.random(n) < float(rng.choice([0.0, 0.001, 0.02]))
        d[m] = rng.integers(0, 256, size=int(m.sum()), dtype=np.uint8)
        return d.tobytes()
    if kind == 2:
        return np.minimum(255, np.abs(rng.normal(0, float(rng.choice([3, 10, 40, 120])), size=n))).astype(np.uint8).tobytes()
    if kind == 3:
        return bytes(n)
    if kind == 4:
        return np.repeat(rng.integers(0, 256, size=n // 50 + 1, dtype=np.uint8), 50)[:n].tobytes()
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(1, 10)), dtype=np.uint8)) for _ in range(int(rng.choice([20, 300, 5000])))]
    return b" ".join(words[int(i)] for i in rng.integers(0, len(words), size=n // 4 + 2))[:n].ljust(n, b".")


def stream(rng):
    nseg = int(rng.choice([1, 1, 1, 2, 3, 6]))
    raw, plain = [], []
    for k in range(nseg):
        n = int(rng.choice([200, 5000, 40000, 150000, 600000, 2000000])) if nseg > 1 else int(rng.choice([60000, 200000, 1000000, 4000000]))
        data = gen(rng, n)
        level = int(rng.choice([0, 1, 3, 6, 9]))
        strat = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FIXED][int(rng.integers(0, 6))]
        if strat == zlib.Z_FIXED and n > 20000:
            strat = zlib.Z_DEFAULT_STRATEGY
        co = zlib.compressobj(level, zlib.DEFLATED, -int(rng.choice([9, 12, 15])), int(rng.choice([8, 8, 8, 9, 1])), strat)
        body = b""
        if rng.random() < 0.15:                # a short header flushed in front of the data (zlib: a small FIXED block, then the sync marker)
            head = gen(rng, int(rng.choice([5, 40, 300, 900])))
            body = co.compress(head) + co.flush(zlib.Z_SYNC_FLUSH)
            data = head + data
        body += co.compress(data[len(data) - n:])
        if rng.random() < 0.2 and len(data) > 1000:
            body += co.flush(zlib.Z_SYNC_FLUSH) + co.compress(data[:777])
            data = data + data[:777]
        raw.append(body + (co.flush() if k == nseg - 1 else co.flush(zlib.Z_FULL_FLUSH)))
        plain.append(data)
    want = b"".join(plain)
    return b"\x78\x9c" + b"".join(raw) + zlib.adler32(want).to_bytes(4, "big"), want


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    eng = hdl_deflate_amd.Engine()
    L = eng.lib
    import ctypes
    from collections import Counter
    whys = Counter()
    dbg = hasattr(L, "hdlz_debug_par_offsets")          # (HDLZ_LIB=.../libhdlz_dbg.so: why a large good stream was handed to the serial decoder)
    if dbg:
        L.hdlz_debug_par_offsets.restype = ctypes.c_size_t
        L.hdlz_debug_par_offsets.argtypes = [ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]
    rng = np.random.default_rng(a.seed)
    t_end = time.time() + a.seconds
    it = taken = big = nbytes = damaged = 0
    while time.time() < t_end:
        it += 1
        nb = int(rng.choice([1, 1, 1, 2, 5]))
        zs, wants = [], []
        for _ in range(nb):
            z, want = stream(rng)
            q = rng.random()
            if q < 0.12:
                zb = bytearray(z)
                for _ in range(int(rng.integers(1, 4))):
                    zb[int(rng.integers(2, len(zb)))] ^= 1 << int(rng.integers(0, 8))
                z, want = bytes(zb), None
                damaged += 1
            elif q < 0.16:
                z, want = z[: int(rng.integers(6, len(z)))], None
                damaged += 1
            zs.append(z); wants.append(want)
        cap = (max(len(w) for w in wants if w is not None) if any(w is not None for w in wants) else 1 << 20) + int(rng.choice([0, 64, 4096]))
        cap = (cap + 15) // 16 * 16
        pitch = (max(len(z) for z in zs) + 64 + 15) // 16 * 16
        host = np.zeros((nb, pitch), np.uint8)
        for k, z in enumerate(zs):
            host[k, : len(z)] = np.frombuffer(z, np.uint8)
        zin = torch.from_numpy(host).cuda()
        ragged = nb > 1 and rng.random() < 0.5
        torch.cuda.synchronize()
        t0 = time.time()
        if ragged:
            flat = torch.from_numpy(np.frombuffer(b"".join(zs) + bytes(64), np.uint8).copy()).cuda()
            offs = torch.from_numpy(np.concatenate([[0], np.cumsum([len(z) for z in zs])]).astype(np.int64)).cuda()
            out, ol, st = eng.inflate_batch(flat, in_off=offs, in_len=pitch, out_pitch=cap)
        else:
            work = torch.zeros(L.hdlz_inflate_work_bytes(nb, len(zs[0]) if nb == 1 else pitch, cap, 0, 0), dtype=torch.uint8, device="cuda") if dbg and nb == 1 else None
            out, ol, st = eng.inflate_batch(zin, in_len=(len(zs[0]) if nb == 1 else None), out_pitch=cap, work=work)
        torch.cuda.synchronize()
        dt = time.time() - t0
        ho, hl, hs = out.cpu().numpy(), ol.cpu().numpy(), st.cpu().numpy()
        for k, z in enumerate(zs):
            zz = z if (ragged or nb == 1) else host[k].tobytes()
            rc, ref = O.inflate(zz, out_cap=cap)
            if hs[k] != rc or ho[k, : hl[k]].tobytes() != ref:
                print("MISMATCH seed %d case %d stream %d: status %d (oracle %d), %d bytes (oracle %d), z %d bytes, batch %d ragged %s"
                      % (a.seed, it, k, hs[k], rc, hl[k], len(ref), len(z), nb, ragged), flush=True)
                sys.exit(1)
            if wants[k] is not None and len(wants[k]) <= cap:
                assert rc == 0 and ref == wants[k], "oracle differs from zlib"
            nbytes += len(ref)
        tot = sum(len(w) for w in wants if w is not None)
        if nb == 1 and wants[0] is not None and len(zs[0]) >= 16384 and tot >= 1 << 19:
            big += 1
            fast = dt < tot / 2.0e8 + 0.004                               # (the chain: < 5 ms per MiB; one wave: ~90 ms per MiB)
            taken += 1 if fast else 0
            if dbg and not ragged:
                oc, oa = ctypes.c_size_t(0), ctypes.c_size_t(0)
                L.hdlz_debug_par_offsets(len(zs[0]), 1, cap, 0, ctypes.byref(oc), ctypes.byref(oa))
                ca = work.view(torch.int32)[oa.value // 4: oa.value // 4 + 64].cpu().tolist()
                zn_ = len(zs[0])
                over = []
                if ca[44]:
                    if ca[40] > 8 * zn_ // 128 + 1024: over.append("cand")
                    if ca[41] > min(8000, max(64, zn_ // 1024)): over.append("blocks")
                    if ca[43] > zn_ // 512 + 256: over.append("stored")
                    if not over: over.append("items/queue nx=%d" % ca[42])
                whys[(ca[0], hex(ca[45]), "+".join(over))] += 1          # fallback flag, why bits, which list overflowed
    print("fuzz_any seed %d: %d cases, %d damaged or cut streams, %.1f MB inflated, all equal to the oracle; %d of %d large good single "
          "streams at whole-GPU speed" % (a.seed, it, damaged, nbytes / 1e6, taken, big), flush=True)
    if dbg:
        print("   (fallback flag, why bits, list overflow) of the large good single streams:", dict(whys), flush=True)


if __name__ == "__main__":
    main()
