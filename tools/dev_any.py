#!/usr/bin/env python3
"""dev: the whole-GPU inflate of streams of any block types (hdlz_inflate_any.hip) on stock-zlib streams: result against zlib, time, and
the control words of both chains (HDLZ_LIB=hdl_deflate_amd/lib/libhdlz_dbg.so, built with -DHDLZ_DEBUG_EXPORTS)"""
import ctypes, os, random, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import hdl_deflate_amd
from hdl_deflate_amd.data import make_blocks

eng = hdl_deflate_amd.Engine()
L = eng.lib
dbg = hasattr(L, "hdlz_debug_par_offsets")
if dbg:
    L.hdlz_debug_par_offsets.restype = ctypes.c_size_t
    L.hdlz_debug_par_offsets.argtypes = [ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]

def text(n, seed):
    r = random.Random(seed)
    words = [bytes(r.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(r.randint(2, 9))) for _ in range(3000)]
    out = bytearray()
    while len(out) < n:
        out += r.choice(words) + b" "
    return bytes(out[:n])

def run(name, z, want, reps=3):
    n = len(want)
    cap = (n + 64 + 15) // 16 * 16
    zin = torch.frombuffer(bytearray(z + bytes(64)), dtype=torch.uint8).cuda().reshape(1, -1)
    wb = L.hdlz_inflate_work_bytes(1, len(z), cap, 0, 0)
    work = torch.zeros(wb, dtype=torch.uint8, device="cuda")
    out = torch.zeros((1, cap), dtype=torch.uint8, device="cuda")
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.time()
        _, ol, st = eng.inflate_batch(zin, in_len=len(z), out_pitch=cap, out=out, work=work)
        torch.cuda.synchronize(); ts.append(time.time() - t0)
    ok = int(st.item()) == 0 and int(ol.item()) == n and out[0, :n].cpu().numpy().tobytes() == want
    info = ""
    if dbg:
        oc, oa = ctypes.c_size_t(0), ctypes.c_size_t(0)
        L.hdlz_debug_par_offsets(len(z), 1, cap, 0, ctypes.byref(oc), ctypes.byref(oa))
        w32 = work.view(torch.int32)
        cf = w32[oc.value // 4: oc.value // 4 + 64].cpu().tolist()
        ca = w32[oa.value // 4: oa.value // 4 + 64].cpu().tolist()
        info = " F[fallback %d notfixed %d ok %d] ANY[fallback %d ok %d total %d nused %d mark %d | ncand %d nblk %d nx %d ns %d over %d why %#x nreq %d reqp %d dbg %s] scratch %.1f MB" % (
            cf[0], cf[7], cf[3], ca[0], ca[3], ca[2], ca[1], ca[4], ca[40], ca[41], ca[42], ca[43], ca[44], ca[45], ca[46], ca[47], ca[48:54], wb / 1e6)
    print("%-34s z %9d -> %10d  %s  %8.3f ms  %8.1f MB/s%s" % (name, len(z), n, "OK " if ok else "BAD", min(ts) * 1e3, n / min(ts) / 1e6, info), flush=True)
    return ok

which = sys.argv[1] if len(sys.argv) > 1 else "all"
allok = True
if which in ("all", "small"):
    for n in (60000, 300000, 1 << 20):
        d = text(n, n)
        allok &= run("text level 6 %d" % n, zlib.compress(d, 6), d)
    d = text(1 << 20, 5)
    allok &= run("text level 1", zlib.compress(d, 1), d)
    allok &= run("text level 9", zlib.compress(d, 9), d)
    d = np.random.default_rng(1).integers(0, 256, 1 << 20, dtype=np.uint8).tobytes()
    allok &= run("random level 6 (stored)", zlib.compress(d, 6), d)
    allok &= run("random level 0", zlib.compress(d, 0), d)
    d = make_blocks(512, 2048, "cpu", seed=3).numpy().tobytes()
    allok &= run("families level 6", zlib.compress(d, 6), d)
    # mixed: segments of different kinds joined at full flushes
    parts, plain = [], []
    for k in range(12):
        seg = text(90000 + 1000 * k, 50 + k) if k % 3 != 1 else np.random.default_rng(k).integers(0, 256, 70000, dtype=np.uint8).tobytes()
        co = zlib.compressobj([6, 0, 1, 9][k % 4], zlib.DEFLATED, -15, 8, zlib.Z_FIXED if k % 5 == 2 and False else zlib.Z_DEFAULT_STRATEGY)
        parts.append(co.compress(seg) + co.flush(zlib.Z_FULL_FLUSH)); plain.append(seg)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    tail = b"the end " * 3
    parts.append(co.compress(tail) + co.flush()); plain.append(tail)
    want = b"".join(plain)
    z = b"\x78\x9c" + b"".join(parts) + zlib.adler32(want).to_bytes(4, "big")
    assert zlib.decompress(z) == want
    allok &= run("mixed 13 segments", z, want)
if which in ("all", "big"):
    d = make_blocks(8192, 2048, "cpu", seed=5).numpy().tobytes()
    allok &= run("families 16 MiB level 6", zlib.compress(d, 6), d)
    d = text(16 << 20, 9)
    allok &= run("text 16 MiB level 6", zlib.compress(d, 6), d)
    allok &= run("text 16 MiB level 1", zlib.compress(d, 1), d)
if which in ("all", "kinds"):
    # data of other compressibility: many short codes per piece (long token lists), long matches (few tokens, many bytes), sparse data
    r = random.Random(77)
    n = 16 << 20
    def logs():
        out = bytearray(); t = 1700000000
        hosts = ["web-%02d" % i for i in range(12)]; paths = ["/api/v1/items/%d" % i for i in range(40)] + ["/index.html", "/static/app.js", "/health"]
        while len(out) < n:
            t += r.randint(0, 3)
            out += ("%d %s GET %s %d %d \"Mozilla/5.0 (X11; Linux x86_64)\" rt=%.3f\n" % (t, r.choice(hosts), r.choice(paths), r.choice([200, 200, 200, 304, 404, 500]), r.randint(100, 90000), r.random())).encode()
        return bytes(out[:n])
    def dna():
        return bytes(r.choice(b"ACGT") for _ in range(1 << 16)) * 4 + np.random.default_rng(3).choice(np.frombuffer(b"ACGT", dtype=np.uint8), n - (1 << 18)).tobytes()
    def sparse():
        a = np.zeros(n, dtype=np.uint8); idx = np.random.default_rng(4).integers(0, n, n // 64); a[idx] = np.random.default_rng(5).integers(1, 256, idx.size, dtype=np.uint8)
        return a.tobytes()
    def floats():
        return np.cumsum(np.random.default_rng(6).normal(size=n // 4)).astype(np.float32).tobytes()
    def b64():
        import base64
        return base64.b64encode(np.random.default_rng(8).integers(0, 256, n, dtype=np.uint8).tobytes())[:n]
    def hexs():
        return np.random.default_rng(9).integers(0, 256, n // 2, dtype=np.uint8).tobytes().hex().encode()[:n]
    for name, f in (("logs", logs), ("dna", dna), ("sparse", sparse), ("float32 walk", floats), ("base64", b64), ("hex", hexs)):
        if len(sys.argv) > 2 and sys.argv[2] != name:
            continue
        d = f()
        if len(sys.argv) > 2 and sys.argv[2] != name:
            continue
        for lvl in ((1, 6, 9) if len(sys.argv) <= 3 else (int(sys.argv[3]),)):
            allok &= run("%s 16 MiB level %d" % (name, lvl), zlib.compress(d, lvl), d)
if which in ("huge",):
    for mib in (64, 256):
        d = (text(16 << 20, 9) * (mib // 16))[: mib << 20] if mib <= 64 else make_blocks((mib << 20) // 2048, 2048, "cpu", seed=6).numpy().tobytes()
        allok &= run("%s %d MiB level 6" % ("text x4" if mib <= 64 else "families", mib), zlib.compress(d, 6), d, reps=2)
        if mib == 64:
            d2 = np.random.default_rng(8).integers(0, 64, mib << 20, dtype=np.uint8).tobytes()
            allok &= run("6-bit random %d MiB level 6" % mib, zlib.compress(d2, 6), d2, reps=2)
if which in ("batch",):
    # batches of stock-zlib streams in ONE call (fixed pitch): the whole-GPU chains against a wave per stream
    shapes = ((16, 1 << 20), (64, 1 << 20), (256, 1 << 20), (64, 256 << 10), (1024, 64 << 10), (16, 16 << 20))
    if len(sys.argv) > 2 and sys.argv[2] == "small":
        shapes = [(ns, n) for n in (8 << 10, 12 << 10, 16 << 10, 24 << 10, 32 << 10, 48 << 10) for ns in (1, 16, 64, 256, 512)]
    if len(sys.argv) > 2 and sys.argv[2] == "sweep":
        shapes = [(ns, n) for n in (48 << 10, 64 << 10, 256 << 10, 1 << 20) for ns in (64, 128, 256, 512, 1024, 2048, 4096) if ns * n <= (1 << 30)]
    for nstr, n in shapes:
        zs, wants = [], []
        for k in range(min(nstr, 8)):
            d = text(n, 100 + k) if k % 2 == 0 else make_blocks(max(1, n // 2048), 2048, "cpu", seed=10 + k).numpy().tobytes()
            zs.append(zlib.compress(d, 6)); wants.append(d)
        zmax = max(len(z) for z in zs)
        pitch = (zmax + 64 + 15) // 16 * 16
        cap = (n + 64 + 15) // 16 * 16
        zin = torch.zeros((nstr, pitch), dtype=torch.uint8)
        for s_ in range(nstr):
            z = zs[s_ % len(zs)]
            zin[s_, :len(z)] = torch.frombuffer(bytearray(z), dtype=torch.uint8)
        zin = zin.cuda()
        out = torch.zeros((nstr, cap), dtype=torch.uint8, device="cuda")
        res = {}
        for name, flags in (("auto", 0), ("wave", 4)):
            wb = L.hdlz_inflate_work_bytes(nstr, zmax, cap, flags, 0)
            work = torch.zeros(wb, dtype=torch.uint8, device="cuda")
            ts = []
            for _ in range(3 if name == "auto" else 1):
                torch.cuda.synchronize(); t0 = time.time()
                _, ol, st = eng.inflate_batch(zin, in_len=zmax, out_pitch=cap, out=out, flags=flags, work=work)
                torch.cuda.synchronize(); ts.append(time.time() - t0)
            ok = bool((st == 0).all()) and bool((ol == n).all())
            for s_ in sorted({0, min(1, nstr - 1), nstr - 1}):
                ok &= out[s_, :n].cpu().numpy().tobytes() == wants[s_ % len(zs)]
            allok &= ok
            res[name] = (min(ts), wb, ok)
        print("%5d x %8d bytes (z <= %8d): auto %8.3f ms %8.1f GB/s scratch %7.1f MB %s | wave per stream %9.3f ms %s" % (
            nstr, n, zmax, res["auto"][0] * 1e3, nstr * n / res["auto"][0] / 1e9, res["auto"][1] / 1e6, "OK" if res["auto"][2] else "BAD",
            res["wave"][0] * 1e3, "OK" if res["wave"][2] else "BAD"), flush=True)
print("ALL OK" if allok else "FAILURES")
