#!/usr/bin/env python3
"""Generate golden vectors by EXECUTING the unmodified reference
(/root/reference/deflate.py) under oracle/standin/myhdl.py.

TEST INFRASTRUCTURE, CONTAINER-ONLY.  /root/reference does not exist on the GPU
box and the reference source never enters this repository: it is read by path,
configuration constants are patched in memory (the reference's own "last
assignment wins" idiom, deflate.py:20-62), compiled and driven through its 10
ports with the streaming protocol of test_deflate.py:115-286 (and the preload
protocol of test_deflate.py:513-560 as a cross-check).  Only the resulting data
(inputs, parameters, outputs, cycle counts) is written to tests/golden/*.json.

Provenance label for every vector: "reference source executed under a
clocked-only stand-in kernel (oracle/standin/myhdl.py), not under MyHDL 0.10".

Usage:  python oracle/gen_golden.py [--quick] [--only NAME]
"""
import argparse
import hashlib
import json
import os
import random
import sys
import time
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("HDLZ_REFERENCE", "/root/reference")
GOLD = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, os.path.join(HERE, "standin"))
import myhdl  # noqa: E402  (the stand-in)

PROVENANCE = ("reference source executed under a clocked-only stand-in kernel "
              "(oracle/standin/myhdl.py), not under MyHDL 0.10")


# --------------------------------------------------------------------------- reference loader
def _patch(lines, lineno, expect, new):
    got = lines[lineno - 1]
    if got.strip().split("#")[0].strip() != expect:
        raise RuntimeError("reference line %d is %r, expected %r" % (lineno, got, expect))
    indent = got[:len(got) - len(got.lstrip())]
    lines[lineno - 1] = indent + new + "\n"


_cache = {}


def load_reference(cwindow=32, match10=True, fast=True, dynamic=True, obsize=512, oneblock=False, lowlut=False):
    """exec deflate.py (by path) with patched constants; returns its namespace.
    lowlut=True is the reference's LOWLUT build (deflate.py:21-22, :43-49: inflate only, DYNAMIC=False, ONEBLOCK=True,
    LMAX=16)."""
    key = (cwindow, match10, fast, dynamic, obsize, oneblock, lowlut)
    if key in _cache:
        return _cache[key]
    with open(os.path.join(REF, "deflate.py")) as f:
        lines = f.readlines()
    if lowlut:
        _patch(lines, 22, "LOWLUT = False", "LOWLUT = True")
        _patch(lines, 26, "COMPRESS = True", "COMPRESS = False")
    _patch(lines, 41, "ONEBLOCK = False", "ONEBLOCK = %s" % bool(oneblock or lowlut))
    _patch(lines, 32, "DYNAMIC = True", "DYNAMIC = %s" % bool(dynamic))
    _patch(lines, 35, "MATCH10 = True", "MATCH10 = %s" % bool(match10))
    _patch(lines, 38, "FAST = True", "FAST = %s" % bool(fast))
    _patch(lines, 57, "CWINDOW = 32", "CWINDOW = %d" % cwindow)
    _patch(lines, 59, "CWINDOW = 256", "CWINDOW = %d" % cwindow)
    _patch(lines, 62, "OBSIZE = 512", "OBSIZE = %d" % obsize)
    ns = {"__name__": "deflate", "print": lambda *a, **k: None}
    exec(compile("".join(lines), os.path.join(REF, "deflate.py"), "exec"), ns)
    assert ns["CWINDOW"] == cwindow and (lowlut or ns["MATCH10"] == bool(match10))
    assert ns["ONEBLOCK"] == bool(oneblock or lowlut) and ns["LMAX"] == (16 if lowlut else 24)
    _cache[key] = ns
    return ns


class Dut(object):
    """fresh signals + fresh reference DUT (test_deflate.py:298-319)."""

    def __init__(self, ns):
        S, intbv, modbv = myhdl.Signal, myhdl.intbv, myhdl.modbv
        L = ns["LMAX"]
        self.ns = ns
        self.i_mode = S(intbv(0)[3:])
        self.o_done = S(bool(0))
        self.i_data = S(intbv()[8:])
        self.o_byte = S(intbv()[8:])
        self.o_iprogress = S(intbv()[L:])
        self.o_oprogress = S(intbv()[L:])
        self.i_waddr = S(modbv()[L:])
        self.i_raddr = S(modbv()[L:])
        self.clk = S(bool(0))
        self.reset = myhdl.ResetSignal(0, 1, True)
        dut = ns["deflate"](self.i_mode, self.o_done, self.i_data, self.o_iprogress,
                            self.o_oprogress, self.o_byte, self.i_waddr, self.i_raddr,
                            self.clk, self.reset)
        self.design = myhdl.Design(dut)
        self.cycles = 0

    def cycle(self):
        self.design.cycle(self.clk)
        self.cycles += 1


def run_stream(ns, payload, start_cmd, short_input=False, max_cycles=None, dut=None):
    """Streaming protocol of test_deflate.py:115-195 (STARTD) / :197-286 (STARTC).

    Returns (result bytes, cycles, final o_oprogress, wait count)."""
    IDLE, WRITE, READ = ns["IDLE"], ns["WRITE"], ns["READ"]
    MAXW = ns["CWINDOW"]
    d = dut or Dut(ns)
    # CLEAR OLD INPUT
    d.i_mode.next = WRITE
    d.i_waddr.next = 0
    d.i_raddr.next = 0
    d.cycle()
    d.i_mode.next = start_cmd
    d.cycle()
    i = 0
    ri = 0
    res = bytearray()
    wait = 0
    c0 = d.cycles
    n = len(payload)
    if max_cycles is None:
        max_cycles = 200 * n + 200000
    while True:
        if ri < d.o_oprogress:
            did_read = 1
            d.i_mode.next = READ
            d.i_raddr.next = ri
            d.cycle()
            ri += 1
        else:
            did_read = 0
        if short_input and i == 0:
            # test_deflate.py:239-248  "SHORT INPUT": one WRITE of 0 at address 4
            d.i_mode.next = WRITE
            d.i_waddr.next = 4
            d.i_data.next = 0
            i = 1
        elif (not short_input) and i < n:
            if d.o_iprogress > i - MAXW:
                d.i_mode.next = WRITE
                d.i_waddr.next = i
                d.i_data.next = payload[i]
                i += 1
            else:
                wait += 1
        else:
            d.i_mode.next = IDLE
        d.cycle()
        if did_read:
            res.append(int(d.o_byte))
        if d.o_done:
            if d.o_oprogress == ri:
                break
        if d.cycles - c0 > max_cycles:
            raise RuntimeError("reference did not finish within %d cycles" % max_cycles)
    d.i_mode.next = IDLE
    d.cycle()
    return bytes(res), d.cycles - c0, int(d.o_oprogress), wait


def run_preload(ns, payload, start_cmd, max_cycles=None):
    """Preload protocol of test_deflate.py:513-560: WRITE everything, START, IDLE, then READ."""
    IDLE, WRITE, READ = ns["IDLE"], ns["WRITE"], ns["READ"]
    d = Dut(ns)
    for a, b in enumerate(payload):
        d.i_mode.next = WRITE
        d.i_waddr.next = a
        d.i_data.next = b
        d.cycle()
    d.i_mode.next = IDLE
    d.cycle()
    d.i_mode.next = start_cmd
    d.cycle()
    d.i_mode.next = IDLE
    c0 = d.cycles
    if max_cycles is None:
        max_cycles = 200 * len(payload) + 200000
    while not d.o_done:
        d.cycle()
        if d.cycles - c0 > max_cycles:
            raise RuntimeError("reference did not finish")
    total = int(d.o_oprogress)
    cyc = d.cycles - c0
    res = bytearray()
    d.i_mode.next = READ
    for a in range(total):
        d.i_raddr.next = a
        d.cycle()
        d.cycle()
        res.append(int(d.o_byte))
    return bytes(res), cyc, total


def ref_compress(data, cwindow=32, match10=True, fast=True, protocol="stream"):
    ns = load_reference(cwindow=cwindow, match10=match10, fast=fast,
                        obsize=32768 if protocol == "preload" else 512)
    if protocol == "preload":
        out, cyc, total = run_preload(ns, data, ns["STARTC"])
    else:
        out, cyc, total, _ = run_stream(ns, data, ns["STARTC"])
    assert total == len(out)
    return out, cyc


def ref_inflate(zdata, dynamic=True, obsize=512, max_cycles=None):
    ns = load_reference(dynamic=dynamic, obsize=obsize)
    try:
        out, cyc, total, _ = run_stream(ns, zdata, ns["STARTD"], max_cycles=max_cycles)
    except myhdl.Error as e:
        return None, "Error: %s" % e, 0
    except RuntimeError as e:       # the reference stalls forever (e.g. COPY hold, deflate.py:1600-1602)
        return None, "HANG: %s" % e, 0
    return out, None, cyc


# --------------------------------------------------------------------------- data families
def family(f, n, seed=1, counter0=0):
    """test_deflate.py:38-66 data families 1..4 ("modes"), seeded; truncated to n bytes."""
    r = random.Random(seed)
    if f == 0:
        s = " ".join("Hello World! " + str(1) + " " for _ in range(n // 10 + 2)).encode()
    elif f == 1:
        s = " ".join("   Hello World! " + str(counter0 + i) + "     " for i in range(n // 10 + 2)).encode()
    elif f == 2:
        s = " ".join("Hi: " + str(r.randrange(0, 0x1000)) + " " for _ in range(n // 4 + 2)).encode()
    elif f == 3:
        s = bytes(r.randrange(256) for _ in range(n))
    elif f == 4:
        s = "".join(str(r.randrange(2)) for _ in range(n)).encode()
    else:
        raise ValueError(f)
    assert len(s) >= n
    return s[:n]


def sha(b):
    return hashlib.sha256(b).hexdigest()[:16]


# --------------------------------------------------------------------------- vector sets
def compress_cases(quick):
    """(name, bytes) inputs for the compress path."""
    r = random.Random(20260928)
    cases = [
        ("aaaaa", b"aaaaa"),
        ("abcabcabcabc", b"abcabcabcabc"),
        ("zeros64", bytes(64)),
        ("hello3", b"Hello World! " * 3),
        ("n5_rand", bytes(r.randrange(256) for _ in range(5))),
        ("n6_same", b"zzzzzz"),
        ("n7_same", b"zzzzzzz"),
        ("n8_ab", b"abababab"),
        ("n9_abc", b"abcabcabc"),
        ("n12_same", b"\x00" * 12),
        ("n13_ff", b"\xff" * 13),
        ("n33_same", b"q" * 33),
        ("period1_100", b"x" * 100),
        ("period2_101", (b"xy" * 60)[:101]),
        ("period3_102", (b"xyz" * 40)[:102]),
        ("period31", (bytes(range(31)) * 5)),
        ("period32", (bytes(range(32)) * 5)),
        ("period33", (bytes(range(33)) * 5)),        # distance 33 > CWINDOW=32: no match
        ("period34_hi", (bytes(range(200, 234)) * 4)),
        ("tail_probe_a", b"abcdefgh" + b"abcdefgh"),
        ("tail_probe_b", b"0123456789abcdef" + b"0123456789abcdef"[:13]),
        ("tail_probe_c", b"0123456789" * 3 + b"01234"),
        ("hi_literals", bytes(range(140, 160)) * 2),
        ("mix_bin", bytes(r.choice([0, 0, 0, 1, 2, 255, 144, 143]) for _ in range(300))),
        ("dna", bytes(r.choice(b"ACGT") for _ in range(400))),
    ]
    for f in (0, 1, 2, 3, 4):
        cases.append(("fam%d_256" % f, family(f, 256)))
    for f in (1, 2, 3, 4):
        cases.append(("fam%d_2048" % f, family(f, 2048)))
    for k in range(12):
        n = r.randrange(5, 80)
        alpha = r.choice([b"ab", b"abc", b"abcdefgh", bytes(range(256))])
        cases.append(("rnd%02d_n%d" % (k, n), bytes(r.choice(alpha) for _ in range(n))))
    if not quick:
        for f in (1, 2, 4):
            cases.append(("fam%d_8192" % f, family(f, 8192)))
        cases.append(("fam2_2049_tilecross", family(2, 2049, seed=7)))
        cases.append(("fam4_4100_tilecross", family(4, 4100, seed=9)))
        cases.append(("fam1_2047", family(1, 2047, seed=3)))
    return cases


CONFIGS = [
    # name, kwargs, which cases (None = all)
    ("cw32_m10", dict(cwindow=32, match10=True, fast=True), None),
    ("cw32_m5", dict(cwindow=32, match10=False, fast=True), "small"),
    ("cw64_m10", dict(cwindow=64, match10=True, fast=True), "small"),
    # windows the reference does not ship: its streaming harness dead-locks when CWINDOW is
    # smaller than the 10-byte stall margin of deflate.py:768, so these use the preload protocol
    # (test_deflate.py:513-560) and inputs that fit IBSIZE = 16*CWINDOW.
    ("cw16_m10_preload", dict(cwindow=16, match10=True, fast=True, protocol="preload"), "pre240"),
    ("cw48_m10_preload", dict(cwindow=48, match10=True, fast=True, protocol="preload"), "pre240"),
    ("cw256_m10_slow", dict(cwindow=256, match10=True, fast=False), "tiny"),
    ("cw256_m5_slow", dict(cwindow=256, match10=False, fast=False), "tiny"),
]


def gen_compress(quick, only):
    cases = compress_cases(quick)
    out = {"provenance": PROVENANCE, "reference": "deflate.py STARTC path (deflate.py:616-633,734-1016)",
           "protocol": "test_deflate.py:197-286 streaming", "vectors": []}
    for cname, kw, which in CONFIGS:
        if only and only not in cname:
            continue
        for name, data in cases:
            if which == "small" and len(data) > 2100:
                continue
            if which == "tiny" and len(data) > 600:
                continue
            if which == "pre240" and len(data) > 240:
                continue
            t0 = time.time()
            res, cyc = ref_compress(data, **kw)
            assert zlib.decompress(res) == data, (cname, name)
            out["vectors"].append({
                "config": cname, "cwindow": kw["cwindow"], "maxmatch": 10 if kw["match10"] else 5,
                "fast": kw["fast"], "protocol": kw.get("protocol", "stream"), "name": name, "n": len(data),
                "in_hex": data.hex(), "out_hex": res.hex(), "out_len": len(res),
                "in_sha256_16": sha(data), "out_sha256_16": sha(res), "cycles": cyc})
            print("compress %-16s %-22s n=%5d -> %5d  %7d cyc  %.1fs" %
                  (cname, name, len(data), len(res), cyc, time.time() - t0), flush=True)
    # preload-protocol cross-check on a few inputs (must give identical bytes)
    pre = []
    for name, data in cases[:8] + [c for c in cases if c[0] == "fam1_256"]:
        a, _ = ref_compress(data, protocol="preload")
        b, _ = ref_compress(data, protocol="stream")
        assert a == b, name
        pre.append(name)
    out["preload_equals_stream_checked_on"] = pre
    return out


def zfixed(data, wbits=9):
    co = zlib.compressobj(strategy=zlib.Z_FIXED, wbits=wbits)
    return co.compress(data) + co.flush()


def inflate_cases(quick):
    r = random.Random(77)
    cs = []
    for f in (0, 1, 2, 4):
        for n in (64, 256, 2048):
            d = family(f, n)
            cs.append(("zfixed_fam%d_%d" % (f, n), zfixed(d), d))
    # multi-block fixed stream (Z_FULL_FLUSH inserts an empty stored block: BTYPE 0 with LEN=0)
    co = zlib.compressobj(strategy=zlib.Z_FIXED, wbits=9)
    d = family(1, 700)
    z = co.compress(d[:300]) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(d[300:]) + co.flush()
    cs.append(("zfixed_multiblock_fullflush", z, d))
    # stored blocks (level 0) and stored for random data
    co = zlib.compressobj(level=0, wbits=9)
    d = bytes(r.randrange(256) for _ in range(300))
    cs.append(("stored_level0_300", co.compress(d) + co.flush(), d))
    # 700-byte stored block with OBSIZE=512: the reference's LEN register is LOBSIZE=9 bits wide
    # (deflate.py:329,:714) -> it copies 700 mod 512 = 188 bytes and stops.  Recorded as observed.
    d = bytes(r.randrange(256) for _ in range(700))
    cs.append(("zfixed_random_700_stored_lenwrap", zfixed(d), d[:700 % 512]))
    # long matches / long distances within OBSIZE=512 history
    d = (b"0123456789abcdefghijklmnopqrstuvwxyz" * 40)[:1400]
    cs.append(("zfixed_long_matches", zfixed(d), d))
    d = b"A" * 1000
    cs.append(("zfixed_run_dist1", zfixed(d), d))
    d = b"AB" * 500
    cs.append(("zfixed_run_dist2", zfixed(d), d))
    return cs


def gen_inflate(quick, only):
    out = {"provenance": PROVENANCE, "reference": "deflate.py STARTD path (deflate.py:635-732,1402-1445,1519-1659)",
           "protocol": "test_deflate.py:115-195 streaming", "vectors": []}
    # (a) stock-zlib streams, default build (DYNAMIC=True parses BTYPE; only 0/1 used here), OBSIZE=512
    for name, z, d in inflate_cases(quick):
        t0 = time.time()
        res, err, cyc = ref_inflate(z, dynamic=True, obsize=512)
        assert err is None and res == d, (name, err)
        out["vectors"].append({"build": "DYNAMIC=True,OBSIZE=512", "name": name, "z_hex": z.hex(),
                               "out_hex": res.hex(), "out_len": len(res), "error": None, "cycles": cyc})
        print("inflate %-34s %5d -> %5d  %7d cyc  %.1fs" % (name, len(z), len(res), cyc, time.time() - t0), flush=True)
    # (b) cfg-4 shape: DYNAMIC=False, OBSIZE=32768, wbits=15 Z_FIXED 2 KiB blocks
    for f in (1, 2, 4):
        d = family(f, 2048)
        z = zfixed(d, wbits=15)
        res, err, cyc = ref_inflate(z, dynamic=False, obsize=32768)
        assert err is None and res == d
        out["vectors"].append({"build": "DYNAMIC=False,OBSIZE=32768", "name": "cfg4_fam%d_2048" % f,
                               "z_hex": z.hex(), "out_hex": res.hex(), "out_len": len(res),
                               "error": None, "cycles": cyc})
        print("inflate cfg4 fam%d  %5d -> %5d  %7d cyc" % (f, len(z), len(res), cyc), flush=True)
    # (c) the reference inflating its own compress output
    for name, data in [("own_fam1_256", family(1, 256)), ("own_fam4_256", family(4, 256)),
                       ("own_zeros64", bytes(64))]:
        z, _ = ref_compress(data)
        res, err, cyc = ref_inflate(z, dynamic=True, obsize=512)
        assert err is None and res == data
        out["vectors"].append({"build": "DYNAMIC=True,OBSIZE=512", "name": name, "z_hex": z.hex(),
                               "out_hex": res.hex(), "out_len": len(res), "error": None, "cycles": cyc})
    # (e) dynamic-tree streams (stock zlib default strategy), default build: SURVEY 8(f) rank 1
    dyn = []
    for f in (1, 2, 4):
        for n in (300, 2048):
            d = family(f, n, seed=11 + f)
            co = zlib.compressobj(6, zlib.DEFLATED, 9)
            dyn.append(("zdyn_fam%d_%d" % (f, n), co.compress(d) + co.flush(), d))
    r2 = random.Random(5)
    d = bytes(r2.choice(b"eeeeeeeeetttttttaaaaaooooiiinnn  shrdlucmfwypvbgkqjxz") for _ in range(3000))
    co = zlib.compressobj(9, zlib.DEFLATED, 9)
    dyn.append(("zdyn_skewed_3000", co.compress(d) + co.flush(), d))
    co = zlib.compressobj(6, zlib.DEFLATED, 9)
    z = co.compress(d[:1500]) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(family(1, 600)) + co.flush()
    dyn.append(("zdyn_multiblock_mixed", z, d[:1500] + family(1, 600)))
    for name, z, d in dyn:
        name += "_btype%d" % ((z[2] >> 1) & 3)      # zlib may still choose a fixed block for short inputs
        t0 = time.time()
        res, err, cyc = ref_inflate(z, dynamic=True, obsize=512)
        assert err is None and res == d, (name, err)
        out["vectors"].append({"build": "DYNAMIC=True,OBSIZE=512", "name": name, "z_hex": z.hex(),
                               "out_hex": res.hex(), "out_len": len(res), "error": None, "cycles": cyc})
        print("inflate %-34s %5d -> %5d  %7d cyc  %.1fs" % (name, len(z), len(res), cyc, time.time() - t0), flush=True)
    # (d) error behaviour: trailer truncated -> "NO EOF!" (deflate.py:1535-1539), or a stall
    #     that never ends (deflate.py:1600-1602), or -- one byte short -- still accepted
    for fam in (1, 4):
        z = zfixed(family(fam, 256))
        for cut in (1, 2, 3, 4, 5, 6):
            res, err, cyc = ref_inflate(z[:-cut], dynamic=True, obsize=512, max_cycles=60000)
            out["vectors"].append({"build": "DYNAMIC=True,OBSIZE=512", "name": "truncated_fam%d_%d" % (fam, cut),
                                   "z_hex": z[:-cut].hex(), "out_hex": res.hex() if res is not None else None,
                                   "out_len": len(res) if res is not None else None,
                                   "error": err, "cycles": cyc or None})
            print("inflate fam%d truncated by %d -> %r" % (fam, cut, err), flush=True)
    return out


def gen_port_modes(quick, only):
    """Per-mode flows of test_deflate.py:105-286 with SEEDED data (tlen reduced from 2500 to keep
    the fixture small): inflate leg then compress leg ON THE SAME DUT, as the reference test does."""
    out = {"provenance": PROVENANCE, "reference": "test_deflate.py:90-296 (modes 0..5)", "modes": []}
    ns = load_reference()
    tlen = 60 if quick else 120
    slen = 1000 if quick else 3000
    for mode in range(6):
        if mode == 5:
            b_data = b""
        elif mode == 3:
            r3 = random.Random(3)
            b_data = bytes(r3.randrange(256) for _ in range(tlen))
        else:
            b_data = family(mode, 10 ** 9, seed=mode) if False else _mode_data(mode, tlen)
        co = zlib.compressobj(wbits=ns["LOBSIZE"])
        zl = co.compress(b_data) + co.flush()
        # the default build is DYNAMIC=True: stock zlib may emit dynamic blocks -> only record the
        # inflate leg when the stream has no BTYPE=2 block (dynamic trees are a "next" row).
        d = Dut(ns)
        rec = {"mode": mode, "b_hex": b_data.hex(), "zl_hex": zl.hex()}
        inf, cyc_i, total_i, _ = run_stream(ns, zl, ns["STARTD"], dut=d)
        assert inf == b_data
        rec["inflate_hex"] = inf.hex()
        rec["inflate_cycles"] = cyc_i
        if len(b_data) < 4:
            comp, cyc_c, total_c, _ = run_stream(ns, b"", ns["STARTC"], short_input=True, dut=d)
            payload = b""
        else:
            payload = bytes(b_data[i % len(b_data)] for i in range(slen))
            comp, cyc_c, total_c, _ = run_stream(ns, payload, ns["STARTC"], dut=d)
        rec["compress_in_hex"] = payload.hex()
        rec["compress_hex"] = comp.hex()
        rec["compress_oprogress"] = total_c
        rec["compress_cycles"] = cyc_c
        rlen = min(len(b_data), slen)
        assert zlib.decompress(comp)[:rlen] == b_data[:rlen]
        out["modes"].append(rec)
        print("mode %d: b=%d zl=%d inflate %d cyc; compress %d -> %d, %d cyc" %
              (mode, len(b_data), len(zl), cyc_i, len(payload), len(comp), cyc_c), flush=True)
    return out


def _mode_data(m, tlen):
    r = random.Random(100 + m)
    if m == 0:
        return " ".join("Hello World! " + str(1) + " " for i in range(tlen)).encode()
    if m == 1:
        return " ".join("   Hello World! " + str(i) + "     " for i in range(tlen)).encode()
    if m == 2:
        return " ".join("Hi: " + str(r.randrange(0, 0x1000)) + " " for i in range(tlen)).encode()
    if m == 4:
        return "".join(str(r.randrange(0, 2)) for i in range(tlen)).encode()
    raise ValueError(m)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default=None)
    ap.add_argument("--sets", default="compress,inflate,port")
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    sets = a.sets.split(",")
    t0 = time.time()
    def dump(name, obj):          # generate first, then replace the fixture atomically
        tmp = os.path.join(GOLD, name + ".tmp")
        with open(tmp, "w") as f:
            json.dump(obj, f, indent=0)
        os.replace(tmp, os.path.join(GOLD, name))
    if "compress" in sets:
        dump("compress_vectors.json", gen_compress(a.quick, a.only))
    if "inflate" in sets:
        dump("inflate_vectors.json", gen_inflate(a.quick, a.only))
    if "port" in sets:
        dump("port_modes.json", gen_port_modes(a.quick, a.only))
    print("done in %.0fs" % (time.time() - t0))


if __name__ == "__main__":
    main()
