#!/usr/bin/env python3
"""Large golden vectors (multi-tile blocks, wide windows) from the EXECUTED reference.

TEST INFRASTRUCTURE, CONTAINER-ONLY -- same loader, same stand-in kernel and same provenance label as
oracle/gen_golden.py (which this script imports); the reference source is read by path and never copied.
Round 1's fixtures stop at 8 KiB (CWINDOW=32), 2 KiB (64) and 600 B (256): everything above one 2048-position
tile of the wide-window kernels was pinned only transitively (GPU = oracle, oracle = reference on small
inputs).  This set pins directly:
  * 64 KiB blocks at CWINDOW=32/MATCH10 for the four data families (SURVEY.md 8(c) lists their lengths
    21131/33322/69124/27087 as *derived only*) and for the pseudo-English text of BASELINE configs[2];
  * 16..64 KiB at CWINDOW=64, 16..32 KiB at CWINDOW=256 (FAST and non-FAST builds), MATCH10 on and off,
    sizes that are not tile multiples.
Each vector is run in its own process (the stand-in simulates 1..45 k cycles/s).  Inputs are stored
zlib-compressed + base64 ("in_b64z"), outputs base64 ("out_b64").

Usage:  python oracle/gen_golden_large.py [--jobs 8] [--only SUBSTR]
"""
import argparse
import base64
import json
import multiprocessing as mp
import os
import random
import sys
import time
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
import gen_golden as G  # noqa: E402


def text(n, seed):
    """Zipf pseudo-English, the CPU twin of hdl_deflate_amd.data.make_text_blocks' word model."""
    from hdl_deflate_amd.data import _vocab
    words = _vocab(4096)
    r = random.Random(seed)
    cum, s = [], 0.0
    for k in range(1, 4097):
        s += 1.0 / k
        cum.append(s)
    import bisect
    out = bytearray()
    while len(out) < n:
        out += (words[bisect.bisect_left(cum, r.random() * s)] + " ").encode()
    return bytes(out[:n])


def cases():
    c = []
    for f in (1, 2, 3, 4):
        c.append(("cw32_m10", dict(cwindow=32, match10=True, fast=True), "fam%d_65536" % f, G.family(f, 65536)))
    c.append(("cw32_m10", dict(cwindow=32, match10=True, fast=True), "text_65536", text(65536, 5)))
    c.append(("cw32_m10", dict(cwindow=32, match10=True, fast=True), "fam2_20011", G.family(2, 20011, seed=11)))
    for f in (2, 4):
        c.append(("cw32_m5", dict(cwindow=32, match10=False, fast=True), "fam%d_16384" % f, G.family(f, 16384)))
    for f in (1, 2, 3, 4):
        c.append(("cw64_m10", dict(cwindow=64, match10=True, fast=True), "fam%d_16384" % f, G.family(f, 16384)))
    c.append(("cw64_m10", dict(cwindow=64, match10=True, fast=True), "text_65536", text(65536, 6)))
    c.append(("cw64_m10", dict(cwindow=64, match10=True, fast=True), "fam2_65536", G.family(2, 65536, seed=2)))
    c.append(("cw64_m10", dict(cwindow=64, match10=True, fast=True), "fam4_18433", G.family(4, 18433, seed=3)))
    c.append(("cw64_m5", dict(cwindow=64, match10=False, fast=True), "text_16384", text(16384, 7)))
    for f in (1, 2, 4):
        c.append(("cw256_m10_slow", dict(cwindow=256, match10=True, fast=False), "fam%d_16384" % f, G.family(f, 16384)))
    c.append(("cw256_m10_slow", dict(cwindow=256, match10=True, fast=False), "text_32768", text(32768, 8)))
    c.append(("cw256_m10_slow", dict(cwindow=256, match10=True, fast=False), "fam2_17001", G.family(2, 17001, seed=4)))
    c.append(("cw256_m10_fast", dict(cwindow=256, match10=True, fast=True), "fam2_16384", G.family(2, 16384)))
    c.append(("cw256_m10_fast", dict(cwindow=256, match10=True, fast=True), "text_16384", text(16384, 9)))
    c.append(("cw256_m5_slow", dict(cwindow=256, match10=False, fast=False), "fam2_16384", G.family(2, 16384)))
    return c


def run_one(job):
    cname, kw, name, data = job
    t0 = time.time()
    res, cyc = G.ref_compress(data, **kw)
    assert zlib.decompress(res) == data, (cname, name)
    return {"config": cname, "cwindow": kw["cwindow"], "maxmatch": 10 if kw["match10"] else 5, "fast": kw["fast"],
            "protocol": "stream", "name": name, "n": len(data),
            "in_b64z": base64.b64encode(zlib.compress(data, 9)).decode(), "out_b64": base64.b64encode(res).decode(),
            "out_len": len(res), "in_sha256_16": G.sha(data), "out_sha256_16": G.sha(res), "cycles": cyc,
            "gen_seconds": round(time.time() - t0, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=8)
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    jobs = [j for j in cases() if not a.only or a.only in j[0] + "/" + j[2]]
    out = {"provenance": G.PROVENANCE, "reference": "deflate.py STARTC path (deflate.py:616-633,734-1016)",
           "protocol": "test_deflate.py:197-286 streaming", "encoding": "in_b64z = base64(zlib(input)), out_b64 = base64(output)",
           "vectors": []}
    t0 = time.time()
    with mp.get_context("fork").Pool(a.jobs) as pool:
        for v in pool.imap_unordered(run_one, jobs):
            out["vectors"].append(v)
            print("%-16s %-14s n=%6d -> %6d  %8d cyc  %6.1fs" % (v["config"], v["name"], v["n"], v["out_len"], v["cycles"],
                                                              v["gen_seconds"]), flush=True)
    out["vectors"].sort(key=lambda v: (v["config"], v["name"]))
    tmp = os.path.join(G.GOLD, "compress_large_vectors.json.tmp")
    with open(tmp, "w") as f:
        json.dump(out, f, indent=0)
    os.replace(tmp, os.path.join(G.GOLD, "compress_large_vectors.json"))
    print("done in %.0fs" % (time.time() - t0))


if __name__ == "__main__":
    main()
