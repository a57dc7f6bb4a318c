#!/usr/bin/env python3
"""Round-2 fixtures from the EXECUTED reference (container-only; same loader, stand-in kernel and provenance label as
oracle/gen_golden.py; the reference source is read by path, never copied): tests/golden/variants_vectors.json
  * ONEBLOCK builds (deflate.py:40-41, :678, :728, :1542, :1617): DYNAMIC=True+ONEBLOCK, DYNAMIC=False+ONEBLOCK and the
    LOWLUT build (deflate.py:21-22, :43-49: inflate only, ONEBLOCK, LMAX=16) on multi-block streams;
  * LMAX=16 limit behaviour (deflate.py:73-76): 65 535 output bytes pass, the 65 536th raises
    "intbv value 65536 out of range" in the 16-bit progress counter;
  * back-pressure traces: the streaming harness with a slow reader / a slow writer
    (tests/port_harness.py), o_iprogress / o_oprogress trajectories sampled every 16 cycles, for both legs.
Usage: python oracle/gen_golden_r2.py"""
import json
import os
import sys
import time
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(REPO, "tests"))
import gen_golden as G  # noqa: E402
import port_harness as H  # noqa: E402


def signals(d):
    return dict(i_mode=d.i_mode, o_done=d.o_done, i_data=d.i_data, o_iprogress=d.o_iprogress, o_oprogress=d.o_oprogress,
                o_byte=d.o_byte, i_waddr=d.i_waddr, i_raddr=d.i_raddr)


def run(ns, payload, start, **kw):
    d = G.Dut(ns)
    try:
        res, total, trace, stats = H.stream_leg(d, signals(d), payload, start, maxw=ns["CWINDOW"], **kw)
        return res, total, trace, stats, None
    except G.myhdl.Error as e:
        return None, 0, [], {}, "Error: %s" % e
    except ValueError as e:
        return None, 0, [], {}, "ValueError: %s" % e


def zfixed(data, wbits=9):
    co = zlib.compressobj(strategy=zlib.Z_FIXED, wbits=wbits)
    return co.compress(data) + co.flush()


def two_blocks(d, cut, strategy, wbits=9, level=6):
    co = zlib.compressobj(level, zlib.DEFLATED, wbits, 8, strategy)
    return co.compress(d[:cut]) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(d[cut:]) + co.flush()


def main():
    out = {"provenance": G.PROVENANCE, "oneblock": [], "lmax16": [], "backpressure": []}
    t0 = time.time()
    d1, d2, d4 = G.family(1, 700), G.family(2, 900, seed=5), G.family(4, 800, seed=6)
    # ---- ONEBLOCK builds
    builds = [("DYNAMIC=True,ONEBLOCK=True,OBSIZE=512", dict(oneblock=True)),
              ("DYNAMIC=False,ONEBLOCK=True,OBSIZE=512", dict(oneblock=True, dynamic=False)),
              ("LOWLUT=True,OBSIZE=512", dict(lowlut=True, dynamic=False))]
    for bname, kw in builds:
        ns = G.load_reference(**kw)
        cases = [("fixed_two_blocks", two_blocks(d1, 300, zlib.Z_FIXED)), ("fixed_one_block", zfixed(d2)),
                 ("fixed_three_blocks_bfinal0_first", two_blocks(d4, 100, zlib.Z_FIXED))]
        if "DYNAMIC=True" in bname:
            cases += [("dynamic_two_blocks", two_blocks(d2, 500, zlib.Z_DEFAULT_STRATEGY)),
                      ("stored_then_fixed", zlib.compress(d1[:200], 0)[:-4 - 5] + zfixed(d1[200:])[2:]),
                      ("stored_level0", zlib.compress(d4[:300], 0))]
        for cname, z in cases:
            res, total, _, stats, err = run(ns, z, ns["STARTD"])
            out["oneblock"].append({"build": bname, "name": cname, "z_hex": z.hex(), "out_hex": res.hex() if res is not None else None,
                                    "error": err, "cycles": stats.get("cycles")})
            print("oneblock %-40s %-34s %5d -> %s %s" % (bname, cname, len(z), len(res) if res is not None else None, err or ""), flush=True)
    # ---- LMAX = 16 (LOWLUT build): the progress counters are 16 bits wide
    ns = G.load_reference(lowlut=True, dynamic=False)
    for n in (65535, 65536):
        d = (b"abcdefgh" * 8200)[:n]
        z = zfixed(d)
        res, total, _, stats, err = run(ns, z, ns["STARTD"])
        out["lmax16"].append({"build": "LOWLUT=True,OBSIZE=512", "n": n, "z_hex": z.hex(),
                              "out_sha256_16": G.sha(res) if res is not None else None, "out_len": len(res) if res is not None else None,
                              "error": err})
        print("lmax16 n=%d -> %s %s" % (n, len(res) if res is not None else None, err or ""), flush=True)
    # ---- back-pressure traces, default build (FAST, CWINDOW=32, OBSIZE=512, IBSIZE=512)
    ns = G.load_reference()
    dd = G.family(2, 3000, seed=9)
    zz = zfixed(dd)
    legs = [("inflate_slow_reader", zz, ns["STARTD"], dict(read_every=4)),
            ("inflate_slow_writer", zz, ns["STARTD"], dict(write_every=5)),
            ("inflate_eager", zz, ns["STARTD"], dict()),
            ("compress_slow_writer", dd[:2500], ns["STARTC"], dict(write_every=4)),
            ("compress_eager", dd[:2500], ns["STARTC"], dict())]
    for name, payload, start, kw in legs:
        res, total, trace, stats, err = run(ns, payload, start, trace_every=16, **kw)
        assert err is None, err
        out["backpressure"].append({"name": name, "leg": "STARTD" if start == ns["STARTD"] else "STARTC", "throttle": kw,
                                    "obsize": ns["OBSIZE"], "ibsize": ns["IBSIZE"], "cwindow": ns["CWINDOW"],
                                    "in_hex": payload.hex(), "out_hex": res.hex(), "oprogress": total, "stats": stats,
                                    "trace_columns": ["cycle", "bytes_written", "bytes_read", "o_iprogress", "o_oprogress"],
                                    "trace": trace})
        print("backpressure %-22s in %5d out %5d cycles %7d ahead<=%d lead<=%d" % (
            name, len(payload), len(res), stats["cycles"], stats["max_ahead_of_reader"], stats["max_writer_lead"]), flush=True)
    tmp = os.path.join(G.GOLD, "variants_vectors.json.tmp")
    with open(tmp, "w") as f:
        json.dump(out, f, indent=0)
    os.replace(tmp, os.path.join(G.GOLD, "variants_vectors.json"))
    print("done in %.0fs" % (time.time() - t0))


if __name__ == "__main__":
    main()
