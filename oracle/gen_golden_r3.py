#!/usr/bin/env python3
"""Round-3 fixtures from the EXECUTED reference (container-only; same loader, stand-in kernel and provenance label as
oracle/gen_golden.py; the reference source is read by path, never copied): tests/golden/streaming_r3_vectors.json
  * writer_timing   -- the STARTC leg of the streaming harness (tests/port_harness.py) on ONE input with the writer supplying a byte
                       every k-th loop iteration, k = 1, 2, 3, 4, 6, 8: the reference's compress stream depends on WHEN bytes arrive
                       (fill_buf prefetches b5..b10 beyond isize while the FSM stalls at deflate.py:768-770, and SEARCHF then compares
                       against those stale registers, deflate.py:913-952) -- which tokens differ, for every k;
  * lagging_reader  -- the same leg with a reader that takes a byte only every k-th iteration (k = 6, 12): the reference's compress
                       side has NO output hold (put / do_flush, deflate.py:535-567, write oram[do & OBS] unconditionally; only the
                       inflate side holds, deflate.py:1531-1534), so a reader that lags by more than OBSIZE reads overwritten bytes.
Usage: python oracle/gen_golden_r3.py            (streaming_r3_vectors.json)
       python oracle/gen_golden_r3.py zero_leaf  (inflate_r3_vectors.json: symbols 286 / 287 of a fixed block in both builds)"""
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
import gen_golden_r2 as R2  # noqa: E402


def main():
    out = {"provenance": G.PROVENANCE, "writer_timing": [], "lagging_reader": []}
    t0 = time.time()
    ns = G.load_reference()                      # default build: FAST, CWINDOW=32, MATCH10, OBSIZE=512, IBSIZE=512
    dd = G.family(2, 3000, seed=9)[:2500]        # the input of round 2's compress_slow_writer / compress_eager fixtures
    for k in (1, 2, 3, 4, 6, 8):
        res, total, trace, stats, err = R2.run(ns, dd, ns["STARTC"], write_every=k)
        assert err is None and zlib.decompress(res) == dd, err
        out["writer_timing"].append({"write_every": k, "in_hex": dd.hex(), "out_hex": res.hex(), "oprogress": total,
                                     "cycles": stats["cycles"], "max_writer_lead": stats["max_writer_lead"]})
        print("writer_timing  write_every %d -> %d bytes, %d cycles" % (k, len(res), stats["cycles"]), flush=True)
    for k in (6, 12):
        res, total, trace, stats, err = R2.run(ns, dd, ns["STARTC"], read_every=k, trace_every=16)
        assert err is None, err
        try:
            ok = zlib.decompress(res) == dd
        except zlib.error:
            ok = False
        out["lagging_reader"].append({"read_every": k, "obsize": ns["OBSIZE"], "ibsize": ns["IBSIZE"], "cwindow": ns["CWINDOW"],
                                      "in_hex": dd.hex(), "read_hex": res.hex(), "oprogress": total, "stats": stats,
                                      "what_the_reader_got_is_a_valid_stream": ok,
                                      "trace_columns": ["cycle", "bytes_written", "bytes_read", "o_iprogress", "o_oprogress"],
                                      "trace": trace})
        print("lagging_reader read_every %d -> read %d bytes of %d, ahead <= %d, valid stream: %s" % (
            k, len(res), total, stats["max_ahead_of_reader"], ok), flush=True)
    tmp = os.path.join(G.GOLD, "streaming_r3_vectors.json.tmp")
    with open(tmp, "w") as f:
        json.dump(out, f, indent=0)
    os.replace(tmp, os.path.join(G.GOLD, "streaming_r3_vectors.json"))
    print("done in %.0fs" % (time.time() - t0))


def _fixed_stream(symbols_bits, tail=12):
    """a zlib stream with ONE final fixed block made of the given (value, nbits, msb_first) fields, zero padding, `tail` more bytes"""
    acc, n = 0, 0

    def put(v, k, msb):
        nonlocal acc, n
        for i in range(k):
            b = (v >> (k - 1 - i)) & 1 if msb else (v >> i) & 1
            acc |= b << n
            n += 1
    put(1, 1, False); put(1, 2, False)                       # BFINAL = 1, BTYPE = 01
    for v, k, msb in symbols_bits:
        put(v, k, msb)
    body = acc.to_bytes((n + 7) // 8, "little")
    return b"\x78\x9c" + body + bytes(tail)


def zero_leaf():
    """tests/golden/inflate_r3_vectors.json: what the executed reference does with literal/length symbols 286 and 287 of a FIXED block
    in its two builds.  stat_leaves (DYNAMIC=False) holds ONE zero leaf, index 483 = symbol 287's 8-bit code followed by a 1
    (deflate.py:212, "< 1 bits" at :1437-1439); a DYNAMIC=True build decodes fixed blocks through leaves built from the fixed lengths.
    Also the 17-byte stream the round-3 damaged-stream fuzz found (symbol 287 where the input ends: the end-of-input check of
    deflate.py:1535-1539 comes first unless the leaf is the zero leaf) and its sibling with the ninth bit set."""
    out = {"provenance": G.PROVENANCE, "reference": "deflate.py STARTD path, NEXT / INFLATE (deflate.py:1402-1445, :1519-1591)", "vectors": []}
    lit_a = (0x30 + 0x61, 8, True)
    cases = [("fuzz_sym287_ninth0_at_end", bytes.fromhex("78dabbe9f8acad9ef167c05b081a17053d"))]
    z2 = bytearray(cases[0][1]); z2[14] ^= 8
    cases.append(("fuzz_sym287_ninth1_at_end", bytes(z2)))
    for sym, code in ((286, 0b11000110), (287, 0b11000111)):
        for ninth in (0, 1):
            cases.append(("lit_a_sym%d_ninth%d" % (sym, ninth), _fixed_stream([lit_a, (code, 8, True), (ninth, 1, False)])))
    for name, z in cases:
        for dyn in (False, True):
            try:
                res, err, cyc = G.ref_inflate(z, dynamic=dyn, obsize=512, max_cycles=60000)
            except Exception as e:                      # not a myhdl.Error: the reference itself crashed (e.g. CopyLength[token >= 29])
                res, err = None, "%s: %s" % (type(e).__name__, e)
            out["vectors"].append({"build": "DYNAMIC=%s,OBSIZE=512" % dyn, "name": name, "z_hex": z.hex(),
                                   "out_hex": res.hex() if res is not None else None, "error": err})
            print("%-28s DYNAMIC=%-5s -> %r" % (name, dyn, err if err else res), flush=True)
    tmp = os.path.join(G.GOLD, "inflate_r3_vectors.json.tmp")
    with open(tmp, "w") as f:
        json.dump(out, f, indent=0)
    os.replace(tmp, os.path.join(G.GOLD, "inflate_r3_vectors.json"))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "zero_leaf":
        zero_leaf()
    else:
        main()
