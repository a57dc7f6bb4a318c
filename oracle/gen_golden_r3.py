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
Usage: python oracle/gen_golden_r3.py"""
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


if __name__ == "__main__":
    main()
