"""CPU: the oracle against the golden vectors recorded from the executed reference, against stock
zlib, against the survey's known-answer vectors and RFC1951."""
import hashlib
import random
import zlib

import pytest

from conftest import load_golden

# known-answer vectors captured by the survey session by executing the reference;
# an independent pin: they were produced by a different stand-in kernel than oracle/standin.
SURVEY_KAT = [
    (b"aaaaa", "789c4b4c4c4c4c040005b401e6"),
    (b"abcabcabcabc", "789c4b4c4a86a2a464001de00499"),
    (bytes(64), "789c63400004400004400004606060000000400001"),
    (b"Hello World! " * 3, "789cf348cdc9c95708cf2fca4951544070101c10475101000c2f0d18"),
]
FAM1_256_OUT = ("789c535050f048cdc9c95708cf2fca49515430000a40004218286c08148000843050d8082800010861a0b03150000210c24061"
                "13a00004208481c2a640010840080385cd8002108010060a9b03052000210c14b6000a40004218286c09148000843050d8d000"
                "2802040a0a0018a03f0e")
# seeded family digests from SURVEY.md 8(c): (family, n) -> (in_sha16, out_len, out_sha16), CWINDOW=32/MATCH10
SURVEY_DIGESTS = {
    (1, 256): ("b49da343894a9334", 112, "e39fec6974d32a23"), (2, 256): ("c8a33ce888c97276", 144, "5fc88078f9fdf408"),
    (3, 256): ("deed204315156fb2", 277, "b1ee45cd6960745b"), (4, 256): ("cc13222a569db927", 123, "951a9aff3df72391"),
    (1, 2048): ("7bd3020e6ee6f060", 724, "f359ea373556bb0e"), (2, 2048): ("5a0abda82be087c8", 1062, "537857cdb9a69eaa"),
    (3, 2048): ("a84089e4ba54830d", 2167, "c8d88c30d7d07929"), (4, 2048): ("3f8a8057f43b4f8f", 867, "2773c0a6ae3c78d3"),
    (1, 16384): (None, 5451, "b12eee36ddb23dc6"), (2, 16384): (None, 8322, "edc452a9a8f0cf6f"),
    (3, 16384): (None, 17284, "8e74a52398951d35"), (4, 16384): (None, 6803, "ed9b34e632940aea"),
    (1, 65536): (None, 21131, None), (2, 65536): (None, 33322, None), (3, 65536): (None, 69124, None),
    (4, 65536): (None, 27087, None),
}


_r = random.Random(8)
DYN_TEXT = bytes(_r.choice(b"eeeeeeeeetttttttaaaaaooooiiinnn  shrdlucmfwypvbgkqjxz") for _ in range(4000))


def sha16(b):
    return hashlib.sha256(b).hexdigest()[:16]


def test_compress_golden_all_configs(oracle):
    g = load_golden("compress_vectors.json")
    assert len(g["vectors"]) >= 250
    configs = set()
    for v in g["vectors"]:
        rc, out = oracle.compress(bytes.fromhex(v["in_hex"]), v["cwindow"], v["maxmatch"])
        assert rc == oracle.OK
        assert out.hex() == v["out_hex"], (v["config"], v["name"])
        configs.add((v["cwindow"], v["maxmatch"]))
    assert {(32, 10), (32, 5), (64, 10), (256, 10), (256, 5), (16, 10), (48, 10)} <= configs


def test_compress_large_golden_multi_tile(oracle):
    """16..64 KiB vectors from the executed reference (oracle/gen_golden_large.py): multi-tile blocks at every window
    width, incl. the four 64 KiB family blocks whose lengths SURVEY.md 8(c) could only derive"""
    from conftest import large_vectors
    vs = large_vectors()
    assert len(vs) >= 24
    seen = set()
    for v, data, ref in vs:
        assert len(data) == v["n"] and sha16(data) == v["in_sha256_16"] and sha16(ref) == v["out_sha256_16"]
        rc, out = oracle.compress(data, v["cwindow"], v["maxmatch"])
        assert rc == oracle.OK and out == ref, (v["config"], v["name"])
        seen.add((v["cwindow"], v["maxmatch"], v["n"] >= 65536))
    assert {(32, 10, True), (64, 10, True), (256, 10, False), (256, 5, False), (32, 5, False), (64, 5, False)} <= seen
    lens = {v["name"]: v["out_len"] for v, _, _ in vs if v["config"] == "cw32_m10"}
    assert [lens["fam%d_65536" % f] for f in (1, 2, 3, 4)] == [21131, 33322, 69124, 27087]


def test_survey_known_answers(oracle):
    for data, hexout in SURVEY_KAT:
        rc, out = oracle.compress(data)
        assert rc == 0 and out.hex() == hexout
    from hdl_deflate_amd.data import family_bytes
    rc, out = oracle.compress(family_bytes(1, 256))
    assert out.hex() == FAM1_256_OUT


@pytest.mark.parametrize("fn", sorted(SURVEY_DIGESTS))
def test_survey_family_digests(oracle, fn):
    from hdl_deflate_amd.data import family_bytes
    f, n = fn
    in_sha, out_len, out_sha = SURVEY_DIGESTS[fn]
    data = family_bytes(f, n)
    if in_sha:
        assert sha16(data) == in_sha
    rc, out = oracle.compress(data)
    assert rc == 0 and len(out) == out_len
    if out_sha:
        assert sha16(out) == out_sha
    assert zlib.decompress(out) == data


def test_zlib_round_trip_property(oracle):
    r = random.Random(5)
    for it in range(300):
        n = r.choice([5, 6, 7, 9, 31, 32, 33, 64, 100, 255, 256, 257, 1000, 2047, 2048, 2049, 4097])
        alpha = r.choice([b"a", b"ab", b"abc", b"abcdefgh", bytes(range(256)), b"\x00\xff"])
        data = bytes(r.choice(alpha) for _ in range(n))
        cw = r.choice([1, 2, 3, 16, 31, 32, 33, 64, 100, 255, 256])
        mm = r.choice([5, 10])
        rc, out = oracle.compress(data, cw, mm)
        assert rc == 0
        assert zlib.decompress(out) == data
        assert len(out) <= oracle.out_bound(n)
        rc2, back = oracle.inflate(out)          # our own inflate restatement on our own output
        assert rc2 == 0 and back == data


def test_short_input_and_bounds(oracle):
    for n in range(5):
        rc, out = oracle.compress(bytes(n))
        assert rc == oracle.E_SHORT_INPUT and out == b""
    assert oracle.out_bound(256) == 296 and oracle.out_bound(2048) == 2312 and oracle.out_bound(65536) == 73736
    # worst case is reached by all-9-bit literals without matches
    data = bytes(144 + (i * 7) % 112 for i in range(2048))
    rc, out = oracle.compress(data, 1, 5)
    assert rc == 0 and len(out) <= 2312


def test_tables_match_rfc1951(oracle):
    L = oracle.lib()
    oc = L.hdlz_oracle_out_codes()
    sl = L.hdlz_oracle_stat_leaves()
    # literal 0 = 00110000 (8 bits) reversed = 0x0c ; EOB (256) = 0000000 ; 144 = 110010000 reversed
    assert oc[0] == 0x0c and oc[256] == 0 and oc[144] == 0x13 and oc[255] == 0x1ff and oc[287] == 0xe3
    for sym in range(286):
        nb = 8 if sym < 144 else 9 if sym < 256 else 7 if sym < 280 else 8
        assert sl[oc[sym]] == (sym << 4) | nb
    assert sl[483] == 0       # the reference's table holds 0 at the never-used symbol 287 (deflate.py:212)


def test_inflate_golden(oracle):
    g = load_golden("inflate_vectors.json")
    assert len(g["vectors"]) >= 25
    for v in g["vectors"]:
        flags = oracle.INFLATE_ASSUME_FIXED if "DYNAMIC=False" in v["build"] else 0
        obsize = 32768 if "OBSIZE=32768" in v["build"] else 512
        rc, out = oracle.inflate(bytes.fromhex(v["z_hex"]), flags=flags, obsize=obsize)
        if v["error"] is None:
            assert rc == 0, v["name"]
            assert out.hex() == v["out_hex"], v["name"]
        else:
            # the reference either raises "NO EOF!" (deflate.py:1535-1539) or stalls forever
            # (deflate.py:1600-1602, recorded as HANG); both are HDLZ_E_NO_EOF here
            assert "NO EOF" in v["error"] or v["error"].startswith("HANG")
            assert rc == oracle.E_NO_EOF and out == b"", v["name"]


def test_inflate_symbols_286_287_of_a_fixed_block(oracle):
    """tests/golden/inflate_r3_vectors.json (oracle/gen_golden_r3.py zero_leaf, executed reference): the ONE zero leaf of the
    DYNAMIC=False build (stat_leaves[483], deflate.py:212 -> "< 1 bits") against the ordinary leaves of symbols 286 / 287 -- their
    code passes NEXT and fails in INFLATE behind the end-of-input check (the reference's CopyLength tuple has 29 entries: IndexError)"""
    g = load_golden("inflate_r3_vectors.json")
    assert len(g["vectors"]) >= 12
    for v in g["vectors"]:
        flags = oracle.INFLATE_ASSUME_FIXED if "DYNAMIC=False" in v["build"] else 0
        rc, out = oracle.inflate(bytes.fromhex(v["z_hex"]), flags=flags, obsize=512)
        want = oracle.E_NO_EOF if "NO EOF" in v["error"] else oracle.E_BAD_SYMBOL
        assert "NO EOF" in v["error"] or "< 1 bits" in v["error"] or v["error"].startswith("IndexError")
        assert rc == want and out == b"", (v["name"], v["build"], rc)


def test_inflate_stock_zlib_streams(oracle):
    r = random.Random(11)
    for it in range(200):
        n = r.choice([0, 1, 5, 64, 300, 2048, 5000])
        alpha = r.choice([b"ab", b"abcdefgh", bytes(range(256)), b"0123456789 "])
        data = bytes(r.choice(alpha) for _ in range(n))
        co = zlib.compressobj(level=r.choice([0, 1, 6, 9]), strategy=zlib.Z_FIXED, wbits=15)
        z = co.compress(data[: n // 2]) + (co.flush(zlib.Z_FULL_FLUSH) if r.random() < 0.3 else b"") + \
            co.compress(data[n // 2:]) + co.flush()
        rc, out = oracle.inflate(z)
        assert rc == 0 and out == data
        rc, out = oracle.inflate(z[:-3])         # trailer must be present (deflate.py:1535-1539)
        assert rc == oracle.E_NO_EOF


def test_inflate_errors(oracle):
    # BTYPE=3
    z = b"\x78\x9c" + bytes([0x07]) + bytes(8)
    assert oracle.inflate(z)[0] == oracle.E_BAD_BTYPE
    # dynamic block: decoded; a damaged code description is BAD_TREE
    z = zlib.compress(DYN_TEXT, 9)
    assert (z[2] >> 1) & 3 == 2
    assert oracle.inflate(z) == (0, DYN_TEXT)
    bad = bytearray(z)
    bad[3] |= 0xFF; bad[4] |= 0xFF          # HLIT/HDIST/HCLEN and the first code-length codes forced to ones
    assert oracle.inflate(bytes(bad))[0] in (oracle.E_BAD_TREE, oracle.E_BAD_SYMBOL, oracle.E_BAD_DISTANCE, oracle.E_NO_EOF)
    # distance reaching before the start of the output
    bad = bytearray(zlib.compressobj(strategy=zlib.Z_FIXED).compress(b"") )
    # fixed block: literal 'a' then match len 3 dist 4 (only 1 byte produced)
    bits = []
    def put(v, n):
        for i in range(n):
            bits.append((v >> i) & 1)
    put(1, 1); put(1, 2)
    put(int('{:08b}'.format(0x30 + 97)[::-1], 2), 8)
    put(int('{:07b}'.format(1)[::-1], 2), 7)          # length symbol 257 (len 3)
    put(int('{:05b}'.format(3)[::-1], 2), 5)          # distance code 3 (dist 4)
    put(0, 7)
    while len(bits) % 8:
        bits.append(0)
    body = bytes(sum(bits[i + k] << k for k in range(8)) for i in range(0, len(bits), 8))
    z = b"\x78\x9c" + body + bytes(4)
    assert oracle.inflate(z)[0] == oracle.E_BAD_DISTANCE
    assert oracle.inflate(b"\x78\x9c\x03")[0] == oracle.E_SHORT_INPUT


def test_inflate_dynamic_streams_vs_zlib(oracle):
    r = random.Random(21)
    for it in range(150):
        n = r.choice([1, 10, 100, 1000, 5000, 40000])
        alpha = r.choice([b"ab", b"abcdefgh", bytes(range(256)), b"0123456789 ", DYN_TEXT[:64]])
        data = bytes(r.choice(alpha) for _ in range(n))
        co = zlib.compressobj(r.choice([1, 6, 9]), zlib.DEFLATED, r.choice([9, 12, 15]))
        z = co.compress(data[: n // 2]) + (co.flush(zlib.Z_FULL_FLUSH) if r.random() < 0.3 else b"") + \
            co.compress(data[n // 2:]) + co.flush()
        assert oracle.inflate(z) == (0, data)
        # random damage never crashes and never "succeeds" with a different length silently
        zb = bytearray(z)
        zb[r.randrange(2, len(zb))] ^= 1 << r.randrange(8)
        rc, out = oracle.inflate(bytes(zb))
        assert rc in range(11)


def test_batch_driver_threads(oracle):
    import numpy as np
    from hdl_deflate_amd.data import family_bytes
    blocks = [family_bytes(1 + b % 4, 200 + 13 * b, seed=b) for b in range(37)]
    flat = np.frombuffer(b"".join(blocks), np.uint8)
    off = np.cumsum([0] + [len(b) for b in blocks]).astype(np.uint64)
    out1, l1, s1 = oracle.compress_batch(flat, off, nthreads=1)
    out4, l4, s4 = oracle.compress_batch(flat, off, nthreads=4)
    assert (l1 == l4).all() and (out1 == out4).all() and (s1 == 0).all()
    for b, blk in enumerate(blocks):
        assert zlib.decompress(out1[b, :l1[b]].tobytes()) == blk


def _variant_flags(oracle, build):
    """variants_vectors.json build label -> inflate flags: DYNAMIC=False and LOWLUT builds ignore BTYPE, ONEBLOCK / LOWLUT
    builds stop at the end of the first block (deflate.py:40-49)"""
    f = 0
    if "DYNAMIC=False" in build or "LOWLUT=True" in build:
        f |= oracle.INFLATE_ASSUME_FIXED
    if "ONEBLOCK=True" in build or "LOWLUT=True" in build:
        f |= oracle.INFLATE_ONEBLOCK
    return f


def test_oneblock_and_lowlut_builds(oracle):
    """ONEBLOCK=True / LOWLUT=True reference builds (oracle/gen_golden_r2.py): the stream ends with its first block"""
    g = load_golden("variants_vectors.json")
    assert len(g["oneblock"]) >= 12
    for v in g["oneblock"]:
        rc, out = oracle.inflate(bytes.fromhex(v["z_hex"]), flags=_variant_flags(oracle, v["build"]), obsize=512)
        assert v["error"] is None and rc == 0 and out.hex() == v["out_hex"], (v["build"], v["name"])
    # without the flag the same streams decode all their blocks
    v = [x for x in g["oneblock"] if x["name"] == "fixed_two_blocks"][0]
    rc, out = oracle.inflate(bytes.fromhex(v["z_hex"]), obsize=512)
    assert rc == 0 and len(out) == 700 and out[:300].hex() == v["out_hex"]
    for v in g["lmax16"]:                      # the decoder itself has no 16-bit limit: that is the port's (test_port_protocol.py)
        rc, out = oracle.inflate(bytes.fromhex(v["z_hex"]), flags=_variant_flags(oracle, v["build"]), obsize=512, out_cap=1 << 17)
        assert rc == 0 and len(out) == v["n"]


def test_empty_distance_code_is_legal(oracle):
    """a dynamic block whose HDIST lengths are all zero holds literals only (RFC1951 3.2.7; zlib and puff accept it)"""
    from conftest import empty_distance_stream
    z = empty_distance_stream()
    assert zlib.decompress(z) == b"aaaaa"                     # stock zlib accepts it
    rc, out = oracle.inflate(z)
    assert rc == 0 and out == b"aaaaa"
