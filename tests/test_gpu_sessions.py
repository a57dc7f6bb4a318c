"""GPU (-m gpu): the resumable kernels behind the port's streaming mode (hdlz_compress_chunk / hdlz_inflate_chunk,
SURVEY.md 8(f) rank 3): a stream fed in arbitrary pieces, with arbitrary caps on the work per call, must give exactly the
bytes of the one-shot path -- which the other GPU tests pin to the oracle and the golden vectors."""
import random
import zlib

import pytest

pytestmark = pytest.mark.gpu


def _feed_compress(engine, data, r, cw, mm, piece_max, cap_choices):
    s = engine.compress_session(cwindow=cw, maxmatch=mm)
    i, seen = 0, 0
    while i < len(data):
        k = r.randint(1, piece_max)
        s.write(data[i:i + k])
        i += k
        st = s.step(max_positions=r.choice(cap_choices))
        assert st == 0 and s.out_len >= seen and not s.done
        seen = s.out_len
    while not s.done:
        st = s.step(final=True, max_positions=r.choice(cap_choices))
        assert st == 0
    return s.output(0, s.out_len)


def test_compress_session_equals_one_shot(engine, oracle):
    from hdl_deflate_amd.data import family_bytes
    r = random.Random(77)
    cases = [(family_bytes(1, 5000), 32, 10), (family_bytes(2, 9000, seed=3), 32, 10), (family_bytes(3, 3000, seed=4), 32, 5),
             (family_bytes(4, 7001, seed=5), 64, 10), (bytes(6000), 32, 10), (b"abcdefghij" * 700, 256, 10),
             (family_bytes(2, 70000, seed=9), 32, 10), (b"xyz" * 11, 32, 10), (b"12345", 32, 10)]
    for data, cw, mm in cases:
        rc, ref = oracle.compress(data, cw, mm)
        assert rc == 0
        for piece_max, caps in ((97, (None,)), (700, (32, 64, 352, None)), (5000, (None, 2048, 4096))):
            got = _feed_compress(engine, data, r, cw, mm, piece_max, caps)
            assert got == ref, (len(data), cw, mm, piece_max)
    # every complete byte reported after a step is final: a prefix of the one-shot result
    data = family_bytes(1, 20000)
    rc, ref = oracle.compress(data)
    s = engine.compress_session()
    for i in range(0, len(data), 1000):
        s.write(data[i:i + 1000])
        assert s.step() == 0
        assert s.output(0, s.out_len) == ref[:s.out_len]
    assert s.step(final=True) == 0 and s.done and s.output(0, s.out_len) == ref


def test_compress_session_short_input(engine):
    s = engine.compress_session()
    s.write(b"abcd")
    assert s.step() == 0 and s.step(final=True) == 1            # HDLZ_E_SHORT_INPUT: the reference never starts


def _feed_inflate(engine, z, r, piece_max, window, flags=0, obsize=0):
    """feed in pieces; the reader lags: the output limit grows by random amounts (the OBSIZE hold)"""
    s = engine.inflate_session(flags=flags, obsize=obsize)
    i, limit, guard = 0, window, 0
    while not s.done:
        guard += 1
        assert guard < 100000
        if i < len(z):
            k = r.randint(1, piece_max)
            s.write(z[i:i + k])
            i += k
        st = s.step(final=(i >= len(z)), out_limit=limit)
        if st != 0:
            return st, b""
        assert s.out_pos <= limit
        if s.need == 2 or r.random() < 0.3:
            limit += r.randint(1, window)                           # the reader advanced
    return 0, s.output(0, s.out_pos)


def test_inflate_session_equals_one_shot(engine, oracle):
    from hdl_deflate_amd.data import family_bytes
    r = random.Random(78)
    plain = [family_bytes(1, 6000), family_bytes(2, 5000, seed=2), family_bytes(3, 3000, seed=3), family_bytes(4, 9000, seed=4),
             bytes(40000), b"Hello World! 1 " * 3000, family_bytes(2, 200000, seed=6)]
    streams = []
    for d in plain:
        streams.append(zlib.compress(d, 6))                                        # dynamic trees
        co = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
        streams.append(co.compress(d) + co.flush())                                 # fixed
        streams.append(zlib.compress(d, 0))                                         # stored blocks
        co = zlib.compressobj(9)
        streams.append(co.compress(d[:len(d) // 3]) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(d[len(d) // 3:]) + co.flush())
    rc, own = oracle.compress(plain[0])
    streams.append(own)
    for z in streams:
        rc, ref = oracle.inflate(z)
        assert rc == 0 and ref == zlib.decompress(z)
        for piece_max, window in ((50, 512), (700, 600), (100000, 32768)):
            st, got = _feed_inflate(engine, z, r, piece_max, window)
            assert st == 0 and got == ref, (len(z), piece_max, window)
    # damaged / truncated streams: same status as the one-shot oracle, however the stream is cut into pieces
    z = zlib.compress(plain[1], 6)
    for bad in (z[:-3], z[:len(z) // 2], z[:2] + bytes([z[2] | 6]) + z[3:], z[:40] + bytes(20) + z[60:]):
        rc, _ = oracle.inflate(bad)
        st, got = _feed_inflate(engine, bad, r, 300, 4096)
        assert st == rc, (rc, st)
    # ONEBLOCK / ASSUME_FIXED builds go through the same session
    co = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
    z = co.compress(plain[0][:2000]) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(plain[0][2000:]) + co.flush()
    st, got = _feed_inflate(engine, z, r, 200, 512, flags=8)
    assert st == 0 and got == plain[0][:2000]
    st, got = _feed_inflate(engine, z[:-1], r, 200, 512, flags=1 | 8)
    assert st == 0 and got == plain[0][:2000]


def test_session_edge_cases(engine, oracle):
    """tiny streams, one-byte feeding, lengths around the look-ahead margin (11) and the lane granularity (32), empty stored
    blocks, a session reused after it finished, calls that have nothing to do"""
    r = random.Random(79)
    for n in list(range(5, 50)) + [63, 64, 65, 75, 76, 77, 2047, 2048, 2049, 2059, 2060, 4096 + 11, 4096 + 43]:
        data = bytes(r.choice(b"abcab") for _ in range(n))
        rc, ref = oracle.compress(data)
        s = engine.compress_session()
        for i in range(0, n, 7):                      # seven bytes at a time, a step after every piece
            s.write(data[i:i + 7])
            assert s.step() == 0 and s.pos % 32 == 0 and s.pos + 11 <= s.n + 0 or s.pos == 0
        assert s.step(final=True) == 0 and s.done and s.output(0, s.out_len) == ref, n
        assert s.step(final=True) == 0 and s.out_len == len(ref)          # a finished session stays as it is
        st, got = _feed_inflate(engine, ref, r, 1, 512)                   # one byte per call
        assert st == 0 and got == data, n
    s = engine.compress_session()
    assert s.step() == 0 and s.step(final=True) == 1                       # nothing written: SHORT_INPUT, no launch
    # empty stored blocks (Z_FULL_FLUSH markers) between and after data, fed in small pieces
    co = zlib.compressobj(6)
    d = bytes(r.choice(b"hello world ") for _ in range(3000))
    z = co.compress(d[:1000]) + co.flush(zlib.Z_FULL_FLUSH) + co.flush(zlib.Z_FULL_FLUSH) + co.compress(d[1000:]) + co.flush(zlib.Z_SYNC_FLUSH) + co.flush()
    assert zlib.decompress(z) == d
    for piece in (1, 3, 50, 5000):
        st, got = _feed_inflate(engine, z, r, piece, 700)
        assert st == 0 and got == d, piece
    # an output limit that never grows: the session reports need == 2 again and again without producing more
    isn = engine.inflate_session()
    isn.write(z)
    assert isn.step(final=True, out_limit=100) == 0 and isn.need == 2 and isn.out_pos <= 100
    p = isn.out_pos
    assert isn.step(final=True, out_limit=100) == 0 and isn.need == 2 and isn.out_pos == p and isn.output(0, p) == d[:p]
    assert isn.step(final=True, out_limit=1 << 20) == 0 and isn.done and isn.output(0, isn.out_pos) == d


def test_inflate_chunk_capacity_is_an_error_not_a_hold(engine):
    """hdlz_inflate_chunk with out_limit >= out_cap on a stream that needs more than out_cap bytes: raising the limit
    can never help, so the session reports HDLZ_E_OUT_CAPACITY (the batch call's code) instead of need = 2 forever; with a
    limit BELOW the capacity the same stop stays a hold"""
    import torch
    L = engine.lib
    d = bytes(random.Random(5).choice(b"hello world ") for _ in range(3000))
    z = zlib.compress(d)
    d_in = torch.frombuffer(bytearray(z + bytes(64)), dtype=torch.uint8).cuda()
    d_out = torch.zeros(1024, dtype=torch.uint8, device="cuda")
    for limit, want_status, want_need in ((4096, 2, 0), (1024, 2, 0), (512, 0, 2)):
        st = torch.zeros(16 + 80, dtype=torch.int32, device="cuda")
        rc = L.hdlz_inflate_chunk(d_in.data_ptr(), len(z), 1, 0, 0, d_out.data_ptr(), 1024, limit, st.data_ptr(),
                                  torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        s = st.cpu().tolist()
        assert (s[10], s[11]) == (want_status, want_need), (limit, s[:12])
        if want_need == 2:
            assert 0 < s[1] <= limit and bytes(d_out[:s[1]].cpu().numpy().tobytes()) == d[:s[1]]
