"""GPU (-m gpu): the HIP path, called through the C-ABI, against the oracle / golden fixtures on the
same inputs -- bit-exact -- and against stock zlib; full-size runs are checked through
size-independent properties."""
import random
import zlib

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

MAPPINGS = (2, 4, 64)   # hdlz_inflate_batch mapping hints: lane per stream, wave per stream, 16 lanes per stream


_r = random.Random(8)
DYN_TEXT = bytes(_r.choice(b"eeeeeeeeetttttttaaaaaooooiiinnn  shrdlucmfwypvbgkqjxz") for _ in range(4000))


def _ragged(torch, engine, blocks, **kw):
    """run a list of byte strings as ONE ragged batch through hdlz_compress_batch"""
    flat = b"".join(blocks) + bytes(64)
    off = np.cumsum([0] + [len(b) for b in blocks]).astype(np.int64)
    d_in = torch.frombuffer(bytearray(flat), dtype=torch.uint8).cuda()
    d_off = torch.from_numpy(off).cuda()
    out, ol, st = engine.compress_batch(d_in, in_off=d_off, **kw)
    torch.cuda.synchronize()
    out, ol, st = out.cpu().numpy(), ol.cpu().numpy(), st.cpu().numpy()
    return [bytes(out[b, :ol[b]].tobytes()) for b in range(len(blocks))], st


def test_native_library_is_loaded(engine):
    import os
    from hdl_deflate_amd import _lib
    assert os.path.exists(_lib.LIB_PATH)
    assert engine.lib.hdlz_device_count() >= 1
    with open("/proc/self/maps") as f:
        assert "libhdlz.so" in f.read()


def test_compress_golden_vectors_bit_exact(engine):
    import torch
    g = load_golden("compress_vectors.json")
    by_cfg = {}
    for v in g["vectors"]:
        by_cfg.setdefault((v["cwindow"], v["maxmatch"]), []).append(v)
    for (cw, mm), vs in sorted(by_cfg.items()):
        outs, st = _ragged(torch, engine, [bytes.fromhex(v["in_hex"]) for v in vs], cwindow=cw, maxmatch=mm)
        for v, o, s in zip(vs, outs, st):
            assert s == 0, (cw, mm, v["name"], s)
            assert o.hex() == v["out_hex"], (cw, mm, v["name"])
            assert zlib.decompress(o).hex() == v["in_hex"]


def test_compress_large_golden_vectors_bit_exact(engine):
    """multi-tile / wide-window vectors recorded from the executed reference (16..64 KiB, CWINDOW 32/64/256): through
    the one-wave-per-block kernels (ragged batch) AND through the multi-wave stream passes"""
    import torch
    from conftest import large_vectors
    by_cfg = {}
    for v, data, ref in large_vectors():
        by_cfg.setdefault((v["cwindow"], v["maxmatch"]), []).append((v, data, ref))
    for (cw, mm), vs in sorted(by_cfg.items()):
        outs, st = _ragged(torch, engine, [d for _, d, _ in vs], cwindow=cw, maxmatch=mm)
        for (v, data, ref), o, s in zip(vs, outs, st):
            assert s == 0 and o == ref, (cw, mm, v["name"])
        for v, data, ref in vs:
            s, o = engine.compress_bytes(data, cwindow=cw, maxmatch=mm)        # >= 16 KiB: hdlz_compress_stream
            assert s == 0 and o == ref, ("stream", cw, mm, v["name"])


def test_compress_random_vs_oracle(engine, oracle):
    import torch
    r = random.Random(2026)
    for cw, mm in [(32, 10), (32, 5), (64, 10), (256, 10), (1, 10), (7, 5), (33, 10), (100, 10), (255, 5)]:
        blocks = []
        for _ in range(60):
            n = r.choice([5, 6, 7, 8, 9, 10, 11, 12, 13, 31, 32, 33, 63, 64, 65, 100, 255, 256, 257, 500, 1000,
                          2047, 2048, 2049, 2050, 2057, 2058, 4095, 4096, 4097, 4106, 5000, 6143, 6144, 6145, 10000])
            alpha = r.choice([b"a", b"ab", b"abc", b"abcd", b"abcdefgh", bytes(range(256)), b"\x00\xff\x90",
                              b"0123456789 ", bytes(range(140, 150))])
            blocks.append(bytes(r.choice(alpha) for _ in range(n)))
        outs, st = _ragged(torch, engine, blocks, cwindow=cw, maxmatch=mm)
        for b, o, s in zip(blocks, outs, st):
            rc, ref = oracle.compress(b, cw, mm)
            assert s == rc == 0
            if o != ref:
                toks = oracle.tokens(b, cw, mm)
                raise AssertionError("mismatch n=%d cw=%d mm=%d first tokens %r" % (len(b), cw, mm, toks[:20]))
            assert zlib.decompress(o) == b


def test_compress_families_fixed_pitch_and_sizes(engine, oracle):
    import torch
    from hdl_deflate_amd.data import make_blocks
    for n, B in [(256, 64), (2048, 128), (65536, 8), (5, 3), (2049, 16), (70000, 3)]:
        d = make_blocks(B, n, "cuda", seed=n)
        out, ol, st = engine.compress_batch(d)
        torch.cuda.synchronize()
        h, out, ol, st = d.cpu().numpy(), out.cpu().numpy(), ol.cpu().numpy(), st.cpu().numpy()
        assert (st == 0).all()
        for b in range(B):
            blk = h[b].tobytes()
            rc, ref = oracle.compress(blk)
            assert rc == 0 and out[b, :ol[b]].tobytes() == ref, (n, b)
            assert zlib.decompress(ref) == blk


def test_compress_every_block_vs_oracle_dense_matches(engine, oracle):
    """all blocks (not a sample) against the threaded oracle on match-dense data: random 2-, 3- and
    4-symbol alphabets make long matches cross lane boundaries with entry skips 8/9, the rare case
    that a sampled check misses (caught a sign-extension bug in the parse chain in round 1)."""
    import torch
    B, n = 16384, 2048
    g = torch.Generator(device="cuda")
    g.manual_seed(42)
    for nsym, cw, mm in [(2, 32, 10), (3, 32, 10), (4, 64, 10), (2, 32, 5), (2, 256, 10)]:
        d = torch.randint(0, nsym, (B, n), generator=g, device="cuda", dtype=torch.uint8) + 48
        out, ol, st = engine.compress_batch(d, cwindow=cw, maxmatch=mm)
        torch.cuda.synchronize()
        h, ho, hl = d.cpu().numpy(), out.cpu().numpy(), ol.cpu().numpy()
        off = np.arange(B + 1, dtype=np.uint64) * n
        ro, rl, rs = oracle.compress_batch(h.reshape(-1), off, cw, mm, out_pitch=ho.shape[1], nthreads=8)
        assert (st.cpu().numpy() == 0).all() and (rs == 0).all()
        assert (hl == rl).all(), (nsym, cw, mm, int((hl != rl).sum()))
        mask = np.arange(ho.shape[1])[None, :] < hl[:, None]
        assert ((ho == ro) | ~mask).all(), (nsym, cw, mm)


def test_compress_one_tile_block_starts(engine, oracle):
    """round 4: in the one-tile kernels the candidates in front of a lane's run are the previous lane's own keys (a DPP rotate), and lane 0
    -- the first 32 bytes of the block -- takes lane 63's: whatever they match must lie in front of the block and be rejected.  Blocks
    whose first bytes are 0xFF / 0x00 runs, whose END repeats their START (what the rotate feeds lane 0), one-symbol alphabets, lengths
    around the tile size: every block against the oracle, uniform batch (k_compress<1, ., true>) and ragged."""
    import torch
    r = random.Random(41)
    blocks = []
    for n in (2048, 2047, 2016, 1990, 1025, 64, 37, 5):
        blocks.append(bytes([0xFF]) * n)
        blocks.append(bytes(n))
        blocks.append((bytes([0, 0, 0]) + bytes([0xFF]) * 40 + bytes(r.getrandbits(8) for _ in range(n)))[:n])
        head = bytes(r.getrandbits(8) for _ in range(40))
        body = bytes(r.choice(b"ab") for _ in range(n))
        blocks.append((head + body)[: max(n - 40, 0)] + head[: n - max(n - 40, 0)])      # the block ends with its own first bytes
        blocks.append((bytes([0xFF, 0xFF, 0xFF]) * 12 + bytes(r.choice(b"xyz") for _ in range(n)))[: max(n - 36, 0)] + (bytes([0xFF]) * 36)[: n - max(n - 36, 0)])
        blocks.append(bytes(r.choice(b"\xff\x00") for _ in range(n)))
    for cw in (32, 17):
        got, st = _ragged(torch, engine, blocks, cwindow=cw, maxmatch=10)
        for k, b in enumerate(blocks):
            rc, ref = oracle.compress(b, cwindow=cw, maxmatch=10)
            assert st[k] == rc and got[k] == ref, (cw, k, len(b))
    # the uniform batch: 2048-byte blocks through the one-tile kernel with a fixed pitch
    full = [b for b in blocks if len(b) == 2048]
    d = torch.frombuffer(bytearray(b"".join(full)), dtype=torch.uint8).cuda().reshape(len(full), 2048)
    out, ol, st = engine.compress_batch(d, cwindow=32, maxmatch=10)
    torch.cuda.synchronize()
    out, ol = out.cpu().numpy(), ol.cpu().numpy()
    for k, b in enumerate(full):
        rc, ref = oracle.compress(b, cwindow=32, maxmatch=10)
        assert rc == 0 and bytes(out[k, :ol[k]].tobytes()) == ref, k


def test_compress_wide_windows_large_blocks_every_block(engine, oracle):
    """CWINDOW=64 and 256 (k_compress<2>, <8>) on 64 KiB multi-tile blocks of pseudo-English and of the
    families, MATCH10 on/off: every block against the threaded oracle"""
    import torch
    from hdl_deflate_amd.data import make_blocks, make_text_blocks
    B, n = 192, 65536
    for mk, cw, mm in [(make_text_blocks, 64, 10), (make_text_blocks, 256, 10), (make_blocks, 64, 5), (make_blocks, 256, 5),
                       (make_text_blocks, 33, 10), (make_blocks, 200, 10)]:
        d = mk(B, n, "cuda", seed=cw + mm)
        out, ol, st = engine.compress_batch(d, cwindow=cw, maxmatch=mm)
        torch.cuda.synchronize()
        h, ho, hl = d.cpu().numpy(), out.cpu().numpy(), ol.cpu().numpy()
        off = np.arange(B + 1, dtype=np.uint64) * n
        ro, rl, rs = oracle.compress_batch(h.reshape(-1), off, cw, mm, out_pitch=ho.shape[1], nthreads=8)
        assert (st.cpu().numpy() == 0).all() and (rs == 0).all()
        assert (hl == rl).all(), (cw, mm)
        mask = np.arange(ho.shape[1])[None, :] < hl[:, None]
        assert ((ho == ro) | ~mask).all(), (cw, mm)
        assert zlib.decompress(ho[0, :hl[0]].tobytes()) == h[0].tobytes()


def test_compress_hashed_finder_adversarial_inputs(engine, oracle):
    """the window-independent match finder (CWINDOW > 64, hdlz_compress_common.h: match_search_hash) on the inputs that stress its
    tables: all positions in ONE hash class (zeros, short periods: 64 lanes on one table address and one leader slot), periods
    around the round length 64 and the window, two-symbol noise (eight keys), random bytes (no class twice), keys that differ in
    one byte only; windows that are not multiples of 32 (the per-position window compare) and the full 256; sizes around tile
    multiples -- every block against the oracle, as one ragged batch per window"""
    import torch
    r = random.Random(41)
    pats = []
    for n in (2049, 4096 + 17, 6000, 65536):
        pats += [bytes(n), bytes([65 + (i % 2) for i in range(n)]), bytes([65 + (i % 7) for i in range(n)]),
                 bytes([(i % 63) for i in range(n)]), bytes([(i % 64) for i in range(n)]), bytes([(i % 65) for i in range(n)]),
                 bytes([(i % 255) for i in range(n)]), bytes([(i % 256) for i in range(n)]), bytes([((i % 257) * 7) & 255 for i in range(n)]),
                 bytes(r.choice(b"01") for _ in range(n)), bytes(r.randrange(256) for _ in range(n)),
                 bytes((97 if (i // 3) % 2 else 98) if i % 3 else r.randrange(4) for i in range(n)),
                 (b"abcdefghij" * 30 + bytes(r.randrange(256) for _ in range(40))) * (n // 340 + 1)]
    pats = [p[:n] for p in pats for n in (len(p),)]
    for cw in (65, 100, 128, 200, 255, 256):
        for mm in ((10,) if cw != 256 else (10, 5)):
            got, st = _ragged(torch, engine, pats, cwindow=cw, maxmatch=mm)
            for k, (p_, g_) in enumerate(zip(pats, got)):
                rc, ref = oracle.compress(p_, cw, mm)
                assert st[k] == rc == 0 and g_ == ref, (cw, mm, k, len(p_))


def test_compress_stream_multiwave_vs_oracle(engine, oracle):
    """hdlz_compress_stream: ONE stream spread over the GPU (three passes) must give the oracle's bytes --
    sizes around tile multiples (2048), match-dense alphabets (matches straddle every tile boundary, all entry
    skips), families, text, all kernel variants, a misaligned source pointer, and the 16 MiB LMAX-sized stream"""
    import torch
    from hdl_deflate_amd.data import make_blocks, make_text_blocks
    g = torch.Generator(device="cuda")
    g.manual_seed(99)

    def check(d, n, cw, mm, tag):
        out, ol, st = engine.compress_stream(d, n, cwindow=cw, maxmatch=mm)
        torch.cuda.synchronize()
        assert int(st.item()) == 0, tag
        got = out[:int(ol.item())].cpu().numpy().tobytes()
        st_o, ref = oracle.compress(d[:n].cpu().numpy().tobytes(), cw, mm)
        assert st_o == 0 and got == ref, (tag, len(got), len(ref))
        return got

    big = 1 << 20
    dense = {k: torch.randint(0, k, (big + 64,), generator=g, device="cuda", dtype=torch.uint8) + 48 for k in (2, 3, 4)}
    for n in (65536, 65537, 65541, 2048 * 37 - 1, 2048 * 37 + 4, 2048 * 37 + 5, 2048 * 37 + 6, 100000, 2048 * 64 + 2047,
              big - 3, big):
        for nsym, cw, mm in ((2, 32, 10), (3, 32, 5), (4, 64, 10), (2, 256, 10)):
            if cw == 256 and n > 200000:
                continue
            check(dense[nsym], n, cw, mm, ("dense", nsym, n, cw, mm))
    fam = make_blocks(16, 65536, "cuda", seed=5).reshape(-1)          # 1 MiB of all families back to back
    txt = make_text_blocks(4, 1 << 20, "cuda", seed=6).reshape(-1)    # 4 MiB pseudo-English
    for d, n, cw, mm in ((fam, fam.numel() - 16, 32, 10), (fam, 300001, 64, 10), (txt, txt.numel() - 16, 32, 10),
                         (txt, 1 << 20, 64, 5), (txt, 262144 + 77, 256, 10)):
        check(d, n, cw, mm, ("mix", n, cw, mm))
    # source pointer misaligned by 1..3 and by 8 (the dword re-align path of the staging loop)
    for k in (1, 2, 3, 8):
        check(txt[k:], 200000 + k, 32, 10, ("misaligned", k))
    # LMAX-sized stream (deflate.py:73-76: 16 MiB), also through zlib
    d = make_text_blocks(16, 1 << 20, "cuda", seed=7).reshape(-1)
    d = torch.cat([d, torch.zeros(16, dtype=torch.uint8, device="cuda")])
    z = check(d, 1 << 24, 32, 10, "lmax")
    assert zlib.decompress(z) == d[:1 << 24].cpu().numpy().tobytes()
    # batches of large blocks through compress_batch = hdlz_compress_streams (all tiles of all blocks share the passes):
    # same bytes as one wave per block, incl. blocks that are not a multiple of the tile size and match-saturated blocks
    for nbk, nbytes, src_t in ((5, 1 << 19, d), (37, (1 << 18) + 1234, d), (3, 300000, dense[2])):
        few = torch.zeros((nbk, (nbytes + 15) // 16 * 16), dtype=torch.uint8, device="cuda")
        few[:, :nbytes] = src_t[:nbk * nbytes].view(nbk, nbytes)
        fo, fl, fs = engine.compress_batch(few, in_len=nbytes)
        saved = engine.MANY_WAVES
        engine.MANY_WAVES = 0                          # never: one wave per block
        bo, bl_, bs_ = engine.compress_batch(few, in_len=nbytes)
        engine.MANY_WAVES = saved
        torch.cuda.synchronize()
        assert int((fs != 0).sum()) == 0 and int((bs_ != 0).sum()) == 0 and torch.equal(fl, bl_), (nbk, nbytes)
        for b in range(nbk):
            assert torch.equal(fo[b, :int(fl[b])], bo[b, :int(bl_[b])]), (nbk, nbytes, b)
        assert fo[0, :int(fl[0])].cpu().numpy().tobytes() == oracle.compress(few[0, :nbytes].cpu().numpy().tobytes())[1]
    # engine.compress_bytes routes large streams here, small ones through the batch kernel: same answer
    blob = d[:70000].cpu().numpy().tobytes()
    st, zz = engine.compress_bytes(blob)
    assert st == 0 and zz == oracle.compress(blob)[1]
    # status codes
    out, ol, st = engine.compress_stream(d, 4)
    assert int(st.item()) == 1 and int(ol.item()) == 0
    small = torch.empty(1024, dtype=torch.uint8, device="cuda")
    out, ol, st = engine.compress_stream(d, 70000, out=small)
    assert int(st.item()) == 2 and int(ol.item()) == 0


def test_calls_are_hip_graph_capturable(engine, oracle):
    """the C-ABI only enqueues asynchronous work on the caller's stream and -- round 6 -- allocates nothing: every buffer, scratch
    included, is the caller's (hdlz_inflate_batch_ws / hdlz_archive_batch_ws).  Batch compress, stream compress, the whole-GPU inflate
    in its SEVERAL-STREAMS form (512 streams of >= HDLZ_INFLATE_PAR_MIN bytes: the form whose pool scratch, as graph memory nodes, aborted
    in hipGraphLaunch in round 5 -- profiles/r06_graph_abort_cause.txt), the lane mapping with its second-pass lists, the archive
    and the 16-lane mapping captured into ONE HIP graph, launched FIFTY times on new data in the same buffers"""
    import torch
    from hdl_deflate_amd.data import make_blocks
    from hdl_deflate_amd.constants import pitch_for
    B, n = 512, 2048
    d = make_blocks(B, n, "cuda", seed=1)
    big = make_blocks(40, n, "cuda", seed=2).reshape(-1)              # one 80 KiB stream
    nbig = big.numel() - 16
    out = torch.empty((B, pitch_for(n)), dtype=torch.uint8, device="cuda")
    sout = torch.empty(pitch_for(nbig), dtype=torch.uint8, device="cuda")
    work = torch.empty((engine.lib.hdlz_stream_work_bytes(nbig) + 7) // 8, dtype=torch.int64, device="cuda")
    back = torch.empty((B, n), dtype=torch.uint8, device="cuda")
    back2 = torch.empty((B, n), dtype=torch.uint8, device="cuda")
    back3 = torch.empty((B, n), dtype=torch.uint8, device="cuda")
    arch = torch.empty(B * out.shape[1] + 64, dtype=torch.uint8, device="cuda")
    aoff = torch.empty(B + 1, dtype=torch.int64, device="cuda")
    L = engine.lib
    iw = torch.empty(L.hdlz_inflate_work_bytes(B, out.shape[1], n, 0, 0), dtype=torch.uint8, device="cuda")     # one scratch buffer for
    assert iw.numel() >= L.hdlz_inflate_work_bytes(B, out.shape[1], n, 2, 0) and iw.numel() >= L.hdlz_inflate_work_bytes(B, 0, n, 64, 1)   # the three inflate calls
    aw = torch.empty(L.hdlz_archive_work_bytes(B), dtype=torch.uint8, device="cuda")
    assert iw.numel() > 4 * B * n                                       # (the whole-GPU path's markers: this call takes that path)
    engine.compress_batch(d, out=out, out_pitch=out.shape[1])          # eager once: device properties get cached
    engine.compress_stream(big, nbig, out=sout, work=work)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            _, ol, st = engine.compress_batch(d, out=out, out_pitch=out.shape[1])
            _, sl, ss = engine.compress_stream(big, nbig, out=sout, work=work)
            _, bl, bs = engine.inflate_batch(out, in_len=None, out_pitch=n, out=back, work=iw)              # k_par_*, blockIdx.y = the stream
            # the lane mapping: pass 1, then the device-side list of streams with dynamic blocks
            _, bl2, bs2 = engine.inflate_batch(out, in_len=None, out_pitch=n, out=back2, flags=2, work=iw)
            # scan + gather in one launch (a ticket and descriptors in scratch), then the 16-lanes-per-stream mapping on the ragged archive
            engine.archive(out, ol, archive=arch, offsets=aoff, work=aw)
            _, bl3, bs3 = engine.inflate_batch(arch, in_off=aoff, out_pitch=n, out=back3, flags=64, work=iw)
            # the entry points that allocate from the library's pool refuse a capturing stream (and leave the capture intact)
            q = bl3.data_ptr()
            assert L.hdlz_inflate_batch(out.data_ptr(), None, out.shape[1], out.shape[1], B, 0, 0, back.data_ptr(), n, q, q, s.cuda_stream) == 8
            assert b"hdlz_inflate_batch_ws" in L.hdlz_last_error()
            assert L.hdlz_archive_batch(out.data_ptr(), out.shape[1], ol.data_ptr(), B, arch.data_ptr(), arch.numel(), aoff.data_ptr(), s.cuda_stream) == 8
    for seed in range(11, 61):
        d.copy_(make_blocks(B, n, "cuda", seed=seed))
        big.copy_(make_blocks(40, n, "cuda", seed=seed + 100).reshape(-1))
        out.zero_(); sout.zero_(); back.zero_(); back2.zero_(); back3.zero_(); aoff.zero_(); iw.fill_(0xA5); aw.fill_(0x5A)
        g.replay()
        torch.cuda.synchronize()
        assert int((st != 0).sum()) == 0 and int(ss.item()) == 0, seed
        if seed < 13:
            h, ho, hl = d.cpu().numpy(), out.cpu().numpy(), ol.cpu().numpy()
            for b in (0, 1, B // 2, B - 1):
                assert ho[b, :hl[b]].tobytes() == oracle.compress(h[b].tobytes())[1]
            assert sout[:int(sl.item())].cpu().numpy().tobytes() == oracle.compress(big[:nbig].cpu().numpy().tobytes())[1]
        # inflate took the full pitch as in_len: trailing zero bytes after the Adler-32 are ignored (D6)
        assert int((bs != 0).sum()) == 0 and torch.equal(back, d) and int((bl != n).sum()) == 0, seed
        assert int((bs2 != 0).sum()) == 0 and torch.equal(back2, d) and int((bl2 != n).sum()) == 0, seed
        assert int(aoff[-1].item()) == int(ol.to(torch.int64).sum().item()), seed
        assert int((bs3 != 0).sum()) == 0 and torch.equal(back3, d) and int((bl3 != n).sum()) == 0, seed


def test_inflate_with_less_scratch_than_asked_for(engine, oracle):
    """hdlz_inflate_batch_ws with LESS scratch than hdlz_inflate_work_bytes (down to none): the same results through mappings that need
    less -- the whole-GPU path in groups of streams that fit, or skipped; the lane mapping in index order; the second pass a wave per
    stream.  Shapes: a few large fixed-block streams (k_par_*), a ragged batch with dynamic-tree streams (lane mapping + second pass)"""
    import zlib
    import numpy as np
    import torch
    from hdl_deflate_amd.data import make_blocks
    L = engine.lib
    # (a) 48 streams of 64 KiB, fixed pitch: the whole-GPU path
    d = make_blocks(48 * 32, 2048, "cuda", seed=21).reshape(48, 65536)
    z, zl, st = engine.compress_batch(d)
    full = L.hdlz_inflate_work_bytes(48, z.shape[1], 65536, 0, 0)
    assert full > 48 * 65536 * 2
    for wb in (full, full // 2, full // 5, full // 48 + 4096, 70000, 256, 0):
        w = torch.full((wb,), 0xEE, dtype=torch.uint8, device="cuda")
        back, bl, bs = engine.inflate_batch(z, out_pitch=65536, work=w)
        assert int((bs != 0).sum()) == 0 and int((bl != 65536).sum()) == 0 and torch.equal(back, d), wb
    # (b) 3000 ragged streams, every third one with dynamic trees, the others Z_FIXED
    blocks = make_blocks(3000, 2048, "cuda", seed=22, families=(1, 2, 4)).cpu().numpy()
    zs = []
    for k in range(3000):
        co = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_DEFAULT_STRATEGY if k % 3 == 0 else zlib.Z_FIXED)
        zs.append(co.compress(blocks[k].tobytes()) + co.flush())
    off = np.zeros(3001, np.int64)
    off[1:] = np.cumsum([len(x) for x in zs])
    flat = torch.from_numpy(np.frombuffer(b"".join(zs) + bytes(64), dtype=np.uint8).copy()).cuda()
    d_off = torch.from_numpy(off).cuda()
    want = torch.from_numpy(blocks).cuda()
    for flags in (2, 64, 0):
        full = L.hdlz_inflate_work_bytes(3000, 0, 2048, flags, 1)
        for wb in (full, full // 2, 1024, 0):
            w = torch.full((wb,), 0xEE, dtype=torch.uint8, device="cuda")
            back, bl, bs = engine.inflate_batch(flat, in_off=d_off, out_pitch=2048, flags=flags, work=w)
            assert int((bs != 0).sum()) == 0 and int((bl != 2048).sum()) == 0 and torch.equal(back, want), (flags, wb)
    # the C-ABI's own checks: alignment of d_work; the archive call needs its scratch
    q = bl.data_ptr()
    assert L.hdlz_inflate_batch_ws(flat.data_ptr(), d_off.data_ptr(), 0, 0, 3000, 0, 0, back.data_ptr(), 2048, q, q, w.data_ptr() + 8 if wb else 8, 4096, None) == 8
    assert L.hdlz_archive_batch_ws(back.data_ptr(), 2048, bl.data_ptr(), 3000, flat.data_ptr(), 64, d_off.data_ptr(), None, 0, None) == 8


def test_api_edge_cases(engine, oracle):
    """empty batches are no-ops, parameter errors are reported by the call (not by a kernel), minimal and
    unaligned shapes work: every entry point of include/hdlz.h"""
    import ctypes
    import torch
    from hdl_deflate_amd.constants import E_BAD_PARAM, pitch_for
    L = engine.lib
    st = torch.cuda.current_stream().cuda_stream
    # nblocks == 0: nothing is dereferenced
    assert L.hdlz_compress_batch(None, None, 0, 0, 0, 32, 10, None, 0, None, None, st) == 0
    assert L.hdlz_inflate_batch(None, None, 0, 0, 0, 0, 0, None, 0, None, None, st) == 0
    assert L.hdlz_compact_batch(None, 0, None, None, 0, None, st) == 0
    # parameter errors
    buf = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    i32 = torch.zeros(4, dtype=torch.int32, device="cuda")
    p, q = buf.data_ptr(), i32.data_ptr()
    assert L.hdlz_compress_batch(p, None, 64, 64, 1, 32, 10, p + 1, 256, q, q, st) == E_BAD_PARAM      # unaligned d_out
    assert L.hdlz_compress_batch(p, None, 64, 64, 1, 32, 10, p, 250, q, q, st) == E_BAD_PARAM          # out_pitch % 4
    assert L.hdlz_compress_batch(None, None, 64, 64, 1, 32, 10, p, 256, q, q, st) == E_BAD_PARAM       # null input
    assert b"null" in L.hdlz_last_error()
    assert L.hdlz_compress_stream(p, 1000, 0, 10, p, 4096, q, q, p, 1 << 20, st) == E_BAD_PARAM         # cwindow 0
    assert L.hdlz_compress_stream(p, 1000, 32, 10, p, 4096, q, q, p, 8, st) == E_BAD_PARAM               # workspace too small
    assert L.hdlz_compress_stream(p, 1000, 32, 10, p, 4096, q, q, p + 4, 1 << 20, st) == E_BAD_PARAM     # workspace alignment
    assert L.hdlz_stream_work_bytes(1 << 24) >= (1 << 24) // 2048 * 24 and L.hdlz_stream_work_bytes(1 << 40) == 0
    assert L.hdlz_out_bound(0) == 8 and L.hdlz_out_bound(2048) == 2312
    assert L.hdlz_status_string(10) == b"BAD_TREE" and L.hdlz_version() >= 1
    # the smallest stream the reference starts on (N = 5), through every compress entry point
    five = b"hello"
    ref = oracle.compress(five)[1]
    assert engine.compress_bytes(five) == (0, ref)
    d = torch.frombuffer(bytearray(five) + bytearray(27), dtype=torch.uint8).cuda()
    so, sl, ss = engine.compress_stream(d, 5)
    assert int(ss.item()) == 0 and so[:int(sl.item())].cpu().numpy().tobytes() == ref
    assert engine.inflate_bytes(ref) == (0, five)
    # a fixed-pitch batch whose pitch is not a multiple of 16 and whose rows are shorter than the pitch
    rows = torch.randint(97, 100, (33, 1000), dtype=torch.uint8, device="cuda")
    flat = torch.cat([rows.reshape(-1), torch.zeros(64, dtype=torch.uint8, device="cuda")])
    out, ol, stt = engine.compress_batch(flat, in_len=900, nblocks=33, out_pitch=pitch_for(1000))
    # (in_len given with a flat tensor: pitch == in_len, so block b starts at b * 900)
    h = flat.cpu().numpy()
    for b in (0, 1, 32):
        assert int(stt[b]) == 0 and out[b, :int(ol[b])].cpu().numpy().tobytes() == oracle.compress(h[b * 900:(b + 1) * 900].tobytes())[1]


def test_compress_small_blocks_packed_kernel(engine, oracle):
    """uniform blocks of 5..1024 bytes take the packed kernel (several blocks per wave-tile): every block
    against the oracle, block counts that leave partial groups, MATCH10 on/off, windows of 7 .. 256"""
    import torch
    from hdl_deflate_amd.data import make_blocks
    for n, B, cw, mm in [(5, 300, 32, 10), (6, 65, 32, 10), (31, 777, 32, 10), (32, 1000, 32, 10), (33, 513, 32, 5),
                         (100, 777, 32, 10), (255, 640, 16, 10), (256, 4096, 32, 10), (257, 333, 32, 10),
                         (500, 1001, 32, 10), (512, 2048, 7, 5), (1000, 130, 32, 10), (1024, 511, 32, 10),
                         # round 6: windows above 32 are packed too (the hashed finder on the packed tile)
                         (5, 300, 256, 10), (33, 513, 64, 10), (40, 777, 256, 10), (100, 777, 48, 10), (255, 640, 256, 5),
                         (256, 4096, 256, 10), (257, 333, 64, 10), (300, 900, 100, 10), (500, 1001, 256, 10), (512, 2048, 33, 10),
                         (777, 300, 200, 10), (1000, 130, 64, 5), (1024, 511, 256, 10)]:
        pitch = (n + 15) // 16 * 16
        raw = make_blocks(B, max(n, 64), "cuda", seed=n)[:, :pitch if pitch <= max(n, 64) else n]
        d = torch.zeros((B, pitch), dtype=torch.uint8, device="cuda")
        d[:, :n] = raw[:, :n]
        if n % 7 == 3:
            d[:, :n] = (d[:, :n] % 3) + 48            # match-dense variant
        if cw > 32 and n % 2 == 0:
            d[1::2] = d[0::2][: d[1::2].shape[0]]     # every second block repeats the one in front: the nearest candidate of its first
                                                      # positions lies in ANOTHER block of the packed tile and must not be taken
        out, ol, st = engine.compress_batch(d, in_len=n, cwindow=cw, maxmatch=mm)
        torch.cuda.synchronize()
        h, ho, hl, hs = d.cpu().numpy(), out.cpu().numpy(), ol.cpu().numpy(), st.cpu().numpy()
        flat = np.ascontiguousarray(h[:, :n]).reshape(-1)
        off = np.arange(B + 1, dtype=np.uint64) * n
        ro, rl, rs = oracle.compress_batch(flat, off, cw, mm, out_pitch=ho.shape[1], nthreads=8)
        assert (hs == rs).all() and (rs == 0).all(), (n, B)
        assert (hl == rl).all(), (n, B, int((hl != rl).sum()))
        mask = np.arange(ho.shape[1])[None, :] < hl[:, None]
        assert ((ho == ro) | ~mask).all(), (n, B)
        assert zlib.decompress(ho[B - 1, :hl[B - 1]].tobytes()) == h[B - 1, :n].tobytes()


def test_compress_small_ragged_blocks_packed_kernel(engine, oracle):
    """ragged batches of small blocks (in_off, any alignment) with a stated length bound take the packed kernel too:
    every block against the oracle, incl. blocks shorter than 5 bytes (SHORT_INPUT), empty blocks, a leading
    misalignment, and a block that violates the stated bound (BAD_PARAM in its status word only)"""
    import torch
    rng = np.random.default_rng(21)
    for maxlen, B, cw, mm, mis in [(40, 3000, 32, 10, 0), (300, 5000, 32, 10, 3), (700, 2000, 16, 5, 1), (1024, 1500, 32, 10, 7),
                                   (33, 4000, 5, 10, 2), (40, 3000, 256, 10, 1), (300, 5000, 64, 10, 2), (700, 2000, 256, 5, 3),
                                   (1024, 1500, 100, 10, 5)]:
        lens = rng.integers(0, maxlen + 1, size=B)
        lens[:8] = [0, 1, 4, 5, 6, maxlen, maxlen, 31]
        total = int(lens.sum())
        nsym = int(rng.choice([2, 4, 26, 256]))
        data = (rng.integers(0, nsym, size=total, dtype=np.uint8) + (0 if nsym == 256 else 97)).astype(np.uint8)
        flat = np.concatenate([np.zeros(mis, np.uint8), data, np.zeros(64, np.uint8)])
        off = (np.concatenate([[0], np.cumsum(lens)]) + mis).astype(np.int64)
        d_in, d_off = torch.from_numpy(flat).cuda(), torch.from_numpy(off).cuda()
        out, ol, st = engine.compress_batch(d_in, in_off=d_off, cwindow=cw, maxmatch=mm)       # engine passes max(len) as the bound
        torch.cuda.synchronize()
        ho, hl, hs = out.cpu().numpy(), ol.cpu().numpy(), st.cpu().numpy()
        ro, rl, rs = oracle.compress_batch(flat, off.astype(np.uint64), cw, mm, out_pitch=ho.shape[1], nthreads=8)
        assert (hs == rs).all(), (maxlen, np.nonzero(hs != rs)[0][:5], hs[hs != rs][:5], rs[hs != rs][:5])
        assert (hl == rl).all(), (maxlen, int((hl != rl).sum()))
        mask = np.arange(ho.shape[1])[None, :] < rl[:, None]
        assert ((ho == ro) | ~mask).all(), maxlen
    # a wrong bound: only the offending block reports it
    lens = np.array([100, 200, 50], dtype=np.int64)
    flat = np.concatenate([rng.integers(97, 100, size=350, dtype=np.uint8), np.zeros(64, np.uint8)])
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    out, ol, st = engine.compress_batch(torch.from_numpy(flat).cuda(), in_off=torch.from_numpy(off).cuda(), max_len=128)
    assert st.cpu().tolist() == [0, 8, 0] and int(ol[1]) == 0


def test_compress_misaligned_inputs(engine, oracle):
    import torch
    from hdl_deflate_amd.data import family_bytes
    blocks = [family_bytes(1 + k % 4, 700 + k, seed=k) for k in range(24)]   # odd lengths -> every alignment
    outs, st = _ragged(torch, engine, blocks)
    for b, o in zip(blocks, outs):
        assert o == oracle.compress(b)[1]


def test_compress_status_codes(engine):
    import torch
    blocks = [b"", b"a", b"ab", b"abc", b"abcd", b"abcde"]
    outs, st = _ragged(torch, engine, blocks)
    assert list(st) == [1, 1, 1, 1, 1, 0] and outs[:5] == [b""] * 5
    d = torch.zeros((2, 2048), dtype=torch.uint8, device="cuda")
    out, ol, st = engine.compress_batch(d, out_pitch=1024)      # < out_bound(2048)
    assert st.cpu().tolist() == [2, 2] and ol.cpu().tolist() == [0, 0]
    with pytest.raises(Exception):
        engine.compress_batch(d, cwindow=300)
    with pytest.raises(Exception):
        engine.compress_batch(d, maxmatch=7)


def test_compress_full_size_cfg2_properties(engine, oracle):
    """BASELINE cfg 2 shape at reduced count on the test box (65536 x 2 KiB = 128 MiB): every block must
    round-trip through the engine's own inflate, a sample must be bit-equal to the oracle and must
    round-trip through stock zlib, and lengths must respect the bound."""
    import torch
    from hdl_deflate_amd.data import make_blocks
    B, n = 65536, 2048
    d = make_blocks(B, n, "cuda", seed=3)
    out, ol, st = engine.compress_batch(d)
    assert int((st != 0).sum().item()) == 0
    assert int(ol.max().item()) <= 2312 and int(ol.min().item()) > 6
    # every row is a zlib stream followed by don't-care bytes up to the pitch: inflate stops at the
    # final block's EOB, so the padded rows can be fed back directly (fixed pitch, in_len = pitch)
    back, bl, bs = engine.inflate_batch(out, out_pitch=n)
    assert int((bs != 0).sum().item()) == 0 and int((bl != n).sum().item()) == 0
    assert torch.equal(back, d)
    h, ho, hl = d.cpu().numpy(), out.cpu().numpy(), ol.cpu().numpy()
    r = random.Random(1)
    for b in [0, 1, 2, 3, B - 1] + [r.randrange(B) for _ in range(200)]:
        blk = h[b].tobytes()
        z = ho[b, :hl[b]].tobytes()
        assert z == oracle.compress(blk)[1]
        assert zlib.decompress(z) == blk
    ratio = float(hl.sum()) / (B * n)
    assert 0.45 < ratio < 0.75          # expected ~0.59 for the 4-family mix


def test_compress_host_pipelined_archive(engine, oracle):
    """Engine.compress_host (pinned host batch -> chunks on three streams -> one host archive + lengths): every block's slice of the
    archive equals the oracle's stream, for chunk sizes that divide the batch, do not divide it, and exceed it; n < 5 is counted"""
    import torch
    from hdl_deflate_amd.data import make_blocks
    B, n = 1500, 700
    d = make_blocks(B, n, "cuda", seed=21)
    h_in = torch.empty((B, n), dtype=torch.uint8, pin_memory=True)
    h_in.copy_(d)
    torch.cuda.synchronize()
    host = h_in.numpy()
    ref = [oracle.compress(host[b].tobytes())[1] for b in range(B)]
    for chunk in (500, 448, 4096, None):
        h_arch, h_len, total, bad = engine.compress_host(h_in, chunk_blocks=chunk)
        assert bad == 0 and total == sum(len(z) for z in ref)
        hl = h_len.numpy().astype(np.int64)
        off = hl.cumsum() - hl
        arch = h_arch[:total].numpy().tobytes()
        for b in range(B):
            assert arch[off[b]:off[b] + hl[b]] == ref[b], (chunk, b)
    # ... and back: the host archive + the scan of its lengths is what inflate_host takes
    offs = np.zeros(B + 1, np.int64)
    np.cumsum(hl, out=offs[1:])
    h_z = torch.empty(total + 64, dtype=torch.uint8).pin_memory()
    h_z[:total].copy_(h_arch[:total])
    rows, rl, rs = engine.inflate_host(h_z, torch.from_numpy(offs), 704, flags=1, chunk_streams=640)
    assert int((rs != 0).sum()) == 0 and int((rl != n).sum()) == 0 and torch.equal(rows[:, :n], h_in)
    # the reference's CWINDOW = 256 build through the same path
    h_arch, h_len, total, bad = engine.compress_host(h_in[:300].contiguous().pin_memory(), cwindow=256, chunk_blocks=128)
    hl = h_len.numpy().astype(np.int64)
    off = hl.cumsum() - hl
    arch = h_arch[:total].numpy().tobytes()
    for b in range(0, 300, 7):
        assert arch[off[b]:off[b] + hl[b]] == oracle.compress(host[b].tobytes(), cwindow=256)[1], b
    h4 = torch.zeros((8, 4), dtype=torch.uint8).pin_memory()
    _, _, total, bad = engine.compress_host(h4)
    assert bad == 8 and total == 0


def test_inflate_host_pipelined(engine, oracle):
    """Engine.inflate_host (pinned host streams + offsets -> chunks on three streams -> pinned rows, lengths, statuses): good, cut and
    damaged stock-zlib streams of both strategies, chunk sizes that divide the batch, do not, and exceed it -- all against the oracle"""
    import torch
    from hdl_deflate_amd.data import make_blocks
    r = random.Random(5)
    B, n = 3000, 600
    h = make_blocks(B, n, "cpu", seed=33, families=(1, 2, 4)).numpy()
    zs = []
    for k in range(B):
        c = zlib.compressobj(6, zlib.DEFLATED, 15, 9, zlib.Z_FIXED if k % 3 else zlib.Z_DEFAULT_STRATEGY)
        z = c.compress(h[k].tobytes()) + c.flush()
        q = r.random()
        if q < 0.05:
            z = bytearray(z)
            z[r.randrange(2, len(z))] ^= 1 << r.randrange(8)
            z = bytes(z)
        elif q < 0.1:
            z = z[: r.randrange(0, len(z))]
        zs.append(z)
    off = np.zeros(B + 1, np.int64)
    np.cumsum([len(z) for z in zs], out=off[1:])
    flat = np.frombuffer(b"".join(zs) + bytes(64), dtype=np.uint8).copy()
    h_z = torch.from_numpy(flat).pin_memory()
    pitch = 608
    ref, rl, rs = oracle.inflate_batch(flat, off.astype(np.uint64), pitch, flags=0, nthreads=8)
    m = np.arange(pitch)[None, :] < rl[:, None]
    for chunk, d2h in ((1000, "copy"), (896, "kernel"), (5000, "copy"), (None, "kernel"), (1000, "kernel")):
        # (d2h = "kernel": the rows reach pinned host memory through hdlz_compact_batch / k_compact_host -- ADVICE r4)
        h_out, h_len, h_st = engine.inflate_host(h_z, torch.from_numpy(off), pitch, chunk_streams=chunk, d2h=d2h)
        assert np.array_equal(h_st.numpy().astype(np.uint32), rs), chunk
        assert np.array_equal(h_len.numpy().astype(np.uint32), rl), chunk
        assert np.array_equal(h_out.numpy()[m], ref[m]), chunk
    assert len(set(rs.tolist())) >= 2


def test_inflate_golden_vectors(engine):
    g = load_golden("inflate_vectors.json")
    for v in g["vectors"]:
        flags = 1 if "DYNAMIC=False" in v["build"] else 0
        obsize = 32768 if "OBSIZE=32768" in v["build"] else 512
        for mapping in MAPPINGS:                 # lane-per-stream (+ dynamic second pass) and wave-per-stream decoders
            st, out = engine.inflate_bytes(bytes.fromhex(v["z_hex"]), flags=flags | mapping, obsize=obsize)
            if v["error"] is None:
                assert st == 0 and out.hex() == v["out_hex"], (v["name"], mapping)
            else:
                assert st == 5 and out == b"", (v["name"], mapping)


def test_inflate_symbols_286_287_of_a_fixed_block(engine, oracle):
    """the executed reference's behaviour for symbols 286 / 287 of a fixed block in both builds (inflate_r3_vectors.json: the
    DYNAMIC=False build's single zero leaf, the end-of-input check in front of the length-symbol check): every mapping equals
    the oracle, which equals the recorded reference (tests/test_oracle_golden.py)"""
    g = load_golden("inflate_r3_vectors.json")
    for v in g["vectors"]:
        flags = 1 if "DYNAMIC=False" in v["build"] else 0
        z = bytes.fromhex(v["z_hex"])
        rc, ref = oracle.inflate(z, flags=flags, obsize=512)
        assert rc == (5 if "NO EOF" in v["error"] else 7)
        for mapping in MAPPINGS:
            st, out = engine.inflate_bytes(z, flags=flags | mapping, obsize=512)
            assert st == rc and out == ref, (v["name"], v["build"], mapping, st, rc)


def test_inflate_random_vs_oracle_and_zlib(engine, oracle):
    import torch
    r = random.Random(99)
    streams, plain = [], []
    for it in range(256):
        n = r.choice([0, 1, 5, 64, 300, 2048, 5000])
        alpha = r.choice([b"ab", b"abcdefgh", bytes(range(256)), b"0123456789 "])
        data = bytes(r.choice(alpha) for _ in range(n))
        co = zlib.compressobj(level=r.choice([0, 1, 6, 9]), strategy=zlib.Z_FIXED, wbits=15)
        z = co.compress(data[: n // 2]) + (co.flush(zlib.Z_FULL_FLUSH) if r.random() < 0.3 else b"") + \
            co.compress(data[n // 2:]) + co.flush()
        if it % 7 == 3:
            z = z[:-r.randrange(1, 5)]          # truncated trailer
        streams.append(z)
        plain.append(data)
    flat = b"".join(streams) + bytes(64)
    off = np.cumsum([0] + [len(s) for s in streams]).astype(np.int64)
    d_in = torch.frombuffer(bytearray(flat), dtype=torch.uint8).cuda()
    for mapping in MAPPINGS:
        out, ol, st = engine.inflate_batch(d_in, in_off=torch.from_numpy(off).cuda(), out_pitch=5008, flags=mapping)
        out, ol, st = out.cpu().numpy(), ol.cpu().numpy(), st.cpu().numpy()
        for k, (z, data) in enumerate(zip(streams, plain)):
            rc, ref = oracle.inflate(z)
            assert st[k] == rc, (k, st[k], rc, mapping)
            assert out[k, :ol[k]].tobytes() == ref, (k, mapping)
            if rc == 0:
                assert ref == data


def test_inflate_dynamic_streams_vs_oracle(engine, oracle):
    """stock-zlib default streams (dynamic trees, mixed with stored/fixed blocks) and damaged copies of
    them: status and bytes must equal the oracle's for every stream"""
    import torch
    r = random.Random(123)
    streams = []
    for it in range(384):
        n = r.choice([1, 10, 100, 1000, 5000, 20000])
        alpha = r.choice([b"ab", b"abcdefgh", bytes(range(256)), b"0123456789 ", DYN_TEXT[:64]])
        data = bytes(r.choice(alpha) for _ in range(n))
        co = zlib.compressobj(r.choice([0, 1, 6, 9]), zlib.DEFLATED, r.choice([9, 12, 15]),
                              strategy=r.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED]))
        z = co.compress(data[: n // 2]) + (co.flush(zlib.Z_FULL_FLUSH) if r.random() < 0.3 else b"") + \
            co.compress(data[n // 2:]) + co.flush()
        if it % 3 == 1:                       # damage one bit somewhere behind the zlib header
            zb = bytearray(z)
            zb[r.randrange(2, len(zb))] ^= 1 << r.randrange(8)
            z = bytes(zb)
        elif it % 11 == 5:
            z = z[:-r.randrange(1, 6)]
        streams.append(z)
    flat = b"".join(streams) + bytes(64)
    off = np.cumsum([0] + [len(s) for s in streams]).astype(np.int64)
    d_in = torch.frombuffer(bytearray(flat), dtype=torch.uint8).cuda()
    cap = 20000 + 16
    for mapping in MAPPINGS:
        out, ol, st = engine.inflate_batch(d_in, in_off=torch.from_numpy(off).cuda(), out_pitch=cap, flags=mapping)
        out, ol, st = out.cpu().numpy(), ol.cpu().numpy(), st.cpu().numpy()
        for k, z in enumerate(streams):
            rc, ref = oracle.inflate(z, out_cap=cap)
            assert st[k] == rc, (k, int(st[k]), rc, len(z), mapping)
            assert out[k, :ol[k]].tobytes() == ref, (k, mapping)


def test_inflate_dynamic_two_stage_cases(engine, oracle):
    """the lane mapping's second pass has two stages (k_inflate_tok<true, 144> / <true, 288>): streams that change sides in the
    middle -- a small-alphabet dynamic block first and a 256-symbol one behind it, a fixed block in front of dynamic ones, a stored
    block between them -- and a batch that mixes all of them with plain fixed streams; every stream against the oracle and zlib"""
    import torch
    r = random.Random(5)
    wide = bytes(min(255, int(abs(r.gauss(0, 70)))) for _ in range(6000))      # ~200 distinct values, skewed: a dynamic block of > 144 symbols
    assert len(set(wide)) > 150 and (zlib.compress(wide)[2] >> 1) & 3 == 2
    narrow = bytes(r.choice(b"abcdef") for _ in range(6000))
    rnd = bytes(r.getrandbits(8) for _ in range(3000))          # incompressible: zlib stores it
    def stream(parts, level=6):
        co = zlib.compressobj(level, zlib.DEFLATED, 15)
        z = b""
        for p_ in parts:
            z += co.compress(p_) + co.flush(zlib.Z_FULL_FLUSH)
        return z + co.flush()
    cases = [stream([narrow, wide]), stream([wide, narrow]), stream([b"ab", narrow, wide]), stream([narrow, rnd, wide, narrow]),
             stream([narrow]), stream([wide]), stream([b"x"]), stream([narrow[:100], wide[:300], narrow[:50]]),
             zlib.compressobj(6, zlib.DEFLATED, 15, 9, zlib.Z_FIXED).compress(narrow) + b""]
    co = zlib.compressobj(6, zlib.DEFLATED, 15, 9, zlib.Z_FIXED)
    cases[-1] = co.compress(narrow) + co.flush()
    btypes = set()
    for z in cases:
        btypes.add((z[2] >> 1) & 3)
    assert btypes >= {1, 2}
    sel = [cases[k % len(cases)] for k in range(640)]
    flat = b"".join(sel) + bytes(64)
    off = np.cumsum([0] + [len(z) for z in sel]).astype(np.int64)
    d_in = torch.frombuffer(bytearray(flat), dtype=torch.uint8).cuda()
    cap = 21040
    want = [oracle.inflate(z, out_cap=cap) for z in cases]
    for z, (rc, ref) in zip(cases, want):
        assert rc == 0 and ref == zlib.decompress(z)
    for mapping in MAPPINGS:
        out, ol, st = engine.inflate_batch(d_in, in_off=torch.from_numpy(off).cuda(), out_pitch=cap, flags=mapping)
        out, ol, st = out.cpu().numpy(), ol.cpu().numpy(), st.cpu().numpy()
        for k in range(len(sel)):
            rc, ref = want[k % len(cases)]
            assert st[k] == rc and out[k, :ol[k]].tobytes() == ref, (k, mapping, int(st[k]))


def test_inflate_auto_mapping_second_pass(engine, oracle):
    """the DEFAULT mapping on batches above HDLZ_INFLATE_WAVE_THRESHOLD: pass 1 one lane per stream; the streams with
    dynamic-tree blocks are counted on the device and redone one lane each when they are at least
    HDLZ_INFLATE_DYN_LANE_MIN (case 1: all 32768), else one wave each (case 2: every fourth).  Every stream against
    the oracle (a small pool of distinct streams -- good, damaged, cut -- repeated)."""
    import torch
    r = random.Random(77)
    pool_dyn, pool_fix = [], []
    for it in range(96):
        n = r.choice([40, 300, 2048, 6000])
        alpha = r.choice([b"abcdefgh", bytes(range(256)), b"0123456789 ", DYN_TEXT[:64]])
        data = bytes(r.choice(alpha) for _ in range(n))
        co = zlib.compressobj(r.choice([1, 6, 9]), zlib.DEFLATED, 15, strategy=zlib.Z_DEFAULT_STRATEGY if it % 2 else zlib.Z_FIXED)
        z = co.compress(data) + co.flush()
        if it % 8 == 3:
            zb = bytearray(z)
            zb[r.randrange(2, len(zb))] ^= 1 << r.randrange(8)
            z = bytes(zb)
        rc, ref = oracle.inflate(z, out_cap=6016)
        (pool_dyn if it % 2 else pool_fix).append((z, rc, ref))
    B, cap = 32768, 6016
    for case, pick in (("all dynamic", lambda k: pool_dyn[k % len(pool_dyn)]),
                       ("every fourth", lambda k: pool_dyn[(k // 4) % len(pool_dyn)] if k % 4 == 0 else pool_fix[k % len(pool_fix)])):
        sel = [pick(k) for k in range(B)]
        flat = b"".join(z for z, _, _ in sel) + bytes(64)
        off = np.cumsum([0] + [len(z) for z, _, _ in sel]).astype(np.int64)
        d_in = torch.frombuffer(bytearray(flat), dtype=torch.uint8).cuda()
        out, ol, st = engine.inflate_batch(d_in, in_off=torch.from_numpy(off).cuda(), out_pitch=cap)
        out, ol, st = out.cpu().numpy(), ol.cpu().numpy(), st.cpu().numpy()
        want_st = np.array([rc for _, rc, _ in sel]); want_len = np.array([len(ref) for _, _, ref in sel])
        assert (st == want_st).all() and (ol == want_len).all(), case
        for k in range(0, B, 37):
            assert out[k, :ol[k]].tobytes() == sel[k][2], (case, k)
        for k in range(192):                    # (every distinct stream at least once)
            assert out[k, :ol[k]].tobytes() == sel[k][2], (case, k)


def test_inflate_small_streams_every_kind(engine, oracle):
    """small streams of every kind (strategy / level / wbits, stored and multi-block streams, distance-1 runs, damaged and cut streams,
    capacities below the output size, both builds) through every mapping: status + length + bytes of every stream against the oracle
    (written for round 3's two-phase inflate, which is gone; the batch is kept for the kernels that stayed)"""
    import torch
    from hdl_deflate_amd import INFLATE_ASSUME_FIXED
    from hdl_deflate_amd.data import make_blocks
    r = random.Random(11)
    B = 2048
    for rd, (pitch, n) in enumerate(((2048, 2048), (516, 556), (1024, 512), (64, 64), (2048, 1900))):
        h = make_blocks(B, n, "cpu", seed=100 + rd, families=(1, 2, 3, 4)).numpy()
        zs = []
        for k in range(B):
            blk = h[k].tobytes()[: r.choice((n, n, n, r.randrange(0, n + 1)))]
            if r.random() < 0.1:
                blk = bytes(r.getrandbits(8) for _ in range(len(blk)))
            if r.random() < 0.05:
                blk = bytes([r.getrandbits(8)]) * len(blk)
            strat = r.choice((zlib.Z_FIXED, zlib.Z_FIXED, zlib.Z_FIXED, zlib.Z_DEFAULT_STRATEGY, zlib.Z_RLE, zlib.Z_HUFFMAN_ONLY))
            c = zlib.compressobj(r.choice((0, 1, 6, 9)), zlib.DEFLATED, r.choice((9, 12, 15)), 9, strat)
            z = (c.compress(blk[: len(blk) // 2]) + (c.flush(zlib.Z_FULL_FLUSH) if r.random() < 0.3 else b"") +
                 c.compress(blk[len(blk) // 2:]) + c.flush())
            q = r.random()
            if q < 0.08 and len(z) > 8:
                z = bytearray(z)
                z[r.randrange(2, len(z))] ^= 1 << r.randrange(8)
                z = bytes(z)
            elif q < 0.14:
                z = z[: r.randrange(0, len(z) + 1)]
            zs.append(z)
        lens = np.array([len(z) for z in zs], dtype=np.int64)
        off = np.zeros(B + 1, np.int64)
        np.cumsum(lens, out=off[1:])
        flat = np.frombuffer(b"".join(zs) + bytes(64), dtype=np.uint8).copy()
        zin, zoff = torch.from_numpy(flat).cuda(), torch.from_numpy(off).cuda()
        for fl in (0, INFLATE_ASSUME_FIXED):
            ref, rl, rs = oracle.inflate_batch(flat, off.astype(np.uint64), pitch, flags=fl, nthreads=8)
            m = np.arange(pitch)[None, :] < rl[:, None]
            for mapping in MAPPINGS:
                out, ol, st = engine.inflate_batch(zin, in_off=zoff, out_pitch=pitch, flags=fl | mapping)
                torch.cuda.synchronize()
                assert np.array_equal(st.cpu().numpy().astype(np.uint32), rs), (rd, fl, mapping)
                assert np.array_equal(ol.cpu().numpy().astype(np.uint32), rl), (rd, fl, mapping)
                assert np.array_equal(out.cpu().numpy()[m], ref[m]), (rd, fl, mapping)
            assert len(set(rs.tolist())) >= 3           # (the batch really holds good, cut and damaged streams)


def _empty_blocks_stream(data, nstored, nfixed, nstored2=1, level=6, strategy=None):
    """a zlib stream whose payload block(s) stand behind `nstored` empty stored blocks (5 bytes each once aligned), `nfixed` empty
    fixed blocks (3 header bits + the 7-bit EOB: what Z_PARTIAL_FLUSH writes) and `nstored2` >= 1 more empty stored blocks, the last
    of which re-aligns the stream to a byte -- hand-made bits, RFC 1951 3.2.3 / 3.2.4"""
    bits = []
    def put(v, n):
        for k in range(n):
            bits.append((v >> k) & 1)
    def empty_stored():
        put(0, 3)                                   # BFINAL = 0, BTYPE = 00
        while len(bits) % 8:
            bits.append(0)
        put(0x0000, 16); put(0xFFFF, 16)            # LEN = 0, NLEN
    for _ in range(nstored):
        empty_stored()
    for _ in range(nfixed):
        put(0, 1); put(1, 2); put(0, 7)             # BFINAL = 0, BTYPE = 01, EOB (7 zero bits)
    for _ in range(nstored2):
        empty_stored()
    head = bytes(sum(bits[i + k] << k for k in range(8)) for i in range(0, len(bits), 8))
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 8, zlib.Z_FIXED if strategy is None else strategy)
    raw = c.compress(data) + c.flush()
    return b"\x78\x9c" + head + raw + zlib.adler32(data).to_bytes(4, "big")


def test_inflate_long_runs_of_empty_blocks(engine, oracle):
    """a step of the 16-lane mapping walked through any number of empty blocks while its input FIFO advances one
    half per step -- ~26 empty stored blocks (or ~100 empty fixed ones) in a row read stale FIFO words.  Streams with 200 empty stored
    and 500 empty fixed blocks in front of the data (and shorter runs at every phase of the FIFO), every mapping and the default
    choices (one large stream: the whole-GPU path hands it to the serial decoder), against the oracle and stock zlib"""
    import torch
    from hdl_deflate_amd.data import make_blocks
    h = make_blocks(40, 2048, "cpu", seed=77, families=(1, 2, 4)).numpy()
    zs, plain = [], []
    shapes = [(200, 500, 1), (0, 500, 1), (200, 0, 1), (26, 0, 1), (27, 3, 2), (1, 100, 1), (0, 101, 3), (60, 60, 60)] + \
             [(k, 7 * k % 11, 1 + k % 3) for k in range(20, 52)]
    for k, (ns, nf, ns2) in enumerate(shapes):
        data = h[k % 40].tobytes()[: 2048 - 13 * (k % 7)]
        z = _empty_blocks_stream(data, ns, nf, ns2, strategy=zlib.Z_DEFAULT_STRATEGY if k % 5 == 4 else None)
        assert zlib.decompress(z) == data
        zs.append(z); plain.append(data)
    B = len(zs)
    off = np.zeros(B + 1, np.int64)
    np.cumsum([len(z) for z in zs], out=off[1:])
    flat = np.frombuffer(b"".join(zs) + bytes(64), dtype=np.uint8).copy()
    ref, rl, rs = oracle.inflate_batch(flat, off.astype(np.uint64), 2048, flags=0, nthreads=4)
    assert (rs == 0).all() and all(ref[k, :rl[k]].tobytes() == plain[k] for k in range(B))
    zin, zoff = torch.from_numpy(flat).cuda(), torch.from_numpy(off).cuda()
    m = np.arange(2048)[None, :] < rl[:, None]
    for mapping in (0,) + MAPPINGS:
        out, ol, st = engine.inflate_batch(zin, in_off=zoff, out_pitch=2048, flags=mapping)
        torch.cuda.synchronize()
        assert np.array_equal(st.cpu().numpy().astype(np.uint32), rs), mapping
        assert np.array_equal(ol.cpu().numpy().astype(np.uint32), rl), mapping
        assert np.array_equal(out.cpu().numpy()[m], ref[m]), mapping
    # one stream at a time, fixed pitch (the shape the port's STARTD has): >= HDLZ_INFLATE_PAR_MIN bytes takes the whole-GPU path first
    for k in (0, 1, 2, 7):
        z = np.frombuffer(zs[k] + bytes(64), dtype=np.uint8).copy()
        out, ol, st = engine.inflate_batch(torch.from_numpy(z).cuda().reshape(1, -1), in_len=len(zs[k]), out_pitch=2048)
        assert int(st.item()) == 0 and out[0, :int(ol.item())].cpu().numpy().tobytes() == plain[k], k


def test_inflate_length_binned_lanes(engine, oracle):
    """round 4: the lane mapping hands a ragged batch of more than HDLZ_INFLATE_BIN_MIN streams to the lanes in the order of their
    compressed-length class (pass 1 and the dynamic-tree pass).  Streams from a few bytes to ~100 KB, of every block type, damaged and
    cut ones among them, in random order: status, length and bytes of every stream equal the oracle's -- for batch sizes on both sides
    of the threshold -- and the rows stay where their stream index puts them."""
    import torch
    r = random.Random(21)
    text = bytes(r.choice(b"the quick brown fox jumps over the lazy dog 0123456789\n") for _ in range(120000))
    zs = []
    for k in range(900):
        n = int(10 ** r.uniform(0.3, 5.0))                  # 2 .. 100 000 plain bytes
        blk = text[r.randrange(0, 1000):][:n]
        if r.random() < 0.15:
            blk = bytes(r.getrandbits(8) for _ in range(min(n, 20000)))
        strat = r.choice((zlib.Z_FIXED, zlib.Z_FIXED, zlib.Z_DEFAULT_STRATEGY, zlib.Z_DEFAULT_STRATEGY, zlib.Z_HUFFMAN_ONLY))
        c = zlib.compressobj(r.choice((0, 1, 6, 9)), zlib.DEFLATED, 15, 9, strat)
        z = c.compress(blk) + c.flush()
        q = r.random()
        if q < 0.05 and len(z) > 8:
            z = bytearray(z); z[r.randrange(2, len(z))] ^= 1 << r.randrange(8); z = bytes(z)
        elif q < 0.10:
            z = z[: r.randrange(0, len(z) + 1)]
        zs.append(z)
    pitch = 100000
    from hdl_deflate_amd import INFLATE_LANE_PER_STREAM
    for B in (64, 65, 900):
        sel = zs[:B]
        off = np.zeros(B + 1, np.int64)
        np.cumsum([len(z) for z in sel], out=off[1:])
        flat = np.frombuffer(b"".join(sel) + bytes(64), dtype=np.uint8).copy()
        ref, rl, rs = oracle.inflate_batch(flat, off.astype(np.uint64), pitch, flags=0, nthreads=8)
        m = np.arange(pitch)[None, :] < rl[:, None]
        out, ol, st = engine.inflate_batch(torch.from_numpy(flat).cuda(), in_off=torch.from_numpy(off).cuda(), out_pitch=pitch,
                                           flags=INFLATE_LANE_PER_STREAM)
        torch.cuda.synchronize()
        assert np.array_equal(st.cpu().numpy().astype(np.uint32), rs), B
        assert np.array_equal(ol.cpu().numpy().astype(np.uint32), rl), B
        assert np.array_equal(out.cpu().numpy()[m], ref[m]), B
    assert len(set(rs.tolist())) >= 3 and int((rs == 0).sum()) > 700


def test_inflate_second_token_group_per_round(engine, oracle):
    """round 4: a round of the lane kernel decodes a SECOND group (up to three literals + a near match) for the lanes whose first group
    is still pending, when enough lanes of the wave can take one.  Streams that live on it -- literals only, short matches only, a
    literal run between two matches, near and far matches mixed, output rows that end inside a group (capacity), damaged and cut
    streams -- against the oracle: status, length and bytes of every stream; own streams (CWINDOW 32 / 256) and stock zlib ones."""
    import torch
    from hdl_deflate_amd import INFLATE_LANE_PER_STREAM
    r = random.Random(77)
    blocks = []
    for k in range(640):
        kind = k % 5
        n = r.choice((40, 700, 2048, 6000))
        if kind == 0:
            b = bytes(r.getrandbits(8) for _ in range(n))                                   # literals only
        elif kind == 1:
            b = bytes(r.choice(b"01") for _ in range(n))                                    # short matches only
        elif kind == 2:
            b = b"".join(b"Hi: %04d" % r.randrange(10000) for _ in range(n // 8 + 1))[:n]   # match, literal run, match
        elif kind == 3:
            w = [bytes(r.choice(b"abcdefgh") for _ in range(r.randrange(3, 12))) for _ in range(40)]
            b = b" ".join(r.choice(w) for _ in range(n // 5 + 1))[:n]                       # near and far matches
        else:
            b = bytes([r.randrange(256)]) * n                                               # one long overlapping copy chain
        blocks.append(b)
    zs = []
    for k, b in enumerate(blocks):
        if k % 3 == 0:
            c = zlib.compressobj(r.choice((1, 6, 9)), zlib.DEFLATED, 15, 9, zlib.Z_FIXED)
            z = c.compress(b) + c.flush()
        else:
            rc, z = oracle.compress(b, cwindow=256 if k % 3 == 1 else 32, maxmatch=10)
            assert rc == 0
        q = r.random()
        if q < 0.04 and len(z) > 8:
            z = bytearray(z); z[r.randrange(2, len(z))] ^= 1 << r.randrange(8); z = bytes(z)
        elif q < 0.08:
            z = z[: r.randrange(0, len(z) + 1)]
        zs.append(z)
    B = len(zs)
    off = np.zeros(B + 1, np.int64)
    np.cumsum([len(z) for z in zs], out=off[1:])
    flat = np.frombuffer(b"".join(zs) + bytes(64), dtype=np.uint8).copy()
    for pitch in (6000, 2044):                               # the second: most rows end inside a group (E_OUT_CAPACITY from the right token)
        ref, rl, rs = oracle.inflate_batch(flat, off.astype(np.uint64), pitch, flags=0, nthreads=8)
        m = np.arange(pitch)[None, :] < rl[:, None]
        out, ol, st = engine.inflate_batch(torch.from_numpy(flat).cuda(), in_off=torch.from_numpy(off).cuda(), out_pitch=pitch,
                                           flags=INFLATE_LANE_PER_STREAM)
        torch.cuda.synchronize()
        assert np.array_equal(st.cpu().numpy().astype(np.uint32), rs), pitch
        assert np.array_equal(ol.cpu().numpy().astype(np.uint32), rl), pitch
        assert np.array_equal(out.cpu().numpy()[m], ref[m]), pitch
    assert int((rs == 0).sum()) > 100 and len(set(rs.tolist())) >= 2


def test_inflate_group_mapping_ring_far_history_and_flush(engine, oracle):
    """k_inflate_grp (16 lanes per stream, hdlz_inflate_grp.hip): what is particular to it -- the 2 KiB history ring wrapping many times,
    copies that reach beyond the ring into the stream's own flushed output (distances up to 32 KiB), distance-1 / -2 / -3 ... -15 runs
    (the byte-l-mod-dist move), 258-byte matches (17 steps), the 1 KiB flushes and the tail, the 128-byte input halves on streams
    whose length is just around a multiple of them, stored and multi-block streams, capacity exactly at / below the output, damaged
    and cut streams: status, length and bytes against the oracle (COPY, /root/reference/deflate.py:1593-1659; D8)"""
    import torch
    r = random.Random(77)
    plains = []
    for n in (1, 5, 127, 128, 129, 255, 256, 1023, 1024, 1025, 2047, 2048, 2049, 4100, 40000, 70000):
        plains.append(bytes(r.choice(b"abcdefghij  \n") for _ in range(n)))
    base = bytes(r.randrange(256) for _ in range(3000))
    plains.append(base * 30)                                                   # far copies: 3000 back, again and again
    plains.append(bytes(r.randrange(256) for _ in range(33000)) * 2)           # distance ~ 32 KiB (wbits = 15)
    for d in range(1, 17):
        plains.append(bytes(r.randrange(256) for _ in range(d)) * (5000 // d))   # period d: overlapping copies of every small distance
    plains.append(bytes(100000))                                               # 258-byte matches at distance 1
    plains.append(DYN_TEXT * 5)
    zs = []
    for k, pl in enumerate(plains):
        c = zlib.compressobj(r.choice([1, 6, 9]), zlib.DEFLATED, 15, 9, zlib.Z_FIXED)
        z = c.compress(pl[:len(pl) // 2]) + (c.flush(zlib.Z_FULL_FLUSH) if k % 3 == 0 else b"") + c.compress(pl[len(pl) // 2:]) + c.flush()
        zs.append(z)
    zs.append(zlib.compress(bytes(r.randrange(256) for _ in range(5000)), 0))  # stored blocks
    damaged = []
    for z in zs[:20]:
        zb = bytearray(z)
        if len(zb) > 8:
            zb[r.randrange(2, len(zb))] ^= 1 << r.randrange(8)
        damaged.append(bytes(zb))
        damaged.append(z[:max(1, len(z) - r.randrange(1, 9))])
    for batch, cap in ((zs, 100096), (damaged, 100096), (zs, 2048), (zs, 4100)):
        off = np.zeros(len(batch) + 1, np.int64)
        np.cumsum([len(z) for z in batch], out=off[1:])
        flat = np.frombuffer(b"".join(batch) + bytes(64), dtype=np.uint8).copy()
        ref, rl, rs = oracle.inflate_batch(flat, off.astype(np.uint64), cap, flags=0, nthreads=4)
        out, ol, st = engine.inflate_batch(torch.from_numpy(flat).cuda(), in_off=torch.from_numpy(off).cuda(), out_pitch=cap, flags=64)
        torch.cuda.synchronize()
        ho, hl, hs = out.cpu().numpy(), ol.cpu().numpy().astype(np.uint32), st.cpu().numpy().astype(np.uint32)
        # (default-strategy text may hold dynamic blocks: the call's second pass decodes those -- same results)
        assert np.array_equal(hs, rs), (cap, np.nonzero(hs != rs)[0][:5], hs[hs != rs][:5], rs[hs != rs][:5])
        assert np.array_equal(hl, rl), cap
        for k in range(len(batch)):
            assert ho[k, :hl[k]].tobytes() == ref[k, :rl[k]].tobytes(), (cap, k)


def test_inflate_error_statuses(engine, oracle):
    cases = [b"\x78\x9c" + bytes([0x07]) + bytes(8),                                  # BTYPE 3
             zlib.compress(DYN_TEXT, 9),               # dynamic block
             b"\x78\x9c\x03",                                                         # short
             # symbol 287 of a fixed block followed by a 0 bit (NOT the zero leaf stat_leaves[483]) right where the input ends: the
             # end-of-input check comes first (found by the damaged-stream fuzz, round 3)
             bytes.fromhex("78dabbe9f8acad9ef167c05b081a17053d")]
    for z in cases:
        for mapping in MAPPINGS:
            st, out = engine.inflate_bytes(z, flags=mapping)
            rc, ref = oracle.inflate(z)
            assert st == rc and out == ref, mapping
    assert engine.inflate_bytes(cases[1]) == (0, DYN_TEXT)
    # output capacity
    z = zlib.compressobj(strategy=zlib.Z_FIXED).compress(b"x" * 1000)
    co = zlib.compressobj(strategy=zlib.Z_FIXED)
    z = co.compress(b"x" * 1000) + co.flush()
    st, out = engine.inflate_bytes(z, out_cap=512)
    assert st == 2


def test_round_trip_of_own_streams_through_an_archive(engine, oracle):
    """what bench.py's "configs[4] round trip" entry does, at reduced count and against the oracle: 64 KiB blocks ->
    STARTC -> one archive (ragged in_off) -> STARTD in every mapping; streams, lengths, statuses and bytes equal the oracle's, the
    bytes equal the blocks (the reference's harness inflates what it compressed, test_deflate.py:197-286)"""
    import torch
    from hdl_deflate_amd.data import make_blocks
    B, n = 192, 65536
    d = make_blocks(B, n, "cuda", seed=77)
    d[5, 1000:] = 0                                            # a block that compresses to almost nothing
    d[6] = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda")      # and one that does not compress at all
    for cw in (32, 256):
        rows, ol, st = engine.compress_batch(d, cwindow=cw)
        assert int((st != 0).sum()) == 0
        arch, offs = engine.compact(rows, ol)
        in_off = torch.cat([offs, (offs[-1:] + ol[-1:].to(torch.int64))])
        z = torch.cat([arch, torch.zeros(64, dtype=torch.uint8, device="cuda")])
        hz, hoff = z.cpu().numpy(), in_off.cpu().numpy().astype(np.uint64)
        host = d.cpu().numpy()
        for b in (0, 5, 6, B - 1):
            assert hz[int(hoff[b]):int(hoff[b + 1])].tobytes() == oracle.compress(host[b].tobytes(), cw, 10)[1]
        ref, rl, rs = oracle.inflate_batch(hz, hoff, n, flags=0, nthreads=8)
        assert (rs == 0).all() and (rl == n).all() and np.array_equal(ref, host)
        for mapping in (0,) + MAPPINGS:
            for fl in (0, 1):
                out, bl, bs = engine.inflate_batch(z, in_off=in_off, out_pitch=n, flags=mapping | fl)
                assert int((bs != 0).sum()) == 0 and int((bl != n).sum()) == 0 and torch.equal(out, d), (cw, mapping, fl)
    # a capacity one byte short: every stream reports OUT_CAPACITY, as the oracle does
    out, bl, bs = engine.inflate_batch(z, in_off=in_off, out_pitch=n - 4, flags=2)
    _, rl, rs = oracle.inflate_batch(hz, hoff, n - 4, flags=0, nthreads=8)
    assert np.array_equal(bs.cpu().numpy().astype(np.uint32), rs) and np.array_equal(bl.cpu().numpy().astype(np.uint32), rl)


def test_compact_archive(engine, oracle):
    """variable-length rows -> one contiguous archive; every stream must still inflate"""
    import torch
    from hdl_deflate_amd.data import make_blocks
    B, n = 1000, 1500
    d = make_blocks(B, n, "cuda", seed=77)
    out, ol, st = engine.compress_batch(d)
    arch, offs = engine.compact(out, ol)
    torch.cuda.synchronize()
    ha, ho, hl, hoff = arch.cpu().numpy(), out.cpu().numpy(), ol.cpu().numpy(), offs.cpu().numpy()
    assert len(ha) == int(hl.sum()) and (hoff == np.cumsum(hl) - hl).all()
    for b in range(B):
        assert ha[hoff[b]:hoff[b] + hl[b]].tobytes() == ho[b, :hl[b]].tobytes()
    # the archive + offsets are a valid ragged inflate input
    in_off = torch.cat([offs, (offs[-1:] + ol[-1:].to(torch.int64))])
    back, bl, bs = engine.inflate_batch(torch.cat([arch, torch.zeros(64, dtype=torch.uint8, device="cuda")]),
                                        in_off=in_off, out_pitch=1504)
    assert int((bs != 0).sum()) == 0 and torch.equal(back[:, :n], d)


def test_archive_scan_and_gather_in_one_launch(engine):
    """hdlz_archive_batch (round 5): offsets by a device-side decoupled look-back + the gather, one launch -- against the two-pass form
    (exclusive scan on the host + hdlz_compact_batch) byte for byte, for batches below / at / above a tile of 256 rows and far above
    the number of tiles one look-back step covers (64); the offsets are a valid ragged inflate input; a too small archive is reported
    through offsets[B] and nothing is written beyond it"""
    import torch
    from hdl_deflate_amd.data import make_blocks
    r = random.Random(61)
    for B, pitch in ((0, 64), (1, 64), (255, 100), (256, 96), (257, 2320), (5000, 304), (70000, 64), (40000, 2320)):
        lens = np.array([r.randrange(0, pitch + 1) for _ in range(B)], dtype=np.int32)
        if B > 10:
            lens[3] = 0; lens[B - 1] = pitch; lens[7:9] = 0
        rows = torch.randint(0, 256, (B, pitch), dtype=torch.uint8, device="cuda")
        d_len = torch.from_numpy(lens).cuda()
        arch, offs = engine.archive(rows, d_len)
        torch.cuda.synchronize()
        ho = offs.cpu().numpy()
        ref_off = np.zeros(B + 1, np.int64); np.cumsum(lens, out=ref_off[1:])
        assert np.array_equal(ho, ref_off), (B, pitch)
        if B:
            a2, _ = engine.compact(rows, d_len)
            assert torch.equal(arch[:int(ref_off[-1])], a2), (B, pitch)
    # a compress job end to end: archive + offsets feed the ragged inflate directly
    B, n = 3000, 1500
    d = make_blocks(B, n, "cuda", seed=78)
    out, ol, st = engine.compress_batch(d)
    arch, offs = engine.archive(out, ol)
    a2, o2 = engine.compact(out, ol)
    total = int(offs[-1].item())
    assert total == a2.numel() and torch.equal(arch[:total], a2) and torch.equal(offs[:-1], o2)
    back, bl, bs = engine.inflate_batch(torch.cat([arch[:total], torch.zeros(64, dtype=torch.uint8, device="cuda")]), in_off=offs, out_pitch=1504)
    assert int((bs != 0).sum()) == 0 and torch.equal(back[:, :n], d)
    # capacity: rows that would end beyond the archive are skipped, the total still says what was needed
    small = torch.full((total // 2,), 0xEE, dtype=torch.uint8, device="cuda")
    guard = small.clone()
    _, offs3 = engine.archive(out, ol, archive=small)
    torch.cuda.synchronize()
    assert int(offs3[-1].item()) == total
    hs, ha, hof, hl = small.cpu().numpy(), a2.cpu().numpy(), o2.cpu().numpy(), ol.cpu().numpy()
    fit = hof + hl <= small.numel()
    last = int((hof + hl)[fit].max())
    assert np.array_equal(hs[:last], ha[:last]) and (hs[last:] == 0xEE).all()


def test_compact_into_pinned_host_memory(engine):
    """hdlz_compact_batch with a PINNED HOST archive (k_compact_host, chosen by hipPointerGetAttributes -- ADVICE r4: nothing exercised
    it): rows of every length 0 .. 67 and multiples of 16, at 16-byte aligned destinations (the 16-byte branch: pitch and offsets
    multiples of 16) and at odd ones (head / body / tail branch), against the same gather on the device and against the rows"""
    import torch
    r = random.Random(16)
    for pitch, aligned in ((96, True), (96, False), (100, False), (2320, True), (2312, False)):
        lens = list(range(0, min(68, pitch))) + [16, 32, 48, 64, 80, pitch, pitch - 1, pitch - 15][:8] + \
               [r.randrange(0, pitch + 1) for _ in range(200)]
        lens = [min(x, pitch) for x in lens]
        B = len(lens)
        rows = torch.randint(0, 256, (B, pitch), dtype=torch.uint8, device="cuda")
        d_len = torch.tensor(lens, dtype=torch.int32, device="cuda")
        if aligned:
            offs = [16 * k * ((pitch + 15) // 16) for k in range(B)]                    # every destination 16-byte aligned
        else:
            offs = list(np.cumsum([0] + [x + (k % 3) for k, x in enumerate(lens)])[:-1] + 1)     # odd, ragged
        total = int(offs[-1] + lens[-1] + 16)
        d_off = torch.tensor(offs, dtype=torch.int64, device="cuda")
        h_arch = torch.full((total,), 0xEE, dtype=torch.uint8).pin_memory()
        d_arch = torch.full((total,), 0xEE, dtype=torch.uint8, device="cuda")
        for arch in (h_arch, d_arch):
            rc = engine.lib.hdlz_compact_batch(rows.data_ptr(), pitch, d_len.data_ptr(), d_off.data_ptr(), B, arch.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream)
            assert rc == 0
        torch.cuda.synchronize()
        ha, da, hr = h_arch.numpy(), d_arch.cpu().numpy(), rows.cpu().numpy()
        assert np.array_equal(ha, da), (pitch, aligned)
        expect = np.full(total, 0xEE, dtype=np.uint8)                                    # nothing outside the rows' bytes is touched
        for b in range(B):
            expect[offs[b]:offs[b] + lens[b]] = hr[b, :lens[b]]
        assert np.array_equal(ha, expect), (pitch, aligned)


def test_port_adapter_on_gpu_modes(engine):
    """the reference's own test flow (test_deflate.py:105-286) through the port adapter on the HIP engine"""
    from test_port_protocol import run_mode_flow
    g = load_golden("port_modes.json")
    for rec in g["modes"]:
        run_mode_flow(rec, engine)


def test_port_startd_of_a_stock_zlib_stream_and_startc_back(engine):
    """the reference's own use, at a size its default build handles (DYNAMIC=True, deflate.py:32): STARTD of ONE stock-zlib level-6 stream
    through the ten ports -- 48 KB of dynamic-tree blocks: hdlz_inflate_any.hip on the whole GPU behind the adapter -- and STARTC of what
    came out, on the same DUT, read back and inflated by stock zlib (test_deflate.py:115-286, the harness body of tests/port_harness.py)"""
    import time
    from test_port_protocol import make_dut, stream_leg, STARTC, STARTD
    r = random.Random(21)
    words = [bytes(r.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(r.randint(2, 9))) for _ in range(400)]
    plain = bytearray()
    while len(plain) < 120000:
        plain += r.choice(words) + b" "
    plain = bytes(plain[:120000])
    z = zlib.compress(plain, 6)
    assert len(z) >= 16384 and (z[2] >> 1) & 3 == 2                  # a dynamic first block: the chain for any block types
    dut, s = make_dut(engine)
    inf, total = stream_leg(dut, s, z, STARTD)
    assert inf == plain and total == len(plain)
    comp, total = stream_leg(dut, s, plain, STARTC)
    assert total == len(comp) and zlib.decompress(comp) == plain
    # ... and the engine call behind STARTD took the whole-GPU path: one wave needs ~10 ms for these 120 KB
    st, got = engine.inflate_bytes(z)
    t0 = time.time()
    st, got = engine.inflate_bytes(z)
    assert (st, got) == (0, plain) and time.time() - t0 < 0.006


def test_inflate_oneblock_lowlut_builds_and_empty_distance_code(engine):
    """ONEBLOCK / LOWLUT reference builds (variants_vectors.json, oracle/gen_golden_r2.py) through both decoders; a dynamic
    block with an empty distance code (RFC1951 3.2.7) is accepted like zlib accepts it"""
    from conftest import empty_distance_stream
    g = load_golden("variants_vectors.json")
    for v in g["oneblock"]:
        flags = (1 if ("DYNAMIC=False" in v["build"] or "LOWLUT" in v["build"]) else 0) | 8
        for mapping in MAPPINGS:
            st, out = engine.inflate_bytes(bytes.fromhex(v["z_hex"]), flags=flags | mapping, obsize=512)
            assert st == 0 and out.hex() == v["out_hex"], (v["build"], v["name"], mapping)
    z = empty_distance_stream()
    for mapping in MAPPINGS:
        st, out = engine.inflate_bytes(z, flags=mapping)
        assert st == 0 and out == b"aaaaa" == zlib.decompress(z)


def test_inflate_bytes_high_ratio_streams(engine):
    """deflate expands up to 1032:1 -- the default capacity of inflate_bytes / the port must hold such streams"""
    from test_port_protocol import make_dut, stream_leg
    from hdl_deflate_amd import STARTD
    for data in (b"Hello World! 1 " * 20000, bytes(1 << 20)):
        z = zlib.compress(data, 9)
        assert len(data) > 260 * len(z) and len(data) > (1 << 16)
        st, out = engine.inflate_bytes(z)
        assert st == 0 and out == data
    z = zlib.compress(bytes(200000), 9)
    dut, s = make_dut(engine)
    res, total = stream_leg(dut, s, z, STARTD)
    assert total == 200000 and res == bytes(200000)


def test_port_streaming_mode_on_gpu(engine):
    """SURVEY 8(f) rank 3 on the HIP engine: the streaming port (bounded circular iram / oram, a resumable kernel call per
    filled window, the OBSIZE hold) under the reference's harness with a slow reader / slow writer -- replay of the
    trajectories recorded from the executed reference -- and through the six reference test modes"""
    from test_port_protocol import (run_backpressure_fixture, run_streaming_mode_flows, make_dut, stream_leg,
                                    run_writer_timing_fixtures, run_lagging_reader_fixtures)
    from hdl_deflate_amd import STARTC, STARTD, HdlzRangeError
    g = load_golden("variants_vectors.json")
    for v in g["backpressure"]:
        stats = run_backpressure_fixture(v, engine)
        assert stats["cycles"] > 0
    # the fixtures on which the port and the executed reference knowingly DIFFER (streaming_r3_vectors.json: writer_timing,
    # lagging_reader), on the hardware path: the documented relation, pinned
    run_writer_timing_fixtures(engine)
    run_lagging_reader_fixtures(engine)
    run_streaming_mode_flows(engine)
    # a stream much longer than both rings, dynamic trees, tiny OBSIZE: every byte still arrives, in order
    data = DYN_TEXT * 30
    z = zlib.compress(data, 6)
    dut, s = make_dut(engine, streaming=True, stream_obsize=512, ibsize=64, window=32)
    res, total = stream_leg(dut, s, z, STARTD, read_every=2)
    assert res == data and dut.launches > 20
    with pytest.raises(ValueError):                          # a 258-byte copy never fits a smaller output buffer (deflate.py:1597)
        dut, s = make_dut(engine, streaming=True, stream_obsize=64)
        stream_leg(dut, s, z, STARTD)
    dut, s = make_dut(engine, streaming=True, stream_obsize=128, ibsize=128, cwindow=32)
    res, total = stream_leg(dut, s, data, STARTC, write_every=2)
    assert zlib.decompress(res) == data and res == engine.compress_bytes(data)[1]
    # LMAX = 16 (LOWLUT build): 65 535 bytes pass, one more does not fit the counters
    for v in g["lmax16"]:
        dut, s = make_dut(engine, lmax=16, inflate_flags=1 | 8, streaming=True)
        if v["error"] is None:
            res, total = stream_leg(dut, s, bytes.fromhex(v["z_hex"]), STARTD)
            assert total == 65535
        else:
            with pytest.raises(HdlzRangeError):
                stream_leg(dut, s, bytes.fromhex(v["z_hex"]), STARTD)


def test_block_chaining_beyond_lmax(engine):
    """inputs longer than the reference's 2^LMAX counters are chained block by block -- one complete
    zlib stream per block, every block inside the counter range -- and inflate back to the input"""
    import torch
    from hdl_deflate_amd.chain import compress_chained, inflate_chained, plan_blocks, MAX_BLOCK
    from hdl_deflate_amd.data import make_blocks
    assert plan_blocks(100, 64) == [(0, 64), (64, 36)] and plan_blocks(130, 64) == [(0, 64), (64, 48), (112, 18)]
    assert MAX_BLOCK % 16 == 0 and 6 + (9 * MAX_BLOCK + 17) // 8 < (1 << 24)
    n = (40 << 20) + 3                                       # 2.5 x the 16 MiB a single START can address, tail of 3 bytes
    flat = torch.cat([make_blocks(20480, 2048, "cuda", seed=5).view(-1), torch.tensor([7, 8, 9], dtype=torch.uint8, device="cuda")])
    assert flat.numel() == n
    archive, offsets, lens = compress_chained(engine, flat, block=8 << 20)
    torch.cuda.synchronize()
    off = offsets.tolist()
    assert len(off) == 7 and off[-1] == archive.numel()      # 4 full blocks + 2 end pieces (the 3-byte tail was avoided)
    host, ha = flat.cpu().numpy().tobytes(), archive.cpu().numpy().tobytes()
    pos = 0
    for b in range(6):
        piece = zlib.decompress(ha[off[b]:off[b + 1]])       # every piece is a complete stock-zlib-readable stream
        assert piece == host[pos:pos + len(piece)] and 5 <= len(piece) <= (8 << 20) and off[b + 1] - off[b] < (1 << 24)
        pos += len(piece)
    assert pos == n
    back = inflate_chained(engine, archive, offsets, block=8 << 20)
    assert torch.equal(back, flat)
