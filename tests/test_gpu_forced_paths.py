"""GPU (-m gpu): the compress kernels' FALLBACK paths, which this hardware / ordinary data never take, against the oracle.

lib/libhdlz_forced.so (built by __graft_entry__.build() / `csrc/build.sh forced`) is libhdlz with
  * -DHDLZ_HASH_FORCE_REORDER: every group of the wide-window finder takes the 64-step reorder loop (the answer to "what if the LDS
    applied the lanes of one returning ds_max out of lane order" -- hdlz_compress_common.h, match_search_hash; matcher3 x CWINDOW +
    first-set-bit pick, /root/reference/deflate.py:407-421, :982-989);
  * -DHDLZ_CHAIN_FORCE_SERIAL: every tile composes its 64 run functions on the scalar unit instead of resolving constant functions by
    DPP steps (chain_skips; the greedy step di += match / di += 1, deflate.py:960, :1008).
The library is chosen through HDLZ_LIB in a SUBPROCESS (one process loads one libhdlz), which runs the compress parity tests of the
normal suite -- golden vectors of the executed reference, random / adversarial / wide-window inputs against the oracle, the packed
small-block kernel, the multi-wave stream passes and the resumable sessions."""
import os
import random
import subprocess
import sys
import zlib

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FORCED = os.path.join(REPO, "hdl_deflate_amd", "lib", "libhdlz_forced.so")


def test_compress_parity_on_the_forced_fallback_paths():
    assert os.path.exists(FORCED), "lib/libhdlz_forced.so is not built: hdl_deflate_amd/csrc/build.sh forced"
    env = dict(os.environ, HDLZ_LIB=FORCED)
    sel = ("test_compress_golden_vectors_bit_exact or test_compress_large_golden_vectors_bit_exact or test_compress_random_vs_oracle "
           "or test_compress_families_fixed_pitch_and_sizes or test_compress_every_block_vs_oracle_dense_matches "
           "or test_compress_one_tile_block_starts or test_compress_wide_windows_large_blocks_every_block "
           "or test_compress_hashed_finder_adversarial_inputs or test_compress_stream_multiwave_vs_oracle "
           "or test_compress_small_blocks_packed_kernel or test_compress_small_ragged_blocks_packed_kernel "
           "or test_compress_misaligned_inputs or test_compress_session_equals_one_shot or test_compress_skip_chain")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "tests/test_gpu_sessions.py",
                        "tests/test_gpu_forced_paths.py", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-k", sel],
                       cwd=REPO, env=env, capture_output=True, text=True, timeout=2400)
    tail = r.stdout[-3000:] + r.stderr[-2000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and " failed" not in r.stdout, tail


def test_compress_skip_chain_every_path(engine, oracle):
    """chain_skips: constant run functions (ordinary data: resolved at once), chains of non-constant ones shorter and longer than
    CHAIN_STEPS (maximal matches back to back through whole runs: zeros, short periods -- the scalar composition), and both in one
    tile; one-tile and multi-tile blocks, entry skips carried across tiles."""
    import numpy as np
    import torch
    r = random.Random(515)
    blocks = []
    text = bytes(r.choice(b"etaoin shrdlu") for _ in range(70000))
    for n in (2048, 2047, 6000, 65536, 70000):
        blocks.append(bytes(n))                                            # one match chain from end to end
        blocks.append((b"abc" * (n // 3 + 1))[:n])
        blocks.append((b"0123456789" * (n // 10 + 1))[:n])                 # period = the longest match
        blocks.append((b"0123456789a" * (n // 11 + 1))[:n])
        for runs in (1, 2, 3, 4, 5, 6, 9, 17):                             # `runs` periodic 32-byte runs between stretches of text
            b = bytearray(text[:n])
            for k in range(0, n - 32 * runs - 64, 2048 // 3):
                b[k + 40:k + 40 + 32 * runs] = (b"xyzw" * 8 * runs)
            blocks.append(bytes(b))
        blocks.append(text[:n])
    flat = b"".join(blocks) + bytes(64)
    off = np.cumsum([0] + [len(b) for b in blocks]).astype(np.int64)
    d_in = torch.frombuffer(bytearray(flat), dtype=torch.uint8).cuda()
    for cw, mm in ((32, 10), (32, 5), (20, 10), (64, 10), (256, 10)):
        out, ol, st = engine.compress_batch(d_in, in_off=torch.from_numpy(off).cuda(), cwindow=cw, maxmatch=mm)
        torch.cuda.synchronize()
        out, ol, st = out.cpu().numpy(), ol.cpu().numpy(), st.cpu().numpy()
        for k, b in enumerate(blocks):
            rc, ref = oracle.compress(b, cw, mm)
            assert st[k] == rc == 0 and out[k, :ol[k]].tobytes() == ref, (cw, mm, k, len(b))
    # the packed small-block kernel and the one-tile kernel (fixed pitch)
    for n in (256, 1024, 2048):
        rows = [bytes(n), (b"ab" * n)[:n], text[:n], (b"0123456789" * n)[:n]] * 8
        d = torch.frombuffer(bytearray(b"".join(rows)), dtype=torch.uint8).cuda().view(len(rows), n)
        out, ol, st = engine.compress_batch(d)
        torch.cuda.synchronize()
        out, ol = out.cpu().numpy(), ol.cpu().numpy()
        for k, b in enumerate(rows):
            assert out[k, :ol[k]].tobytes() == oracle.compress(b)[1], (n, k)
            assert zlib.decompress(out[k, :ol[k]].tobytes()) == b
