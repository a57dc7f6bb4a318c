"""GPU (-m gpu): EVERY unit of the two headline configurations against the oracle -- not a sample.
  * BASELINE configs[1]: all 2^20 blocks of 2 KiB (the bench workload, same generator and seed) compressed on the GPU are
    compared byte for byte with oracle.compress_batch run on all host cores (~3 s on the GPU box's 256 threads);
  * BASELINE configs[3]: all 2^20 stock-zlib Z_FIXED streams inflated on the GPU, compared with oracle.inflate_batch.
Done in slices of 2^17 units to bound host memory."""
import multiprocessing as mp
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

B_TOTAL, N, SLICE = 1 << 20, 2048, 1 << 17


def _cores():
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


def test_configs1_every_block_vs_oracle(engine, oracle):
    import torch
    from hdl_deflate_amd.data import make_blocks
    from hdl_deflate_amd.constants import pitch_for
    d_in = make_blocks(B_TOTAL, N, "cuda", seed=0)                  # bench.py's headline workload
    out, ol, st = engine.compress_batch(d_in)
    torch.cuda.synchronize()
    assert int((st != 0).sum().item()) == 0
    pitch = pitch_for(N)
    cores = _cores()
    total_out = 0
    for s0 in range(0, B_TOTAL, SLICE):
        h = d_in[s0:s0 + SLICE].cpu().numpy()
        off = np.arange(SLICE + 1, dtype=np.uint64) * N
        ref, rl, rs = oracle.compress_batch(h.reshape(-1), off, 32, 10, out_pitch=pitch, nthreads=cores)
        assert (rs == 0).all()
        gl = ol[s0:s0 + SLICE].cpu().numpy().astype(np.uint32)
        assert (gl == rl).all(), "output lengths differ in slice %d" % s0
        g = out[s0:s0 + SLICE].cpu().numpy()
        mask = np.arange(pitch, dtype=np.uint32)[None, :] < rl[:, None]      # compare exactly the out_len bytes of every row
        assert np.array_equal(np.where(mask, g, 0), np.where(mask, ref, 0)), "bytes differ in slice %d" % s0
        total_out += int(rl.sum())
    assert 0.45 < total_out / (B_TOTAL * N) < 0.75


def _zfixed(args):
    import zlib
    buf, n = args
    out = []
    for k in range(0, len(buf), n):
        co = zlib.compressobj(strategy=zlib.Z_FIXED, wbits=15)
        out.append(co.compress(buf[k:k + n]) + co.flush())
    return b"".join(out), [len(z) for z in out]


def test_configs3_every_stream_vs_oracle(engine, oracle):
    import torch
    from hdl_deflate_amd.data import make_blocks
    d_plain = make_blocks(B_TOTAL, N, "cuda", seed=4, families=(1, 2, 4))      # bench.py's configs[3] workload
    cores = _cores()
    nproc = min(cores, 64)
    for s0 in range(0, B_TOTAL, SLICE):
        host = d_plain[s0:s0 + SLICE].cpu().numpy()
        per = (SLICE + nproc * 2 - 1) // (nproc * 2)
        with mp.get_context("fork").Pool(nproc) as pool:
            parts = pool.map(_zfixed, [(host[k:k + per].tobytes(), N) for k in range(0, SLICE, per)])
        lens = np.fromiter((l for _, ls in parts for l in ls), dtype=np.int64, count=SLICE)
        off = np.zeros(SLICE + 1, np.int64)
        np.cumsum(lens, out=off[1:])
        flat = np.frombuffer(b"".join(p for p, _ in parts) + bytes(64), dtype=np.uint8)
        d_in = torch.from_numpy(flat.copy()).cuda()
        d_off = torch.from_numpy(off).cuda()
        for flags in (1, 1 | 2, 1 | 64):                # DYNAMIC=False semantics: the default mapping, the lane kernel by hint, 16 lanes per stream
            out, ol, st = engine.inflate_batch(d_in, in_off=d_off, out_pitch=N, flags=flags)
            torch.cuda.synchronize()
            assert int((st != 0).sum().item()) == 0 and int((ol != N).sum().item()) == 0
            assert torch.equal(out, d_plain[s0:s0 + SLICE])
        ref, rl, rs = oracle.inflate_batch(flat, off.astype(np.uint64), N, flags=1, nthreads=cores)
        assert (rs == 0).all() and (rl == N).all()
        assert np.array_equal(ref, out.cpu().numpy()), "GPU output differs from the oracle in slice %d" % s0
