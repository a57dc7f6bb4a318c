"""CPU, world_size 2 over gloo: the N>1 path of SURVEY 8(e) -- contiguous block shards, all-gather of
per-block output lengths, exclusive scan for archive offsets.  No payload collective exists."""
import datetime
import os
import queue
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hdl_deflate_amd.shard import shard_range, gather_lengths, archive_offsets, gather_archive, LengthGather


def test_shard_range_covers_everything():
    for nb in (0, 1, 7, 8, 9, 131072, 1000003):
        for w in (1, 2, 3, 4, 8):
            rs = [shard_range(nb, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == nb
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1


def _rendezvous(rank, world, port, q):
    """init_process_group; a failure HERE (gloo's full-mesh connect on a busy host) is reported as infrastructure, not as a result"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    except Exception as e:          # noqa: BLE001
        q.put((rank, "rendezvous", repr(e)))
        raise


def _spawn_world(worker, world, extra, timeout):
    """Run `worker(rank, world, port, *extra, q)` on `world` spawned processes and return their queue entries.  A rendezvous that fails or
    never completes is retried on a fresh port (twice); a worker that fails AFTER it -- an assertion of the test -- is not."""
    ctx = mp.get_context("spawn")
    for attempt in range(3):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        q = ctx.Queue()
        ps = [ctx.Process(target=worker, args=(r, world, port) + tuple(extra) + (q,)) for r in range(world)]
        for p in ps:
            p.start()
        res, infra = [], None
        try:
            for _ in ps:
                item = q.get(timeout=timeout)
                if len(item) == 3 and item[1] == "rendezvous":
                    infra = item
                    break
                res.append(item)
        except queue.Empty:
            infra = ("?", "rendezvous", "no answer within %d s" % timeout) if not res else None
            if infra is None:
                for p in ps:
                    p.terminate()
                raise
        if infra is None:
            for p in ps:
                p.join(60)
                assert p.exitcode == 0
            return res
        for p in ps:                # the exact processes this call started
            p.terminate()
        for p in ps:
            p.join(30)
        if attempt == 2:
            raise RuntimeError("gloo rendezvous failed three times: %r" % (infra,))


def _worker(rank, world, port, nblocks, q):
    _rendezvous(rank, world, port, q)
    try:
        from oracle import oracle as O
        from hdl_deflate_amd.data import family_bytes
        b0, b1 = shard_range(nblocks, rank, world)
        blocks = [family_bytes(1 + b % 4, 300 + (b % 5), seed=b, counter0=16 * b) for b in range(b0, b1)]
        lens = torch.tensor([len(O.compress(x)[1]) for x in blocks], dtype=torch.int32)
        all_len = gather_lengths(lens, nblocks)
        lg = LengthGather(nblocks, "cpu")                 # the reusable form bench.py holds: same answer, twice
        assert lg.gather(lens).tolist() == all_len.tolist() and lg.gather(lens).tolist() == all_len.tolist()
        offs, total = archive_offsets(all_len)
        # payload gather of the per-rank archives (8(f) rank 2): concatenation = the global archive
        mine = torch.frombuffer(bytearray(b"".join(O.compress(x)[1] for x in blocks)), dtype=torch.uint8)
        whole = gather_archive(mine, mine.numel())
        dist.barrier()                   # nobody tears the process group (rank 0: the store) down while another rank is still at work
        q.put((rank, all_len.tolist(), offs.tolist(), total, bytes(whole.numpy().tobytes())))
    finally:
        dist.destroy_process_group()


def test_length_allgather_world2():
    _run_world2(11)                  # uneven shards: 6 + 5 (padded gather)
    _run_world2(12)                  # equal shards: gathered straight into the result (the BASELINE configs[4] shape)


def _run_world2(nblocks):
    res = _spawn_world(_worker, 2, (nblocks,), 120)
    from oracle import oracle as O
    from hdl_deflate_amd.data import family_bytes
    want = [len(O.compress(family_bytes(1 + b % 4, 300 + (b % 5), seed=b, counter0=16 * b))[1]) for b in range(nblocks)]
    blobs = [O.compress(family_bytes(1 + b % 4, 300 + (b % 5), seed=b, counter0=16 * b))[1] for b in range(nblocks)]
    for rank, all_len, offs, total, whole in res:
        assert all_len == want
        assert offs == list(np.cumsum([0] + want[:-1]))
        assert total == sum(want)
        assert whole == b"".join(blobs)


def test_single_process_passthrough():
    l = torch.tensor([3, 4, 5], dtype=torch.int32)
    assert gather_lengths(l, 3).tolist() == [3, 4, 5]
    offs, tot = archive_offsets(l)
    assert offs.tolist() == [0, 3, 7] and tot == 12


# ---- the shapes of the real job (BASELINE configs[4]: 131 072 blocks over 8 ranks), an UNEVEN job, and a
# LengthGather on a process group that is not the default one -- eight gloo ranks, lengths only (no compress: this is the exchange step)
def _lens_of(nblocks):
    g = torch.Generator().manual_seed(nblocks)
    return torch.randint(30, 74000, (nblocks,), dtype=torch.int32, generator=g)


def _worker8(rank, world, port, q):
    _rendezvous(rank, world, port, q)
    try:
        res = {}
        for nblocks in (131072, 131075):
            want = _lens_of(nblocks)
            b0, b1 = shard_range(nblocks, rank, world)
            lg = LengthGather(nblocks, "cpu")
            assert lg.equal == (nblocks % world == 0)
            for _ in range(2):                                    # buffers are reused between steps
                got = lg.gather(want[b0:b1].clone())
                assert torch.equal(got, want), (rank, nblocks)
            assert torch.equal(gather_lengths(want[b0:b1].clone(), nblocks), want)
            offs, total = archive_offsets(got)
            res[nblocks] = (int(offs[-1]), total)
        # a sub-group of the odd ranks: group-relative rank and world size decide the shard
        odd = dist.new_group(ranks=[1, 3, 5, 7])
        if rank % 2 == 1:
            nblocks = 1003
            want = _lens_of(nblocks)
            gr, gw = dist.get_rank(odd), dist.get_world_size(odd)
            assert gw == 4 and gr == rank // 2
            b0, b1 = shard_range(nblocks, gr, gw)
            lg = LengthGather(nblocks, "cpu", group=odd)
            assert torch.equal(lg.gather(want[b0:b1].clone()), want)
            blob = torch.arange(b0, b1, dtype=torch.int64).to(torch.uint8)
            whole = gather_archive(blob, blob.numel(), group=odd)
            assert torch.equal(whole, torch.arange(nblocks, dtype=torch.int64).to(torch.uint8))
        # the even ranks are done here while the odd ones still talk inside their group: nobody tears the default group (rank 0: the store)
        # down before everybody is through (a rare hang of this test otherwise: 2 of ~35 runs)
        dist.barrier()
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_length_allgather_world8_job_shapes_and_subgroup():
    res = _spawn_world(_worker8, 8, (), 150)
    for nblocks in (131072, 131075):
        want = _lens_of(nblocks).to(torch.int64)
        for rank, r in res:
            assert r[nblocks] == (int(want[:-1].sum()), int(want.sum())), (rank, nblocks)
