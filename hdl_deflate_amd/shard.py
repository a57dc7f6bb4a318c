"""Multi-GPU block sharding.

Every block is an independent zlib stream (own header, own Adler-32, no cross-block history: the
reference has exactly one stream per START), so the node's GPUs are used as an embarrassingly
parallel shard: rank r owns a contiguous range of blocks, inputs are resident per GPU, and the ONLY
exchange step is an all-gather of the per-block uint32 output lengths (RCCL over xGMI with backend
"nccl"; gloo on CPU in the tests) followed by an exclusive scan that gives every block its offset in
the concatenated archive.  No payload crosses xGMI on the timed path."""
import torch
import torch.distributed as dist


def shard_range(nblocks, rank, world):
    """contiguous range [b0, b1) of rank `rank`; sizes differ by at most one block"""
    base, rem = divmod(nblocks, world)
    b0 = rank * base + min(rank, rem)
    return b0, b0 + base + (1 if rank < rem else 0)


def gather_lengths(local_len, nblocks, group=None):
    """all-gather per-block output lengths of contiguous shards -> int32 [nblocks] on every rank (one-off form; a
    loop should hold a LengthGather, which keeps its buffers)."""
    if not (dist.is_available() and dist.is_initialized()):
        assert local_len.numel() == nblocks
        return local_len.to(torch.int32)
    world = dist.get_world_size(group)          # (a world of ONE rank still goes through the collective, like LengthGather)
    rank = dist.get_rank(group)
    sizes = [shard_range(nblocks, r, world) for r in range(world)]
    maxn = max(b1 - b0 for b0, b1 in sizes)
    b0, b1 = sizes[rank]
    assert local_len.numel() == b1 - b0
    # RCCL ("nccl") gathers device tensors directly; gloo (CPU tests, or two ranks sharing one GPU) needs
    # host tensors -- 4 bytes per block, so the hop is negligible
    dev = local_len.device
    cdev = torch.device("cpu") if dist.get_backend(group) == "gloo" else dev
    pad = torch.zeros(maxn, dtype=torch.int32, device=cdev)
    pad[:b1 - b0] = local_len.to(torch.int32).to(cdev)
    full = torch.empty(world * maxn, dtype=torch.int32, device=cdev)
    dist.all_gather_into_tensor(full, pad, group=group)
    full = full.to(dev)
    parts = [full[r * maxn: r * maxn + (s1 - s0)] for r, (s0, s1) in enumerate(sizes)]
    return torch.cat(parts)


class LengthGather(object):
    """the per-step exchange of the sharded job with everything that does not change between steps hoisted out:
    shard sizes, the padded send buffer and the receive buffer are set up once; equal shards (the BASELINE configs[4]
    job: 131 072 blocks over 1/2/4/8 ranks) are gathered straight from the kernel's out_len tensor into the result --
    no pad, no cat, no allocation, no host sync on the timed path."""

    def __init__(self, nblocks, device, group=None):
        self.nblocks, self.group, self.dev = nblocks, group, torch.device(device)
        self.single = not (dist.is_available() and dist.is_initialized())     # (a world of ONE rank still goes through RCCL)
        if self.single:
            return
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.sizes = [shard_range(nblocks, r, self.world) for r in range(self.world)]
        self.maxn = max(b1 - b0 for b0, b1 in self.sizes)
        self.equal = all(b1 - b0 == self.maxn for b0, b1 in self.sizes)
        self.cdev = torch.device("cpu") if dist.get_backend(group) == "gloo" else self.dev
        self.full = torch.empty(self.world * self.maxn, dtype=torch.int32, device=self.cdev)
        self.pad = None if self.equal else torch.zeros(self.maxn, dtype=torch.int32, device=self.cdev)

    def gather(self, local_len):
        """local_len int32[own shard] -> int32[nblocks] (a view of an internal buffer, overwritten by the next call)"""
        if self.single:
            assert local_len.numel() == self.nblocks
            return local_len
        b0, b1 = self.sizes[self.rank]
        assert local_len.numel() == b1 - b0 and local_len.dtype == torch.int32
        src = local_len if self.cdev == local_len.device else local_len.to(self.cdev)
        if not self.equal:
            self.pad[:b1 - b0] = src
            src = self.pad
        dist.all_gather_into_tensor(self.full, src.contiguous(), group=self.group)
        if self.equal:
            return self.full if self.cdev == self.dev else self.full.to(self.dev)
        parts = [self.full[r * self.maxn: r * self.maxn + (s1 - s0)] for r, (s0, s1) in enumerate(self.sizes)]
        return torch.cat(parts).to(self.dev)


def archive_offsets(all_len):
    """exclusive scan of the gathered lengths -> (int64 offsets [nblocks], total bytes)"""
    l64 = all_len.to(torch.int64)
    incl = torch.cumsum(l64, 0)
    return incl - l64, int(incl[-1].item()) if l64.numel() else 0


def gather_archive(local_archive, nbytes_local, group=None, sizes=None):
    """concatenate the per-rank archives on every rank (payload gather; NOT on the timed path): rank r's bytes are broadcast into
    their place of ONE result buffer -- exact sizes, no padding to the largest archive, no concatenation copy (round 4 padded every
    rank to the maximum and all-gathered: 8 x the largest archive staged per rank).  `sizes` (bytes per rank, e.g. from the gathered
    lengths) spares the size all-gather and its host sync.  Returns uint8 [sum of sizes]."""
    if not (dist.is_available() and dist.is_initialized()):
        return local_archive[:nbytes_local]
    world = dist.get_world_size(group)          # (world == 1 included: the first real multi-GPU run must not be the first RCCL payload gather)
    rank = dist.get_rank(group)
    dev = local_archive.device
    cdev = torch.device("cpu") if dist.get_backend(group) == "gloo" else dev
    if sizes is None:
        t = torch.zeros(world, dtype=torch.int64, device=cdev)
        dist.all_gather_into_tensor(t, torch.tensor([nbytes_local], dtype=torch.int64, device=cdev), group=group)
        sizes = t.tolist()
    sizes = [int(x) for x in sizes]
    assert len(sizes) == world and sizes[rank] == int(nbytes_local)
    out = torch.empty(sum(sizes), dtype=torch.uint8, device=cdev)
    pos = 0
    for r in range(world):
        piece = out[pos:pos + sizes[r]]
        if r == rank:
            piece.copy_(local_archive[:nbytes_local])
        if sizes[r]:
            dist.broadcast(piece, src=dist.get_global_rank(group, r) if group is not None else r, group=group)
        pos += sizes[r]
    return out if cdev == dev else out.to(dev)
