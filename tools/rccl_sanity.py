#!/usr/bin/env python3
"""RCCL sanity on one GPU (world_size 1): the collectives bench.py / shard.py issue at N > 1"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
from hdl_deflate_amd.shard import gather_lengths, archive_offsets
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
ol = torch.arange(1000, dtype=torch.int32, device=dev)
al = gather_lengths(ol, 1000)
assert torch.equal(al, ol)
t = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
print("RCCL ok:", torch.cuda.nccl.version(), float(t))
dist.destroy_process_group()
