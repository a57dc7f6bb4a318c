#!/usr/bin/env python3
"""workload for rocprofv3: one 16 MiB and one 256 MiB stream through hdlz_compress_stream, 5 reps each"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hdl_deflate_amd import Engine
from hdl_deflate_amd.data import make_text_blocks
e = Engine()
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 16
d = make_text_blocks(mib, 1 << 20, "cuda", seed=3).reshape(-1)
d = torch.cat([d, torch.zeros(16, dtype=torch.uint8, device="cuda")])
for _ in range(6):
    e.compress_stream(d, mib << 20)
torch.cuda.synchronize()
