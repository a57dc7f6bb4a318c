#!/usr/bin/env python3
"""Reproducer for the hipGraphLaunch abort of round 5: a captured graph that holds hdlz_inflate_batch in its
several-streams whole-GPU form (k_par_*) with POOL scratch (hipMallocFromPoolAsync / hipFreeAsync -> graph memory nodes), replayed.
  HDLZ_LIB=hdl_deflate_amd/lib/libhdlz_poolcap.so graph_abort.py pool REPS   (built with -DHDLZ_ALLOW_POOL_IN_CAPTURE: what round 5 shipped until 00ebd5b)
  graph_abort.py ws REPS                                                     (hdlz_inflate_batch_ws: caller-owned scratch, no memory nodes)
pool: wrong bytes from the 3rd launch on / a hang / an abort (profiles/r06_graph_abort_cause.txt); ws: clean."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import hdl_deflate_amd
from hdl_deflate_amd.data import make_blocks
from hdl_deflate_amd.constants import pitch_for
mode, reps = (sys.argv[1] if len(sys.argv) > 1 else "pool"), int(sys.argv[2]) if len(sys.argv) > 2 else 30
eng = hdl_deflate_amd.Engine()
L, B, n = eng.lib, 512, 2048
d = make_blocks(B, n, "cuda", seed=1)
out = torch.empty((B, pitch_for(n)), dtype=torch.uint8, device="cuda")
eng.compress_batch(d, out=out, out_pitch=out.shape[1])
back = torch.empty((B, n), dtype=torch.uint8, device="cuda")
ol, st = torch.empty(B, dtype=torch.int32, device="cuda"), torch.empty(B, dtype=torch.int32, device="cuda")
work = torch.empty(L.hdlz_inflate_work_bytes(B, out.shape[1], n, 0, 0), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
g, s = torch.cuda.CUDAGraph(), torch.cuda.Stream()
with torch.cuda.stream(s), torch.cuda.graph(g, stream=s):
    a = (out.data_ptr(), None, out.shape[1], out.shape[1], B, 0, 0, back.data_ptr(), n, ol.data_ptr(), st.data_ptr())
    rc = L.hdlz_inflate_batch(*a, s.cuda_stream) if mode == "pool" else L.hdlz_inflate_batch_ws(*a, work.data_ptr(), work.numel(), s.cuda_stream)
    assert rc == 0, L.hdlz_last_error()
for r in range(reps):
    back.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert int((st != 0).sum()) == 0 and torch.equal(back, d), "launch %d of the graph: wrong bytes" % r
print("graph_abort %s: %d launches clean (lib %s)" % (mode, reps, hdl_deflate_amd._lib.LIB_PATH), flush=True)
