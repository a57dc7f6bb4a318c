#!/usr/bin/env python3
"""bench.py -- the hot path on BASELINE.json's metric: MB/s of uncompressed input consumed by
STARTC (CWINDOW=32, MATCH10, static tree), whole job, plus the compression ratio.

Workload at N=1 = BASELINE configs[1]: 2^20 x 2 KiB synthetic blocks per GPU (families 1..4 of
test_deflate.py:38-66, every block distinct, 2 GiB >> 256 MB Infinity Cache), resident in HBM before
the timed region.  A "step" = one hdlz_compress_batch launch over all of the rank's blocks (+ for
N>1 the RCCL all-gather of the per-block output lengths).  Weak scaling: per-GPU work
is fixed, value = bytes all ranks consumed / max-over-ranks time.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- dominant kernel (k_compress): algorithmic bytes per launch (N_in + N_out + 4 per block,
                  SURVEY 8(d)) / average launch duration measured with HIP events on the launch stream,
                  against the 8 TB/s HBM3E peak
  cpu_baseline -- the CPU oracle (a port: oracle/hdlz_oracle.c) timed on this box's host cores on a
                  bounded sample of the same blocks (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def measured_traffic(key):
    """HBM bytes per launch measured with rocprofv3 PMC passes for exactly this kernel/config (profiles/traffic.json)"""
    try:
        with open(os.path.join(REPO, "profiles", "traffic.json")) as f:
            e = json.load(f).get(key)
        return (e["traffic_bytes"], e["source"]) if e else (None, None)
    except Exception:
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=1 << 20, help="blocks per GPU")
    ap.add_argument("--block-size", type=int, default=2048)
    ap.add_argument("--cwindow", type=int, default=32)
    ap.add_argument("--maxmatch", type=int, default=10)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline duration (0 = skip)")
    ap.add_argument("--verify", type=int, default=256, help="blocks checked against zlib outside the timed region")
    ap.add_argument("--data", default="families", choices=["families", "text"],
                    help="families = test_deflate.py families 1-4 (BASELINE configs[1]); text = Zipf pseudo-English "
                         "(enwik8 stand-in for configs[2]: enwik8 cannot be fetched, no network)")
    ap.add_argument("--zlib-strategy", default="fixed", choices=["fixed", "default"],
                    help="inflate mode: fixed = Z_FIXED streams (configs[3]); default = stock zlib streams with "
                         "dynamic trees (exercises the second pass k_inflate_dyn, SURVEY 8(f) rank 1)")
    ap.add_argument("--mode", default="compress", choices=["compress", "inflate"],
                    help="compress = BASELINE metric (default); inflate = configs[3] side metric (1 GPU)")
    a = ap.parse_args()
    if a.mode == "inflate":
        return bench_inflate(a)

    import torch
    import torch.distributed as dist
    import hdl_deflate_amd
    from hdl_deflate_amd.data import make_blocks, make_text_blocks
    from hdl_deflate_amd.shard import gather_lengths
    from hdl_deflate_amd.constants import pitch_for

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == a.gpus, "launch with torchrun --nproc-per-node %d for --gpus %d" % (a.gpus, a.gpus)
    ndev = torch.cuda.device_count()
    backend = os.environ.get("HDLZ_BENCH_BACKEND", "nccl")   # "gloo": functional check of the N>1 flow on fewer GPUs
    if backend == "nccl":
        assert local < ndev, "one GPU per rank is required with RCCL"
    local = min(local, ndev - 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    eng = hdl_deflate_amd.Engine(dev)
    B, n = a.blocks, a.block_size
    nblocks_total = B * world
    # rank r owns the contiguous block range [r*B, (r+1)*B) of the job (weak scaling)
    if a.data == "text":
        d_in = make_text_blocks(B, n, dev, seed=rank)
    else:
        d_in = make_blocks(B, n, dev, seed=0, first_block=rank * B)
    pitch = pitch_for(n)
    d_out = torch.empty((B, pitch), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def step():
        out, ol, st = eng.compress_batch(d_in, cwindow=a.cwindow, maxmatch=a.maxmatch, out=d_out, out_pitch=pitch)
        all_len = gather_lengths(ol, nblocks_total) if world > 1 else ol
        return ol, st, all_len

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()

    # ---- timed region: exactly K steps, barrier + synchronize on both sides
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ol, st, all_len = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    cdev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- kernel-only duration of the dominant kernel: HIP events on the launch stream (= torch's
    # current stream, which is the stream handed to the C-ABI), separate launches
    evs = []
    for _ in range(max(3, a.steps)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.compress_batch(d_in, cwindow=a.cwindow, maxmatch=a.maxmatch, out=d_out, out_pitch=pitch)
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    k_ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    k_avg = sum(k_ms) / len(k_ms)

    # ---- achievable HBM ceiling on this box: a plain device copy of the same input (read + write)
    cp = torch.empty_like(d_in)
    cp.copy_(d_in)
    torch.cuda.synchronize()
    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    c0.record()
    for _ in range(5):
        cp.copy_(d_in)
    c1.record()
    torch.cuda.synchronize()
    copy_gbs = 2.0 * d_in.numel() * 5 / (c0.elapsed_time(c1) * 1e-3) / 1e9
    del cp

    # ---- checks outside the timed region
    bad = int((st != 0).sum().item())
    out_bytes_local = int(ol.to(torch.int64).sum().item())
    in_bytes_local = B * n
    tot = torch.tensor([out_bytes_local, in_bytes_local, bad], dtype=torch.int64, device=cdev)
    if world > 1:
        dist.all_reduce(tot)
        assert int(all_len.to(torch.int64).sum().item()) == int(tot[0].item())
    out_bytes, in_bytes, bad = (int(x) for x in tot.tolist())
    assert bad == 0, "%d blocks failed" % bad
    if rank == 0 and a.verify:
        import zlib
        idx = torch.linspace(0, B - 1, a.verify).long().unique()
        hi = d_in[idx.to(dev)].cpu().numpy()
        ho = d_out[idx.to(dev)].cpu().numpy()
        hl = ol[idx.to(dev)].cpu().numpy()
        for k in range(len(idx)):
            assert zlib.decompress(ho[k, :hl[k]].tobytes()) == hi[k].tobytes(), "zlib round trip failed"

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = dt / a.steps * 1e3
    value = in_bytes / (dt / a.steps) / 1e6                      # MB/s, whole job
    algo_bytes = in_bytes_local + out_bytes_local + 4 * B        # per launch (one GPU)
    achieved = algo_bytes / (k_avg * 1e-3) / 1e9
    kname = "k_compress<%d>" % (1 if a.cwindow <= 32 else 2 if a.cwindow <= 64 else 8)
    traffic, tsrc = measured_traffic("%s|blocks=%d|block=%d|data=%s" % (kname, B, n, a.data))
    res = {
        "metric": "compress_input_throughput (CWINDOW=%d, MATCH10=%s, static tree)" % (a.cwindow, a.maxmatch == 10),
        "value": round(value, 1), "unit": "MB/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": ("BASELINE configs[1]: %d x %d B blocks per GPU, families 1-4 (test_deflate.py:38-66), "
                                "distinct blocks, HBM-resident" % (B, n)) if a.data == "families" else
                               ("%d x %d B blocks per GPU of Zipf pseudo-English (enwik8 stand-in), HBM-resident" % (B, n)),
                   "cwindow": a.cwindow, "maxmatch": a.maxmatch, "blocks_per_gpu": B, "block_bytes": n,
                   "parallelism": "block-shard x%d (length all-gather only)" % world},
        "per_gpu_MBps": round(value / world, 1),
        "compression_ratio_out_over_in": round(out_bytes / in_bytes, 4),
        "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "traffic_source": tsrc, "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms_avg": round(k_avg, 4),
                     "kernel_ms_min": round(k_ms[0], 4), "device_copy_GBps": round(copy_gbs, 1),
                     "frac_of_device_copy": round(achieved / copy_gbs, 4),
                     "note": "bytes = N_in + N_out + 4 per block; this path is VALU-issue bound, not HBM bound "
                             "(DESIGN.md)"},
    }
    if world == 1 and a.cpu_seconds > 0:
        res["cpu_baseline"] = cpu_baseline(d_in, n, a)
    print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


def _zfixed_chunk(args):
    import zlib
    buf, n, strat = args
    out = []
    for k in range(0, len(buf), n):
        co = zlib.compressobj(strategy=zlib.Z_FIXED if strat == "fixed" else zlib.Z_DEFAULT_STRATEGY, wbits=15)
        out.append(co.compress(buf[k:k + n]) + co.flush())
    return out


def bench_inflate(a):
    """BASELINE configs[3]: B stock-zlib Z_FIXED streams (wbits=15) over 2 KiB blocks of families 1/2/4
    (family 3 would make zlib emit stored blocks, which the DYNAMIC=False reference mis-decodes), made on
    the host cores with stock zlib outside the timed region; DYNAMIC=False semantics
    (HDLZ_INFLATE_ASSUME_FIXED); every stream checked against the original block."""
    import multiprocessing as mp
    import numpy as np
    import torch
    import hdl_deflate_amd
    from hdl_deflate_amd.data import make_blocks
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    eng = hdl_deflate_amd.Engine(dev)
    B, n = a.blocks, a.block_size
    d_plain = make_blocks(B, n, dev, seed=4, families=(1, 2, 4))
    host = d_plain.cpu().numpy()
    nproc = min(os.cpu_count() or 1, 64)
    per = (B + nproc - 1) // nproc
    with mp.get_context("fork").Pool(nproc) as pool:
        parts = pool.map(_zfixed_chunk, [(host[k:k + per].tobytes(), n, a.zlib_strategy) for k in range(0, B, per)])
    streams = [z for p in parts for z in p]
    lens = np.fromiter((len(z) for z in streams), dtype=np.int64, count=B)
    off = np.zeros(B + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    flat = np.frombuffer(b"".join(streams) + bytes(64), dtype=np.uint8)
    d_in = torch.from_numpy(flat.copy()).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d_out = torch.empty((B, n), dtype=torch.uint8, device=dev)
    flags = hdl_deflate_amd.INFLATE_ASSUME_FIXED if a.zlib_strategy == "fixed" else 0

    def step():
        return eng.inflate_batch(d_in, in_off=d_off, out_pitch=n, flags=flags, out=d_out)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out, ol, st = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    evs = []
    for _ in range(max(3, a.steps)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        step()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    k_ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    k_avg = sum(k_ms) / len(k_ms)
    assert int((st != 0).sum().item()) == 0 and int((ol != n).sum().item()) == 0
    assert torch.equal(d_out, d_plain), "inflate output differs from the original blocks"
    z_bytes, u_bytes = int(off[-1]), B * n
    algo = z_bytes + u_bytes + 4 * B
    achieved = algo / (k_avg * 1e-3) / 1e9
    traffic, tsrc = measured_traffic("k_inflate|streams=%d|block=%d" % (B, n))
    res = {"metric": "inflate_output_throughput (zlib Z_FIXED streams, DYNAMIC=False)" if a.zlib_strategy == "fixed"
           else "inflate_output_throughput (stock zlib streams, dynamic trees, two passes)",
           "value": round(u_bytes / (dt / a.steps) / 1e6, 1), "unit": "MB/s", "n_gpus": 1, "steps": a.steps,
           "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": "BASELINE configs[3]: %d zlib %s streams over %d B blocks, families 1/2/4, "
                                  "HBM-resident" % (B, "Z_FIXED" if a.zlib_strategy == "fixed" else "default-strategy (dynamic trees)", n),
                      "streams": B, "block_bytes": n},
           "input_MBps": round(z_bytes / (dt / a.steps) / 1e6, 1),
           "compression_ratio_out_over_in": round(z_bytes / u_bytes, 4),
           "roofline": {"bound": "hbm", "kernel": "k_inflate", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                        "traffic_source": tsrc, "algorithmic_bytes_per_launch": algo, "kernel_ms_avg": round(k_avg, 4),
                        "kernel_ms_min": round(k_ms[0], 4)}}
    if a.cpu_seconds > 0:
        from oracle import oracle as O
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        S = min(B, 1 << 19)
        t1 = time.perf_counter()
        _, l2, s2 = O.inflate_batch(flat, off[:S + 1].astype(np.uint64), n, flags=flags, nthreads=cores)
        dtc = time.perf_counter() - t1
        assert (s2 == 0).all()
        res["cpu_baseline"] = {"value": round(S * n / dtc / 1e6, 1), "unit": "MB/s", "cores": cores, "kind": "port",
                               "sample": "first %d streams, oracle/hdlz_oracle.c inflate, %d threads, %.2f s" % (S, cores, dtc)}
    print(json.dumps(res), flush=True)


def cpu_baseline(d_in, n, a):
    """the CPU oracle (kind "port") on a bounded sample of the same blocks, all host cores"""
    import numpy as np
    from oracle import oracle as O          # cpu_baseline leg: the oracle is the thing timed here, as allowed
    O.lib()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    B = d_in.shape[0]
    pitch = O.out_bound(n)

    def run(sample, threads):
        S = sample.shape[0]
        off = (np.arange(S + 1, dtype=np.uint64) * n)
        out = np.ones((S, pitch), np.uint8)               # pre-touched: no first-touch faults in the timed call
        t0 = time.perf_counter()
        _, ol, st = O.compress_batch(sample.reshape(-1), off, a.cwindow, a.maxmatch, nthreads=threads, out=out)
        dt = time.perf_counter() - t0
        assert (st == 0).all()
        return dt

    probe = d_in[:min(B, 4096)].cpu().numpy()
    run(probe, cores)                                      # warm: library load, thread start-up
    rate1 = probe[:512].size / run(probe[:512], 1)         # single-thread bytes/s
    # bounded sample: about cpu_seconds of single-core-equivalent work per core, capped by the data we have
    S = int(min(B, max(4096, rate1 * cores * a.cpu_seconds * 0.5 / n)))
    sample = d_in[:S].cpu().numpy()
    dt = run(sample, cores)
    # familiar yardstick: stock zlib level 1, Z_FIXED, one core, on 16 MiB of the same blocks
    import zlib
    zs = sample[:min(S, (16 << 20) // n)]
    t0 = time.perf_counter()
    for k in range(zs.shape[0]):
        co = zlib.compressobj(1, zlib.DEFLATED, 15, 8, zlib.Z_FIXED)
        co.compress(zs[k].tobytes()); co.flush()
    zl_rate = zs.size / (time.perf_counter() - t0)
    return {"value": round(sample.size / dt / 1e6, 1), "unit": "MB/s", "cores": cores, "kind": "port",
            "stock_zlib_level1_zfixed_single_core_MBps": round(zl_rate / 1e6, 1),
            "sample": "first %d blocks of the same workload (%.1f MiB), oracle/hdlz_oracle.c, %d threads, %.2f s"
                      % (S, sample.size / 2 ** 20, cores, dt),
            "single_thread_MBps": round(rate1 / 1e6, 1),
            "reference_constants": {"fpga_100MHz_3cyc_per_byte_MBps": 33, "standin_sim_KBps": "0.5-1 (BASELINE.md)"}}


if __name__ == "__main__":
    main()
