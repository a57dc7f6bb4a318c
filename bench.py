#!/usr/bin/env python3
"""bench.py -- the hot path on BASELINE.json's metric: MB/s of uncompressed input consumed by
STARTC (CWINDOW=32, MATCH10, static tree), whole job, plus the compression ratio.

--gpus 1 (default).  Headline = BASELINE configs[1]: 2^20 x 2 KiB synthetic blocks (families 1..4 of
  test_deflate.py:38-66, every block distinct, 2 GiB >> 256 MB Infinity Cache), resident in HBM before the
  timed region; a "step" = one hdlz_compress_batch launch over all blocks.  The same JSON line carries a
  "secondary" array with the other single-GPU configurations of BASELINE.json, each with its own roofline:
    * configs[4] shape on one GPU: the WHOLE 8 GiB job of 131 072 x 64 KiB blocks, CWINDOW=32 (north_star's
      target shape; its ms_per_step is T(1) of the strong-scaling curve below)
    * configs[2]: CWINDOW=64 + MATCH10 on 64 KiB blocks of Zipf pseudo-English (enwik8 is not obtainable:
      no network), next to CWINDOW=32 on the same data (ratio vs throughput)
    * configs[3]: inflate of 2^20 stock-zlib Z_FIXED streams, DYNAMIC=False semantics, every stream checked
--gpus N > 1 (one rank per GPU; launched by torch.distributed.run, or plainly as `python3 bench.py --gpus N`, which starts
  its own N ranks).  BASELINE configs[4] as written: the 8 GiB
  job of 131 072 x 64 KiB blocks is split into contiguous shards of B/N blocks (shard.shard_range); a step =
  the rank's hdlz_compress_batch launch + the RCCL all-gather of the uint32[B/N] output lengths
  (SURVEY 8(e)); STRONG scaling: total work is fixed, value = 8 GiB / max-over-ranks step time.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- dominant kernel: algorithmic bytes per launch (N_in + N_out + 4 per block) /
                  average launch duration measured with HIP events on the launch stream, against the 8 TB/s
                  HBM3E peak
  cpu_baseline -- the CPU oracle (a port: oracle/hdlz_oracle.c) timed on this box's host cores on a bounded
                  sample of the same blocks (rank 0, N=1 only)
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

METRIC_C32 = "compress_input_throughput (CWINDOW=32, MATCH10=True, static tree)"
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
CFG5_BLOCKS, CFG5_BLOCK = 131072, 65536          # BASELINE configs[4]: 8 GiB of 64 KiB blocks


def measured_counters(key, kname=None, grid=None):
    """counters measured with rocprofv3 PMC passes for exactly this kernel / workload (profiles/traffic.json, written by
    tools/update_traffic.py from the summaries under profiles/).  The entry is REFUSED -- (None, reason) -- when it was recorded for
    another kernel symbol, launch grid or library version than the one this run just launched: a stale number must not pass
    as this run's traffic (VERDICT r3 #5c)."""
    try:
        with open(os.path.join(REPO, "profiles", "traffic.json")) as f:
            e = json.load(f).get(key)
    except Exception:
        return None, "profiles/traffic.json unreadable"
    if not e:
        return None, None
    from hdl_deflate_amd import _lib
    ver = _lib.load().hdlz_version()
    norm = lambda x: x.replace(" ", "")
    if e.get("hdlz_version") != ver:
        return None, "STALE: %s was recorded with libhdlz version 0x%06x, this is 0x%06x" % (e.get("source"), e.get("hdlz_version") or 0, ver)
    if kname is not None and not norm(e.get("kernel", "")).startswith(norm(kname).split("+")[0].rstrip(">")):
        return None, "STALE: %s was recorded for kernel %s, this run launched %s" % (e.get("source"), e.get("kernel"), kname)
    if grid is not None and e.get("grid") not in (None, grid):
        return None, "STALE: %s was recorded for a grid of %s threads, this run launched %d" % (e.get("source"), e.get("grid"), grid)
    return e, e.get("source")


def compress_grid(nblocks, ncu=256):
    """threads hdlz_compress_batch launches for the one-block-per-wave kernels (hdlz_compress.hip: launch_compress)"""
    return min(ncu * 256, nblocks) * 64


def kname_for(cwindow, n=1 << 16):
    """the kernel symbol hdlz_compress_batch launches for this window / block size (hdlz_compress.hip: launch_compress)"""
    nch = 1 if cwindow <= 32 else 2 if cwindow <= 64 else 8
    if nch == 1 and 5 <= n <= 1024:
        return "k_compress_small<false, %s>" % ("true" if cwindow == 32 else "false")      # several small blocks per wave-tile
    return "k_compress<%d, %s, %s>" % (nch, "true" if cwindow == 32 * nch else "false", "true" if (nch == 1 and n <= 2048) else "false")


def lengths_digest(torch, lens):
    """position-weighted checksum of a length vector: equal digests <=> (for all practical purposes) the same length at every index"""
    l64 = lens.to(torch.int64).reshape(-1)
    w = torch.arange(1, l64.numel() + 1, dtype=torch.int64, device=l64.device)
    return int(((l64 * (w % 1000003)).sum() % 2305843009213693951).item())


def median(v):
    v = sorted(v)
    return v[len(v) // 2] if len(v) % 2 else 0.5 * (v[len(v) // 2 - 1] + v[len(v) // 2])


def roofline(kname, algo_bytes, k_ms, traffic_key=None, extra=None, grid=None, in_bytes=None):
    """`hbm`: algorithmic bytes / average launch duration against the 8 TB/s peak (SURVEY 8(d) fixes this class for every kernel of
    the path).  `issue`: the resource that actually binds these integer kernels -- VALU issue: wave-instructions per launch
    (SQ_INSTS_VALU) x cycles per instruction / 1024 SIMDs against the kernel's cycles (GRBM_GUI_ACTIVE / 8 XCDs), both from the PMC
    file named in `source`, so that the fraction can be recomputed from the line alone."""
    k_avg = sum(k_ms) / len(k_ms)
    achieved = algo_bytes / (k_avg * 1e-3) / 1e9
    e, tsrc = measured_counters(traffic_key, kname, grid) if traffic_key else (None, None)
    r = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": e["traffic_bytes"] if e else None, "traffic_source": tsrc,
         "algorithmic_bytes_per_launch": algo_bytes, "kernel_ms_avg": round(k_avg, 4), "kernel_ms_median": round(median(k_ms), 4),
         "kernel_ms_min": round(min(k_ms), 4), "launches_timed": len(k_ms)}
    if e and e.get("valu_insts") and e.get("valu_fast_frac") is not None:
        # VALU-pipe occupancy, priced with the measured two-class table (VERDICT r4 #3a): a wave64 instruction holds its SIMD's VALU pipe
        # for 1.96 (add/sub/and/or/xor/lshr/ashr/mov/min_u16/bitop3) or 3.25 shader cycles (everything else), counted in s_memtime ticks
        # (profiles/r04_ubench/ubench_valu_cycles*.txt); the kernel's own mix of the two classes comes from its ISA (tools/valu_mix.py),
        # the launch's cycles in the SAME ticks from its duration x the counter's clock under this load (tools/exp_tile_timing.py)
        ff = e["valu_fast_frac"]
        cpi = 1.96 * ff + 3.25 * (1.0 - ff)
        est = e["valu_insts"] * cpi / 1024.0
        cyc = k_avg * 1e-3 * e.get("shader_clock_ghz", 2.3) * 1e9
        r["issue"] = {"valu_insts_per_launch": int(e["valu_insts"]), "salu_insts_per_launch": int(e.get("salu_insts") or 0),
                      "valu_insts_per_byte": round(e["valu_insts"] / in_bytes, 3) if in_bytes else None,
                      "valu_fast_frac": round(ff, 3), "cycles_per_valu_inst": round(cpi, 3), "est_valu_pipe_cycles": int(est),
                      "kernel_cycles": int(cyc), "shader_clock_ghz": e.get("shader_clock_ghz", 2.3), "frac": round(est / cyc, 3),
                      "source": e.get("source"), "note": e.get("issue_note")}
    if extra:
        r.update(extra)
    return r


def time_steps(torch, dist, step, steps, warmup, world, cdev):
    """W untimed warm-up steps, then EXACTLY K steps bracketed by barrier + synchronize; max over ranks"""
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(steps):
        last = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, last


def kernel_ms(torch, launch, n):
    """duration of single launches: HIP events on the launch stream (torch's current stream = the stream handed to the C-ABI)"""
    evs = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    return sorted(e0.elapsed_time(e1) for e0, e1 in evs)


def zlib_spot_check(torch, d_in, d_out, ol, count):
    import zlib
    B = d_in.shape[0]
    idx = torch.linspace(0, B - 1, min(count, B)).long().unique().to(d_in.device)
    hi, ho, hl = d_in[idx].cpu().numpy(), d_out[idx].cpu().numpy(), ol[idx].cpu().numpy()
    for k in range(len(idx)):
        assert zlib.decompress(ho[k, :hl[k]].tobytes()) == hi[k].tobytes(), "zlib round trip failed"


def run_compress(torch, eng, d_in, cwindow, maxmatch, steps, warmup, verify, d_out=None):
    """single-GPU compress measurement of one workload -> dict (wall step time, kernel durations, sizes)"""
    from hdl_deflate_amd.constants import pitch_for
    B, n = d_in.shape
    pitch = pitch_for(n)
    if d_out is None:
        d_out = torch.empty((B, pitch), dtype=torch.uint8, device=d_in.device)

    def step():
        return eng.compress_batch(d_in, cwindow=cwindow, maxmatch=maxmatch, out=d_out, out_pitch=pitch)

    dt, (out, ol, st) = time_steps(torch, None, step, steps, warmup, 1, None)
    k_ms = kernel_ms(torch, step, max(10, steps))
    assert int((st != 0).sum().item()) == 0, "blocks failed"
    out_bytes = int(ol.to(torch.int64).sum().item())
    if verify:
        zlib_spot_check(torch, d_in, d_out, ol, verify)
    return {"dt": dt, "k_ms": k_ms, "in_bytes": B * n, "out_bytes": out_bytes, "B": B, "n": n, "d_out": d_out, "ol": ol}


def end_to_end(torch, eng, d_in, r, cwindow, maxmatch, reps=5):
    """SURVEY 8(d) "Timing": the same job INCLUDING the PCIe hops -- pinned host input -> HBM, the launch, pitched output rows +
    lengths -> pinned host memory, all on the launch stream; reported beside the metric, never as `value`"""
    from hdl_deflate_amd.constants import pitch_for
    B, n = d_in.shape
    pitch = pitch_for(n)
    d_out, ol = r["d_out"], r["ol"]
    h_in = torch.empty((B, n), dtype=torch.uint8, pin_memory=True)
    h_in.copy_(d_in)
    h_out = torch.empty((B, pitch), dtype=torch.uint8, pin_memory=True)
    h_len = torch.empty(B, dtype=ol.dtype, pin_memory=True)
    d_stage = torch.empty_like(d_in)
    tot, h2d, d2h = [], [], []
    for _ in range(reps + 1):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev[0].record()
        d_stage.copy_(h_in, non_blocking=True)
        ev[1].record()
        _, ol2, _ = eng.compress_batch(d_stage, cwindow=cwindow, maxmatch=maxmatch, out=d_out, out_pitch=pitch)
        ev[2].record()
        h_out.copy_(d_out, non_blocking=True)
        h_len.copy_(ol2, non_blocking=True)
        ev[3].record()
        torch.cuda.synchronize()
        tot.append((time.perf_counter() - t0) * 1e3)
        h2d.append(ev[0].elapsed_time(ev[1]))
        d2h.append(ev[2].elapsed_time(ev[3]))
    tot, h2d, d2h = tot[1:], h2d[1:], d2h[1:]                      # the first pass warms the pinned pages
    assert int(h_len.to(torch.int64).sum().item()) == r["out_bytes"]
    # the same job through Engine.compress_host: chunks on three streams (H2D of chunk k + 1 beside the kernels of chunk k beside the
    # D2H of chunk k - 1), and what comes back is the ARCHIVE (streams back to back + lengths), not the pitched rows
    import zlib
    h_arch = h_out.view(-1)
    pipe = []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, _, total, bad = eng.compress_host(h_in, cwindow=cwindow, maxmatch=maxmatch, h_archive=h_arch, h_len=h_len)
        torch.cuda.synchronize()
        pipe.append((time.perf_counter() - t0) * 1e3)
    pipe = pipe[1:]
    eng.release_host_buffers()                                      # (the staging buffers of compress_host: not needed by the entries that follow)
    assert bad == 0 and total == r["out_bytes"] and int(h_len.to(torch.int64).sum().item()) == total
    hl = h_len.numpy().astype("int64")
    hoff = hl.cumsum() - hl
    for b in (0, 1, B // 2, B - 1):
        assert zlib.decompress(h_arch[int(hoff[b]):int(hoff[b] + hl[b])].numpy().tobytes()) == h_in[b].numpy().tobytes(), "host archive"
    return {"ms_median": round(median(tot), 3), "ms_min": round(min(tot), 3), "input_MBps": round(B * n / median(tot) / 1e3, 1),
            "h2d_ms": round(median(h2d), 3), "h2d_GBps": round(B * n / median(h2d) / 1e6, 1),
            "d2h_ms": round(median(d2h), 3), "d2h_GBps": round(B * pitch / median(d2h) / 1e6, 1),
            "d2h_bytes": B * pitch + 4 * B, "reps": reps,
            "pipelined": {"ms_median": round(median(pipe), 3), "ms_min": round(min(pipe), 3), "ms_all": [round(x, 2) for x in pipe], "input_MBps": round(B * n / median(pipe) / 1e3, 1),
                          "d2h_bytes": r["out_bytes"] + 4 * B,
                          "note": "Engine.compress_host: the batch in chunks on three streams, compress + scan + hdlz_compact_batch per chunk, "
                                  "the archive (not the pitched rows) copied back; zlib round trip of blocks of the host archive checked"},
            "note": "pinned host buffers; H2D of the input, one hdlz_compress_batch launch, D2H of the pitched rows (out_pitch = %d) and "
                    "the lengths; PCIe-bound, not the metric" % pitch}


def compress_entry(name, workload, r, cwindow, maxmatch, steps, warmup, traffic_key=None):
    algo = r["in_bytes"] + r["out_bytes"] + 4 * r["B"]
    return {"name": name, "metric": "compress_input_throughput (CWINDOW=%d, MATCH10=%s, static tree)" % (cwindow, maxmatch == 10),
            "value": round(r["in_bytes"] / (r["dt"] / steps) / 1e6, 1), "unit": "MB/s", "steps": steps, "warmup": warmup,
            "ms_per_step": round(r["dt"] / steps * 1e3, 4), "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload, "cwindow": cwindow, "maxmatch": maxmatch, "blocks": r["B"], "block_bytes": r["n"]},
            "compression_ratio_out_over_in": round(r["out_bytes"] / r["in_bytes"], 4),
            "roofline": roofline(kname_for(cwindow, r["n"]), algo, r["k_ms"], traffic_key, None, compress_grid(r["B"]), r["in_bytes"])}


# ------------------------------------------------------------------------------------------------ the ONE line
LINE_MAX = 12288            # the driver keeps only the tail of stdout: a line beyond ~16 KB was not parsed (BENCH_r05.parsed == None)
DETAIL_DEFAULT = os.path.join("profiles", "r06_bench_detail.json")
ROOF_KEEP = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic_bytes_per_launch",
             "kernel_ms_avg", "kernel_ms_median", "kernel_ms_min", "launches_timed", "device_copy_GBps", "frac_of_device_copy")
ISSUE_KEEP = ("frac", "valu_insts_per_launch", "valu_insts_per_byte", "source")
TOP_KEEP = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "per_gpu_MBps", "compression_ratio_out_over_in", "roofline", "cpu_baseline", "archive", "end_to_end",
            "input_MBps", "length_allgather_ms_avg", "lengths_digest", "T1_ms", "T1_kernel_ms_median", "speedup_vs_T1",
            "T1_lengths_digest", "scaling_curve", "error", "visible_devices", "secondary", "detail", "name", "inflate_MBps", "compress_MBps",
            "inflate_ms", "compress_ms", "inflate_ms_with_hint", "ms_with_hint", "one_wave_ms", "wave_per_stream_ms", "compress_roofline")
SEC_KEEP = ("name", "metric", "value", "unit", "ms_per_step", "compression_ratio_out_over_in", "input_MBps", "roofline",
            "compress_roofline", "inflate_MBps", "compress_MBps", "wave_per_stream_ms", "one_wave_ms", "inflate_ms_with_hint", "ms_with_hint", "status")


def _numbers_only(d, depth=2):
    """a sub-object of the line without its prose: numbers, booleans, None and strings of at most 48 characters"""
    out = {}
    for k, v in d.items():
        if k == "note" or k.endswith("_note") or k == "ms_all":
            continue
        if isinstance(v, dict):
            if depth > 0:
                out[k] = _numbers_only(v, depth - 1)
        elif not isinstance(v, str) or len(v) <= 48:
            out[k] = v
    return out


def _slim_roofline(r):
    s = {k: r[k] for k in ROOF_KEEP if k in r}
    if isinstance(r.get("issue"), dict):
        s["issue"] = {k: r["issue"][k] for k in ISSUE_KEEP if k in r["issue"]}
    return s


def _short(s, n):
    return s if len(s) <= n else s[:n - 3] + "..."


def slim_line(res, detail_path=None, line_max=LINE_MAX):
    """the line the driver parses: the contract's keys, `roofline`, `cpu_baseline` and the numbers of every secondary entry; all
    prose (`note`s, long workload texts, per-entry configs) stays in the detail file named in `detail`"""
    top = {k: res[k] for k in TOP_KEEP if k in res and k not in ("secondary", "roofline", "compress_roofline", "cpu_baseline", "archive", "end_to_end", "config")}
    if isinstance(res.get("compress_roofline"), dict):
        top["compress_roofline"] = _slim_roofline(res["compress_roofline"])
    if "config" in res:
        top["config"] = {k: (_short(v, 200) if isinstance(v, str) else v) for k, v in res["config"].items()}
    if "roofline" in res:
        top["roofline"] = _slim_roofline(res["roofline"])
    if "cpu_baseline" in res:
        cb = res["cpu_baseline"]
        top["cpu_baseline"] = {k: (_short(v, 160) if isinstance(v, str) else v) for k, v in cb.items() if k != "note"}
    for k in ("archive", "end_to_end"):
        if isinstance(res.get(k), dict):
            top[k] = _numbers_only(res[k])
    if detail_path:
        top["detail"] = detail_path
    sec = []
    for e in res.get("secondary", []):
        se = {k: e[k] for k in SEC_KEEP if k in e and k not in ("roofline", "compress_roofline")}
        if "metric" in se:
            se["metric"] = _short(se["metric"], 80)
        for k in ("roofline", "compress_roofline"):
            if isinstance(e.get(k), dict):
                se[k] = _slim_roofline(e[k])
                se[k]["kernel"] = _short(str(se[k].get("kernel")), 48)
                for drop in ("peak", "unit", "bound", "kernel_ms_median", "kernel_ms_min", "launches_timed"):
                    se[k].pop(drop, None)
        sec.append(se)
    if sec:
        top["secondary"] = sec
    enc = lambda o: json.dumps(o, separators=(",", ":"))
    line = enc(top)
    # never lose the headline to the size of the rest: shed secondary detail in steps until the line fits
    for shed in ("traffic_source", "metric", "compress_roofline", "issue", "roofline"):
        if len(line) < line_max:
            break
        for se in top.get("secondary", []):
            se.pop(shed, None)
            for k in ("roofline", "compress_roofline"):
                if isinstance(se.get(k), dict):
                    se[k].pop(shed, None)
        line = enc(top)
    while len(line) >= line_max and top.get("secondary"):
        top["secondary"].pop()
        top["secondary_truncated"] = True
        line = enc(top)
    assert len(line) < line_max, "bench line of %d bytes" % len(line)
    return line


def emit(res, a=None):
    """write the full result (every note, config and sub-measurement) to the detail file, print the slim line"""
    path = getattr(a, "detail", None) if a is not None else None
    named = None
    if path:
        for p in (path, os.path.join("gpurun_out", os.path.basename(path)) if os.path.isdir(os.path.join(REPO, "gpurun_out")) else None):
            if not p:
                continue
            try:
                full = p if os.path.isabs(p) else os.path.join(REPO, p)
                os.makedirs(os.path.dirname(full), exist_ok=True)
                with open(full, "w") as f:
                    json.dump(res, f, indent=1)
                    f.write("\n")
                named = named or p
            except OSError:
                pass
    print(slim_line(res, named), flush=True)


# ------------------------------------------------------------------------------------------------ N = 1
def main_single(a):
    import torch
    import hdl_deflate_amd
    from hdl_deflate_amd.data import make_blocks, make_text_blocks
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    eng = hdl_deflate_amd.Engine(dev)
    B, n = a.blocks, a.block_size
    if a.data == "text":
        d_in = make_text_blocks(B, n, dev, seed=0)
    else:
        d_in = make_blocks(B, n, dev, seed=0)
    if a.shuffle:
        # (VERDICT r5 #6b) the same blocks in a seeded random ORDER: the families no longer alternate with period 4, so a wave's blocks
        # -- it strides over the batch -- are a mix of families instead of one
        g = torch.Generator(device="cpu")
        g.manual_seed(a.shuffle)
        perm = torch.randperm(B, generator=g).to(dev)
        d_in = d_in[perm].contiguous()
        del perm
    torch.cuda.synchronize()
    r = run_compress(torch, eng, d_in, a.cwindow, a.maxmatch, a.steps, a.warmup, a.verify)

    # achievable HBM ceiling on this box: a plain device copy of the same input (read + write)
    cp = torch.empty_like(d_in)
    cp.copy_(d_in)
    torch.cuda.synchronize()
    c_ms = kernel_ms(torch, lambda: cp.copy_(d_in), 5)
    copy_gbs = 2.0 * d_in.numel() / (sum(c_ms) / len(c_ms) * 1e-3) / 1e9
    del cp

    value = r["in_bytes"] / (r["dt"] / a.steps) / 1e6
    algo = r["in_bytes"] + r["out_bytes"] + 4 * B
    kname = kname_for(a.cwindow, n)
    rl = roofline(kname, algo, r["k_ms"], "k_compress<%d>|blocks=%d|block=%d|data=%s" % (1 if a.cwindow <= 32 else 2 if a.cwindow <= 64 else 8, B, n, a.data),
                  {"device_copy_GBps": round(copy_gbs, 1),
                   "note": "bytes = N_in + N_out + 4 per block; this path is VALU-issue bound, not HBM bound: see `issue`; kernel_ms_* are "
                           "HIP events around %d separate launches AFTER the timed loop of ms_per_step (two loops: they differ by noise)"
                           % len(r["k_ms"])}, compress_grid(B) if n > 1024 else None, r["in_bytes"])
    rl["frac_of_device_copy"] = round(rl["achieved"] / copy_gbs, 4)
    res = {
        "metric": "compress_input_throughput (CWINDOW=%d, MATCH10=%s, static tree)" % (a.cwindow, a.maxmatch == 10),
        "value": round(value, 1), "unit": "MB/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": round(r["dt"] / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": ("BASELINE configs[1]: %d x %d B blocks, families 1-4 (test_deflate.py:38-66), "
                                "distinct blocks, HBM-resident%s" % (B, n, ", block order shuffled (seed %d)" % a.shuffle if a.shuffle else "")) if a.data == "families" else
                               ("%d x %d B blocks of Zipf pseudo-English (enwik8 stand-in), HBM-resident" % (B, n)),
                   "cwindow": a.cwindow, "maxmatch": a.maxmatch, "blocks_per_gpu": B, "block_bytes": n,
                   "parallelism": "single GPU; --gpus N block-shards BASELINE configs[4] (8 GiB of 64 KiB blocks) over N ranks"},
        "per_gpu_MBps": round(value, 1),
        "compression_ratio_out_over_in": round(r["out_bytes"] / r["in_bytes"], 4),
        "roofline": rl,
    }
    # SURVEY 8(f) rank 2 beside the pitched-row figure: the same job delivered as ONE contiguous archive + offset index
    # (exclusive scan of the lengths + hdlz_compact_batch behind the compress launch; a fused compress -> archive would save the
    # compaction's read + write of N_out, at most the compact_ms below)
    if a.archive:
        offs = torch.empty(B, dtype=torch.int64, device=dev)
        arch = torch.empty(r["out_bytes"], dtype=torch.uint8, device=dev)

        def to_archive():
            l64 = r["ol"].to(torch.int64)
            torch.cumsum(l64, 0, out=offs)
            offs.sub_(l64)
            eng.compact(r["d_out"], r["ol"], offsets=offs, archive=arch)

        def both():
            eng.compress_batch(d_in, cwindow=a.cwindow, maxmatch=a.maxmatch, out=r["d_out"], out_pitch=r["d_out"].shape[1])
            to_archive()

        offs1 = torch.empty(B + 1, dtype=torch.int64, device=dev)

        def to_archive1():                     # round 5: scan + gather in one launch (hdlz_archive_batch)
            eng.archive(r["d_out"], r["ol"], archive=arch, offsets=offs1)

        def both1():
            eng.compress_batch(d_in, cwindow=a.cwindow, maxmatch=a.maxmatch, out=r["d_out"], out_pitch=r["d_out"].shape[1])
            to_archive1()

        to_archive()
        ref_arch = arch.clone()
        c_ms = kernel_ms(torch, to_archive, 10)
        b_ms = kernel_ms(torch, both, 10)
        arch.zero_()
        to_archive1()
        assert os.environ.get("HDLZ_BENCH_NOCHECK") or (torch.equal(arch, ref_arch) and torch.equal(offs1[:-1], offs) and
                                                        int(offs1[-1].item()) == r["out_bytes"]), "hdlz_archive_batch differs from scan + compact"
        del ref_arch
        c1_ms = kernel_ms(torch, to_archive1, 10)
        b1_ms = kernel_ms(torch, both1, 10)
        res["archive"] = {"archive_ms": round(median(c1_ms), 4), "compress_archive_ms": round(median(b1_ms), 4),
                          "input_MBps": round(r["in_bytes"] / median(b1_ms) / 1e3, 1), "archive_bytes": r["out_bytes"],
                          "two_pass": {"scan_plus_compact_ms": round(median(c_ms), 4), "compress_scan_compact_ms": round(median(b_ms), 4),
                                       "input_MBps": round(r["in_bytes"] / median(b_ms) / 1e3, 1)},
                          "note": "the job as one contiguous archive + int64 offset index: compress into pitched rows, then hdlz_archive_batch "
                                  "(scan of the lengths by a decoupled look-back + the gather, one launch; HIP events around the two launches); "
                                  "two_pass: round 4's form (three scan launches + hdlz_compact_batch); both archives compared byte for byte"}
    if a.end_to_end:
        res["end_to_end"] = end_to_end(torch, eng, d_in, r, a.cwindow, a.maxmatch)
    if a.cpu_seconds > 0:
        res["cpu_baseline"] = cpu_baseline(d_in, n, a)
    del r, d_in
    torch.cuda.empty_cache()

    if a.secondary:
        sec = []
        # -- configs[4] shape, whole job on one GPU (= T(1) of the strong-scaling curve of --gpus N)
        d5 = make_blocks(a.cfg5_blocks, CFG5_BLOCK, dev, seed=0)
        r5 = run_compress(torch, eng, d5, 32, 10, a.steps, a.warmup, 16)
        sec.append(compress_entry("configs[4]-shape, 1 GPU", "BASELINE configs[4] on ONE GPU: %d x 64 KiB blocks (%.1f GiB), families 1-4, "
                                  "CWINDOW=32 -- north_star's target block shape; ms_per_step = T(1) of `bench.py --gpus N`"
                                  % (a.cfg5_blocks, a.cfg5_blocks * CFG5_BLOCK / 2 ** 30), r5, 32, 10, a.steps, a.warmup,
                                  "k_compress<1>|blocks=%d|block=65536|data=families" % a.cfg5_blocks))
        sec.append(bench_roundtrip(torch, eng, a, d5, r5))
        d_out5 = r5["d_out"]
        del r5, d5
        # -- configs[2]: CWINDOW=64 + MATCH10 on 64 KiB text blocks, next to CWINDOW=32 on the same data
        dt_ = make_text_blocks(a.text_blocks, CFG5_BLOCK, dev, seed=0)
        d_out_t = d_out5[:a.text_blocks]
        r64 = run_compress(torch, eng, dt_, 64, 10, a.steps, a.warmup, 16, d_out=d_out_t)
        e64 = compress_entry("configs[2]", "BASELINE configs[2]: CWINDOW=64 + MATCH10 on %d x 64 KiB blocks of Zipf pseudo-English "
                             "(enwik8 stand-in: enwik8 cannot be fetched, no network)" % a.text_blocks, r64, 64, 10, a.steps, a.warmup,
                             "k_compress<2>|blocks=%d|block=65536|data=text" % a.text_blocks)
        # CWINDOW=32 on the same data, for the ratio side of the trade: 1024 of the blocks through the multi-wave stream passes
        # (hdlz_compress_streams: bit-identical output, other kernel symbols -- the rocprof rows of this command stay one
        # workload per kernel); its THROUGHPUT is the configs[4]-shape entry's: k_compress<1> is data-independent within 1 %
        nb32 = min(a.text_blocks, 1024)
        o32, l32, s32 = eng.compress_batch(dt_[:nb32], cwindow=32, maxmatch=10)
        assert int((s32 != 0).sum().item()) == 0
        e64["same_data_cwindow32"] = {"compression_ratio_out_over_in": round(int(l32.to(torch.int64).sum().item()) / (nb32 * CFG5_BLOCK), 4),
                                      "sample_blocks": nb32, "throughput": "see the configs[4]-shape entry (same kernel, data-independent)"}
        del o32, l32, s32
        sec.append(e64)
        # -- the reference's own non-FAST window (CWINDOW = 256, deflate.py:58-59) on the same text blocks: the window-independent finder
        r256 = run_compress(torch, eng, dt_, 256, 10, a.steps, a.warmup, 16, d_out=d_out_t)
        sec.append(compress_entry("cwindow=256", "CWINDOW=256 + MATCH10 (the reference's non-FAST build, deflate.py:58-59) on the configs[2] blocks: "
                                  "%d x 64 KiB of Zipf pseudo-English" % a.text_blocks, r256, 256, 10, a.steps, a.warmup,
                                  "k_compress<8>|blocks=%d|block=65536|data=text" % a.text_blocks))
        del r64, r256, dt_, d_out_t, d_out5
        torch.cuda.empty_cache()
        # -- configs[3]: inflate
        sec.append(bench_inflate(a, eng, cpu=False))
        # -- streams with dynamic-tree blocks (stock zlib, default strategy): pass 1 flags them, k_inflate_tok<true>
        #    decodes them one lane each; the same streams one wave each beside it
        sec.append(bench_inflate(a, eng, cpu=False, streams=min(a.streams, 1 << 18), strategy="default"))
        # -- round 5: the same configs[3] streams through the 16-lanes-per-stream mapping (history in LDS: traffic ~1.0x algorithmic; the
        #    default for batches of HDLZ_INFLATE_GROUP_MIN .. _MAX streams), a quarter of them -- the measured answer to VERDICT r4 #1
        sec.append(bench_inflate(a, eng, cpu=False, streams=min(a.streams, 1 << 18), kernel="group", name="configs[3], 16 lanes per stream"))
        # -- the reference's own use: ONE stream at a time.  STARTC then STARTD on one 16 MiB stream (the most a port with LMAX = 24
        #    holds), each on the whole GPU (k_stream_*, k_par_*)
        sec.append(bench_single_stream(torch, eng, dev, a))
        sec.append(bench_zlib_stream(torch, eng, dev, a))
        sec.append(bench_few_large(torch, eng, dev, a))
        res["secondary"] = sec
    emit(res, a)


def bench_roundtrip(torch, eng, a, d_plain, r):
    """the round trip of the configs[4] job (the reference's harness always inflates what it compressed,
    test_deflate.py:197-286): the streams STARTC just wrote -- compacted into ONE archive, so the inflate reads ragged streams
    through in_off, the form a stored archive has -- through hdlz_inflate_batch, every stream compared with its block"""
    B, n = d_plain.shape
    dev = d_plain.device
    l64 = r["ol"].to(torch.int64)
    offs = torch.zeros(B + 1, dtype=torch.int64, device=dev)
    torch.cumsum(l64, 0, out=offs[1:])
    total = int(offs[-1].item())
    arch = torch.zeros(total + 64, dtype=torch.uint8, device=dev)
    eng.compact(r["d_out"], r["ol"], offsets=offs[:B], archive=arch)
    back = torch.empty((B, n), dtype=torch.uint8, device=dev)

    def step():
        return eng.inflate_batch(arch, in_off=offs, out_pitch=n, out=back)

    dt, (_, bl, bs) = time_steps(torch, None, step, a.steps, a.warmup, 1, None)
    k_ms = kernel_ms(torch, step, max(3, a.steps))
    assert int((bs != 0).sum().item()) == 0 and int((bl != n).sum().item()) == 0, "round trip: streams failed"
    assert torch.equal(back, d_plain), "round trip: inflated bytes differ from the blocks that were compressed"
    algo = total + B * n + 4 * B
    return {"name": "configs[4] round trip", "metric": "inflate_output_throughput (the streams STARTC wrote for the configs[4] job, one archive)",
            "value": round(B * n / (dt / a.steps) / 1e6, 1), "unit": "MB/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "STARTD of the %d streams the 'configs[4]-shape' entry wrote (CWINDOW=32 streams of 64 KiB blocks, compacted "
                                   "into one %.2f GB archive, ragged in_off), every stream compared with its block" % (B, total / 1e9),
                       "streams": B, "block_bytes": n, "archive_bytes": total},
            "input_MBps": round(total / (dt / a.steps) / 1e6, 1),
            "roofline": roofline("k_inflate_tok<false>", algo, k_ms, "k_inflate_tok|roundtrip|streams=%d|block=%d" % (B, n), None,
                                 (B + 255) // 256 * 256, B * n)}


def bench_single_stream(torch, eng, dev, a, n=1 << 24):
    from hdl_deflate_amd.data import make_blocks
    d = make_blocks(n // 2048, 2048, dev, seed=5).reshape(-1)
    out, ol, st = eng.compress_stream(d, n)
    zn = int(ol.item())
    assert int(st.item()) == 0
    zin = torch.cat([out[:zn], torch.zeros(64, dtype=torch.uint8, device=dev)]).reshape(1, -1)
    back = torch.empty((1, n), dtype=torch.uint8, device=dev)

    def step_c():
        return eng.compress_stream(d, n, out=out)

    def step_d():
        return eng.inflate_batch(zin, in_len=zn, out_pitch=n, out=back)

    def step_dh():                           # with the caller's hint "a single fixed block" (what STARTC writes): only that chain is launched
        return eng.inflate_batch(zin, in_len=zn, out_pitch=n, out=back, flags=128)

    _, bl, bs = step_d()
    torch.cuda.synchronize()
    assert int(bs.item()) == 0 and int(bl.item()) == n and torch.equal(back.reshape(-1), d[:n]), "single-stream round trip failed"
    kc = kernel_ms(torch, step_c, max(3, a.steps))
    kd = kernel_ms(torch, step_d, max(3, a.steps))
    kh = kernel_ms(torch, step_dh, max(3, a.steps))
    ms_c, ms_d = sum(kc) / len(kc), sum(kd) / len(kd)
    algo = n + zn + 4                         # one stream = N_in + N_out + one length word, in either direction
    return {"name": "one 16 MiB stream", "metric": "single-stream throughput (STARTC then STARTD of ONE stream, whole GPU each)",
            "value": round(n / ms_d / 1e3, 1), "unit": "MB/s", "ms_per_step": round(ms_d, 4), "higher_is_better": True,
            "config": {"workload": "one stream of %d bytes (families 1-4), CWINDOW=32, MATCH10: compressed by hdlz_compress_stream, "
                                   "inflated by hdlz_inflate_batch(nstreams = 1), round trip checked" % n,
                       "stream_bytes": n, "compressed_bytes": zn},
            "inflate_MBps": round(n / ms_d / 1e3, 1), "inflate_ms": round(ms_d, 4),
            "inflate_ms_with_hint": round(sum(kh) / len(kh), 4),       # HDLZ_INFLATE_ONE_FIXED_BLOCK: the chain for other block types is not launched beside it
            "compress_MBps": round(n / ms_c / 1e3, 1), "compress_ms": round(ms_c, 4),
            # (VERDICT r4 #3b) the path is a CHAIN of kernels: algorithmic bytes over the duration of the whole call (HIP events around it);
            # traffic = the counters summed over all kernels of one call (tools/prof_single.py -> profiles/r05_single_stream_pmc_summary.txt)
            "roofline": roofline("k_par_* (STARTD: all kernels of hdlz_inflate_batch(nstreams = 1))", algo, kd, "k_par|stream=%d" % n),
            "compress_roofline": roofline("k_stream_* (STARTC: all kernels of hdlz_compress_stream)", algo, kc, "k_stream|stream=%d" % n),
            "note": "one wave (every single stream before hdlz_inflate_par.hip): 9 MB/s; timed with HIP events around whole calls "
                    "(all kernels of the path)"}


def bench_zlib_stream(torch, eng, dev, a, n=1 << 24, level=6, with_wave=True):
    """STARTD of ONE stock-zlib stream (level 6: dynamic-tree blocks, 32 KiB distances) -- what the reference's default build
    (DYNAMIC=True, deflate.py:32) is fed -- on the whole GPU (hdlz_inflate_any.hip, round 6; one wave up to round 5: 11 MB/s)"""
    import zlib
    from hdl_deflate_amd.data import make_blocks
    plain = make_blocks(n // 2048, 2048, "cpu", seed=7).numpy().tobytes()
    z = zlib.compress(plain, level)
    zn = len(z)
    zin = torch.frombuffer(bytearray(z + bytes(64)), dtype=torch.uint8).to(dev).reshape(1, -1)
    want = torch.frombuffer(bytearray(plain), dtype=torch.uint8).to(dev)
    back = torch.empty((1, n), dtype=torch.uint8, device=dev)
    work = torch.empty(eng.lib.hdlz_inflate_work_bytes(1, zn, n, 0, 0), dtype=torch.uint8, device=dev)

    def step():
        return eng.inflate_batch(zin, in_len=zn, out_pitch=n, out=back, work=work)

    def step_wave():
        return eng.inflate_batch(zin, in_len=zn, out_pitch=n, out=back, flags=4)

    _, bl, bs = step()
    torch.cuda.synchronize()
    assert int(bs.item()) == 0 and int(bl.item()) == n and torch.equal(back.reshape(-1), want), "zlib stream: inflated bytes differ"
    kd = kernel_ms(torch, step, max(3, a.steps))
    kw = kernel_ms(torch, step_wave, 1) if with_wave else [0.0]      # (profiling runs leave it out: its kernel would be summed into the path's family)
    ms_d = sum(kd) / len(kd)
    return {"name": "one 16 MiB zlib level-%d stream" % level, "metric": "single-stream inflate throughput (stock zlib stream, any block types, whole GPU)",
            "value": round(n / ms_d / 1e3, 1), "unit": "MB/s", "ms_per_step": round(ms_d, 4), "higher_is_better": True,
            "config": {"workload": "zlib.compress(level %d) of %d bytes (families 1-4): dynamic-tree blocks, distances up to 32 KiB; inflated by "
                                   "hdlz_inflate_batch_ws(nstreams = 1), bytes compared with the input" % (level, n),
                       "stream_bytes": n, "compressed_bytes": zn, "scratch_bytes": int(work.numel())},
            "input_MBps": round(zn / ms_d / 1e3, 1), "one_wave_ms": round(kw[0], 2),
            "roofline": roofline("k_any_* + k_par_emit/jump (STARTD: all kernels of hdlz_inflate_batch(nstreams = 1))", n + zn + 4, kd,
                                 "k_any|stream=%d|level=%d" % (n, level)),
            "note": "HIP events around whole calls (all kernels of the path); one_wave_ms: the same stream through k_inflate_dyn (flag 4), "
                    "what every such stream cost up to round 5"}


def bench_few_large(torch, eng, dev, a, nstreams=256, n=1 << 20, with_wave=True):
    """a FEW LARGE streams in one hdlz_inflate_batch call (fixed pitch): the whole-GPU path over all of them (k_par_*, blockIdx.y = the
    stream) -- the batch kernels decode a stream as one serial chain, so such a batch ran at one stream's latency (flag 4: the wave mapping)"""
    from hdl_deflate_amd.data import make_blocks
    d = make_blocks(nstreams * (n // 2048), 2048, dev, seed=6).reshape(nstreams, n)
    zo, zl, st = eng.compress_batch(d, cwindow=32, maxmatch=10)
    assert int(st.max().item()) == 0
    zsum = int(zl.sum().item())
    back = torch.empty((nstreams, n), dtype=torch.uint8, device=dev)

    def step():
        return eng.inflate_batch(zo, out_pitch=n, out=back)

    def step_wave():
        return eng.inflate_batch(zo, out_pitch=n, out=back, flags=4)

    def step_hint():
        return eng.inflate_batch(zo, out_pitch=n, out=back, flags=128)

    _, bl, bs = step()
    torch.cuda.synchronize()
    assert int(bs.max().item()) == 0 and int(bl.min().item()) == n and torch.equal(back, d), "few-large-streams round trip failed"
    kd = kernel_ms(torch, step, max(3, a.steps))
    kw = kernel_ms(torch, step_wave, 2) if with_wave else [0.0]      # (profiling runs leave it out: its kernel is part of the path's family)
    kh = kernel_ms(torch, step_hint, max(3, a.steps)) if with_wave else [0.0]
    ms_d = sum(kd) / len(kd)
    total = nstreams * n
    algo = total + zsum + 4 * nstreams
    return {"name": "%d streams of 1 MiB" % nstreams, "metric": "inflate_output_throughput (a few large streams, one call)",
            "value": round(total / ms_d / 1e3, 1), "unit": "MB/s", "ms_per_step": round(ms_d, 4), "higher_is_better": True,
            "config": {"workload": "%d own streams of %d bytes (families 1-4, CWINDOW=32, MATCH10) in rows of one pitch: hdlz_inflate_batch, "
                                   "no mapping hint, round trip checked" % (nstreams, n), "streams": nstreams, "stream_bytes": n,
                       "compressed_bytes": zsum},
            "wave_per_stream_ms": round(sum(kw) / len(kw), 3), "ms_with_hint": round(sum(kh) / len(kh), 4),
            "roofline": roofline("k_par_* (STARTD: all kernels of hdlz_inflate_batch, %d streams)" % nstreams, algo, kd,
                                 "k_par|streams=%d|stream=%d" % (nstreams, n)),
            "note": "every kernel of the single-stream path launched once for all streams (blockIdx.y = the stream); timed with HIP events "
                    "around whole calls; wave_per_stream_ms: the same batch through k_inflate_dyn (flag 4), what it cost before round 5"}


# ------------------------------------------------------------------------------------------------ N > 1
def main_sharded(a):
    """BASELINE configs[4]: strong scaling of the 8 GiB job over N ranks, RCCL all-gather of the lengths"""
    import torch
    import torch.distributed as dist
    import hdl_deflate_amd
    from hdl_deflate_amd.data import make_blocks
    from hdl_deflate_amd.shard import LengthGather, shard_range
    from hdl_deflate_amd.constants import pitch_for
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:                                     # a launcher with another world size than --gpus: say so, do not hang
        if rank == 0:
            print(json.dumps({"metric": METRIC_C32, "value": None, "unit": "MB/s", "n_gpus": a.gpus,
                              "error": "WORLD_SIZE=%d but --gpus %d" % (world, a.gpus)}), flush=True)
        sys.exit(2)
    ndev = torch.cuda.device_count()
    backend = os.environ.get("HDLZ_BENCH_BACKEND", "nccl")   # "gloo": functional check of the N>1 flow on fewer GPUs
    if backend == "nccl":
        assert local < ndev, "one GPU per rank is required with RCCL"
    local = min(local, ndev - 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import datetime
    tmo = datetime.timedelta(minutes=30)                     # (ranks > 0 wait at a barrier while rank 0 runs T(1) of the whole job)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
    cdev = dev if backend == "nccl" else torch.device("cpu")

    eng = hdl_deflate_amd.Engine(dev)
    total, n = a.cfg5_blocks, CFG5_BLOCK
    pitch = pitch_for(n)
    b0, b1 = shard_range(total, rank, world)               # contiguous shard of the job
    B = b1 - b0
    d_in = make_blocks(B, n, dev, seed=0, first_block=b0)  # exactly the blocks the 1-GPU run holds at [b0, b1) (any world size)
    d_out = torch.empty((B, pitch), dtype=torch.uint8, device=dev)
    lg = LengthGather(total, dev)
    torch.cuda.synchronize()

    def step():
        out, ol, st = eng.compress_batch(d_in, cwindow=32, maxmatch=10, out=d_out, out_pitch=pitch)
        return ol, st, lg.gather(ol)                       # the ONLY exchange step: uint32[B/N] lengths

    dt, (ol, st, all_len) = time_steps(torch, dist, step, a.steps, a.warmup, world, cdev)
    k_ms = kernel_ms(torch, lambda: eng.compress_batch(d_in, cwindow=32, maxmatch=10, out=d_out, out_pitch=pitch), max(10, a.steps))
    g_ms = kernel_ms(torch, lambda: lg.gather(ol), max(10, a.steps)) if backend == "nccl" else [0.0]
    # T(1): the WHOLE job on rank 0's GPU (the other ranks wait at the barrier), so that the line carries its own single-GPU
    # reference -- same blocks, same kernel, same process; the driver computes efficiency from its own N = 1 run.  AFTER the timed
    # shards: an 18 GiB allocation that came and went before them was seen to cost a later launch up to 10 % (placement of the buffers)
    t1 = None
    if a.t1:
        if rank == 0:
            d1 = make_blocks(total, n, dev, seed=0)
            r1 = run_compress(torch, eng, d1, 32, 10, a.steps, a.warmup, 0)
            t1 = {"ms": r1["dt"] / a.steps * 1e3, "out_bytes": r1["out_bytes"], "k_ms": r1["k_ms"], "digest": lengths_digest(torch, r1["ol"])}
            del r1, d1
            torch.cuda.empty_cache()
        dist.barrier()

    tot = torch.tensor([int(ol.to(torch.int64).sum().item()), B * n, int((st != 0).sum().item())], dtype=torch.int64, device=cdev)
    dist.all_reduce(tot)
    out_bytes, in_bytes, bad = (int(x) for x in tot.tolist())
    assert bad == 0, "%d blocks failed" % bad
    assert all_len.numel() == total and int(all_len.to(torch.int64).sum().item()) == out_bytes, "gathered lengths disagree"
    assert torch.equal(all_len[b0:b1].to(ol.device), ol.to(torch.int32)), "own shard not at its place in the gathered lengths"
    digest = lengths_digest(torch, all_len)
    if t1 is not None:                                     # the shards together ARE the single-GPU job: same bytes in, same bytes out,
        assert out_bytes == t1["out_bytes"], "sharded job wrote %d bytes, the 1-GPU job %d" % (out_bytes, t1["out_bytes"])
        assert digest == t1["digest"], "the gathered lengths are not the 1-GPU job's lengths"     # ... the same length at every index
    if rank == 0 and a.verify:
        zlib_spot_check(torch, d_in, d_out, ol, min(a.verify, 32))
    if rank == 0:
        algo = B * n + int(ol.to(torch.int64).sum().item()) + 4 * B
        value = in_bytes / (dt / a.steps) / 1e6
        res = {"metric": "compress_input_throughput (CWINDOW=32, MATCH10=True, static tree)", "value": round(value, 1),
               "unit": "MB/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": "BASELINE configs[4]: %d x 64 KiB blocks (%.1f GiB total, fixed), families 1-4, contiguous "
                                      "shards of %d blocks per rank, HBM-resident" % (total, total * n / 2 ** 30, B),
                          "cwindow": 32, "maxmatch": 10, "blocks_total": total, "blocks_per_gpu": B, "block_bytes": n,
                          "parallelism": "block-shard x%d; one all-gather of uint32[%d] lengths per step over %s (no payload "
                                         "crosses xGMI)" % (world, B, "RCCL" if backend == "nccl" else backend)},
               "per_gpu_MBps": round(value / world, 1),
               "compression_ratio_out_over_in": round(out_bytes / in_bytes, 4),
               "length_allgather_ms_avg": round(sum(g_ms) / len(g_ms), 4), "lengths_digest": digest,
               "roofline": roofline(kname_for(32, n), algo, k_ms, None, {"note": "rank 0's shard; per-GPU figure"})}
        if t1 is not None:
            res["T1_ms"] = round(t1["ms"], 4)
            res["T1_kernel_ms_median"] = round(median(t1["k_ms"]), 4)
            res["speedup_vs_T1"] = round(t1["ms"] / (dt / a.steps * 1e3), 3)
            res["T1_lengths_digest"] = t1["digest"]
            res["note"] = ("T1_ms = the whole job (%d blocks) on rank 0's GPU alone, measured in this run AFTER the timed shards (same blocks: "
                           "output byte counts and the position-weighted digest of the gathered lengths asserted equal to the 1-GPU "
                           "job's); speedup_vs_T1 / n_gpus is the strong-scaling efficiency" % total)
        else:
            res["note"] = "T(1) of this job is the ms_per_step of the 'configs[4]-shape, 1 GPU' entry of `bench.py --gpus 1`"
        emit(res, a)
    dist.barrier()                 # (rank 0 measured T1 meanwhile: nobody leaves the group before everybody is through)
    dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ inflate
def _zfixed_chunk(args):
    import zlib
    buf, n, strat = args
    out = []
    for k in range(0, len(buf), n):
        co = zlib.compressobj(strategy=zlib.Z_FIXED if strat == "fixed" else zlib.Z_DEFAULT_STRATEGY, wbits=15)
        out.append(co.compress(buf[k:k + n]) + co.flush())
    return b"".join(out), [len(z) for z in out]


def bench_inflate(a, eng=None, cpu=True, streams=None, strategy=None, kernel=None, name=None):
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
    if eng is None:
        eng = hdl_deflate_amd.Engine(dev)
    if streams is not None or strategy is not None or kernel is not None:     # a secondary entry: same flow, other streams / mapping
        import copy
        a = copy.copy(a)
        a.streams = streams or a.streams
        a.zlib_strategy = strategy or a.zlib_strategy
        a.inflate_kernel = kernel or a.inflate_kernel
        a.end_to_end = a.end_to_end and kernel is None
    B, n = a.streams, a.stream_block
    d_plain = make_blocks(B, n, dev, seed=4, families=(1, 2, 4))
    host = d_plain.cpu().numpy()
    nproc = min(os.cpu_count() or 1, 64)
    per = (B + nproc * 4 - 1) // (nproc * 4)
    # (close + join, not the context manager: its terminate() sends SIGTERM, which a profiler's signal handler in the forked workers can
    #  swallow -- a traced run of this file hung there for good)
    pool = mp.get_context("fork").Pool(nproc)
    try:
        parts = pool.map(_zfixed_chunk, [(host[k:k + per].tobytes(), n, a.zlib_strategy) for k in range(0, B, per)])
    finally:
        pool.close()
        pool.join()
    del host
    lens = np.fromiter((l for _, ls in parts for l in ls), dtype=np.int64, count=B)
    off = np.zeros(B + 1, np.int64)
    np.cumsum(lens, out=off[1:])
    flat = np.frombuffer(b"".join(p for p, _ in parts) + bytes(64), dtype=np.uint8)
    del parts
    d_in = torch.from_numpy(flat.copy()).to(dev)
    d_off = torch.from_numpy(off).to(dev)
    d_out = torch.empty((B, n), dtype=torch.uint8, device=dev)
    flags = hdl_deflate_amd.INFLATE_ASSUME_FIXED if a.zlib_strategy == "fixed" else 0
    flags |= {"default": 0, "lane": 2, "wave": 4, "group": 64}[a.inflate_kernel]       # mapping hint (include/hdlz.h); group = 16 lanes per stream

    def step():
        return eng.inflate_batch(d_in, in_off=d_off, out_pitch=n, flags=flags, out=d_out)

    dt, (out, ol, st) = time_steps(torch, None, step, a.steps, a.warmup, 1, None)
    k_ms = kernel_ms(torch, step, max(3, a.steps))
    assert int((st != 0).sum().item()) == 0 and int((ol != n).sum().item()) == 0
    assert os.environ.get("HDLZ_BENCH_NOCHECK") or torch.equal(d_out, d_plain), "inflate output differs from the original blocks"   # (NOCHECK: timing experiments with deliberately broken builds)
    z_bytes, u_bytes = int(off[-1]), B * n
    algo = z_bytes + u_bytes + 4 * B
    fixed = a.zlib_strategy == "fixed"
    res = {"name": name or ("configs[3]" if fixed else "dynamic trees"),
           "metric": "inflate_output_throughput (zlib Z_FIXED streams, DYNAMIC=False)" if fixed
           else "inflate_output_throughput (stock zlib streams, dynamic trees, two passes)",
           "value": round(u_bytes / (dt / a.steps) / 1e6, 1), "unit": "MB/s", "n_gpus": 1, "steps": a.steps,
           "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
           "config": {"workload": "BASELINE configs[3]: %d zlib %s streams over %d B blocks, families 1/2/4, every stream "
                                  "compared with its original block, HBM-resident"
                                  % (B, "Z_FIXED" if fixed else "default-strategy (dynamic trees)", n),
                      "streams": B, "block_bytes": n},
           "input_MBps": round(z_bytes / (dt / a.steps) / 1e6, 1),
           "compression_ratio_out_over_in": round(z_bytes / u_bytes, 4),
           "roofline": roofline(("k_inflate_dyn" if a.inflate_kernel == "wave" else "k_inflate_grp" if a.inflate_kernel == "group" else "k_inflate_tok<false>") +
                                ("" if fixed or a.inflate_kernel == "wave" else " + k_inflate_tok<true>"), algo, k_ms,
                                "%s|streams=%d|block=%d|%s" % ("k_inflate_dyn" if a.inflate_kernel == "wave" else "k_inflate_grp" if a.inflate_kernel == "group" else "k_inflate_tok", B, n, a.zlib_strategy),
                                None, None, u_bytes)}
    if fixed and a.end_to_end:
        # SURVEY 8(d) "Timing" for STARTD: the job from pinned HOST buffers (streams in, rows out), the three steps one after the other;
        # PCIe-bound, never `value`
        h_z = torch.empty(flat.size, dtype=torch.uint8, pin_memory=True)
        h_z.copy_(torch.from_numpy(flat.copy()))
        h_rows = torch.empty((B, n), dtype=torch.uint8, pin_memory=True)
        h_l = torch.empty(B, dtype=torch.int32, pin_memory=True)
        h_s = torch.empty(B, dtype=torch.int32, pin_memory=True)
        d_stage = torch.empty_like(d_in)
        seq = []
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            d_stage.copy_(h_z, non_blocking=True)
            _, ol2, st2 = eng.inflate_batch(d_stage, in_off=d_off, out_pitch=n, flags=flags, out=d_out)
            h_rows.copy_(d_out, non_blocking=True)
            h_l.copy_(ol2, non_blocking=True)
            h_s.copy_(st2, non_blocking=True)
            torch.cuda.synchronize()
            seq.append((time.perf_counter() - t0) * 1e3)
        assert os.environ.get("HDLZ_BENCH_NOCHECK") or (int((h_s != 0).sum().item()) == 0 and int((h_l != n).sum().item()) == 0 and
                                                        torch.equal(h_rows, d_plain.cpu())), "host rows differ"
        seq = seq[1:]
        # the same job through Engine.inflate_host: chunks on three streams, the rows written into the pinned host rows by a kernel
        h_rows.zero_()
        pipe = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.inflate_host(h_z, torch.from_numpy(off), n, flags=flags, h_out=h_rows, h_len=h_l, h_status=h_s)
            torch.cuda.synchronize()
            pipe.append((time.perf_counter() - t0) * 1e3)
        pipe = pipe[1:]
        assert os.environ.get("HDLZ_BENCH_NOCHECK") or (int((h_s != 0).sum().item()) == 0 and int((h_l != n).sum().item()) == 0 and
                                                        torch.equal(h_rows, d_plain.cpu())), "host rows differ (inflate_host)"
        eng.release_host_buffers()
        res["end_to_end"] = {"ms_median": round(median(seq), 3), "output_MBps": round(u_bytes / median(seq) / 1e3, 1),
                             "d2h_floor_ms": round(u_bytes / 57.0e6, 2),
                             "pipelined": {"ms_median": round(median(pipe), 3), "ms_min": round(min(pipe), 3), "ms_all": [round(x, 2) for x in pipe],
                                           "output_MBps": round(u_bytes / median(pipe) / 1e3, 1),
                                           "note": "Engine.inflate_host: chunks on three streams (H2D | hdlz_inflate_batch | rows -> pinned host rows by "
                                                   "hdlz_compact_batch), no host synchronisation inside the job; every row compared with its block"},
                             "note": "pinned host buffers: H2D of the streams + offsets, hdlz_inflate_batch, D2H of the rows, lengths and statuses, one after the other on the launch stream; "
                                     "every row compared with its original block; d2h_floor_ms = the rows alone at the 57 GB/s this link delivers"}
        del h_z, h_rows, d_stage
    if not fixed:
        # the same streams one WAVE each (k_inflate_dyn: what small batches and sessions run), half of them: the mapping's own figure
        Bw = max(1, B // 2)
        wflags = hdl_deflate_amd.INFLATE_WAVE_PER_STREAM

        def step_w():
            return eng.inflate_batch(d_in, in_off=d_off[:Bw + 1], out_pitch=n, flags=wflags, out=d_out[:Bw])

        d_out.zero_()
        _, olw, stw = step_w()
        assert int((stw != 0).sum().item()) == 0 and torch.equal(d_out[:Bw], d_plain[:Bw]), "wave-per-stream inflate differs"
        w_ms = kernel_ms(torch, step_w, max(3, a.steps))
        res["wave_per_stream"] = {"streams": Bw, "MBps": round(Bw * n / (sum(w_ms) / len(w_ms)) / 1e3, 1),
                                  "ms": round(sum(w_ms) / len(w_ms), 4), "kernel": "k_inflate_dyn<false>"}
    if cpu and a.cpu_seconds > 0:
        from oracle import oracle as O
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        S = min(B, 1 << 19)
        t1 = time.perf_counter()
        _, l2, s2 = O.inflate_batch(flat, off[:S + 1].astype(np.uint64), n, flags=flags, nthreads=cores)
        dtc = time.perf_counter() - t1
        assert (s2 == 0).all()
        res["cpu_baseline"] = {"value": round(S * n / dtc / 1e6, 1), "unit": "MB/s", "cores": cores, "kind": "port",
                               "sample": "first %d streams, oracle/hdlz_oracle.c inflate, %d threads, %.2f s" % (S, cores, dtc)}
    return res


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
            "note": "a stated baseline, not a target: the GPU/CPU ratio says nothing about kernel quality (roofline.frac does)",
            "reference_constants": {"fpga_100MHz_3cyc_per_byte_MBps": 33, "standin_sim_KBps": "0.5-1 (BASELINE.md)"}}


def spawn_ranks(a):
    """`python3 bench.py --gpus N` started WITHOUT a launcher (the form the driver uses for N = 1): start the N ranks here, one
    per visible device, through torch.distributed.run on 127.0.0.1; rank 0 prints the ONE JSON line.  With fewer than N
    visible devices a JSON line says so and the exit code is non-zero (HDLZ_BENCH_BACKEND=gloo: functional check, ranks share
    the devices that are there)."""
    import socket
    import subprocess
    import torch
    ndev = torch.cuda.device_count()
    backend = os.environ.get("HDLZ_BENCH_BACKEND", "nccl")
    if ndev < 1 or (backend == "nccl" and ndev < a.gpus):
        print(json.dumps({"metric": METRIC_C32, "value": None, "unit": "MB/s", "n_gpus": a.gpus, "visible_devices": ndev,
                          "error": "--gpus %d needs %d visible GPUs (one rank per GPU over RCCL); %d visible -- not run"
                                   % (a.gpus, a.gpus, ndev)}), flush=True)
        return 3
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env, cwd=REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--blocks", type=int, default=1 << 20, help="headline: blocks (N=1)")
    ap.add_argument("--block-size", type=int, default=2048)
    ap.add_argument("--cwindow", type=int, default=32)
    ap.add_argument("--maxmatch", type=int, default=10)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline duration (0 = skip)")
    ap.add_argument("--verify", type=int, default=256, help="blocks checked against zlib outside the timed region")
    ap.add_argument("--data", default="families", choices=["families", "text"],
                    help="families = test_deflate.py families 1-4 (BASELINE configs[1]); text = Zipf pseudo-English")
    ap.add_argument("--no-secondary", dest="secondary", action="store_false",
                    help="N=1: headline only (skip the configs[4]-shape / configs[2] / configs[3] entries)")
    ap.add_argument("--cfg5-blocks", type=int, default=CFG5_BLOCKS, help="total 64 KiB blocks of the configs[4] job")
    ap.add_argument("--text-blocks", type=int, default=16384, help="64 KiB text blocks of the configs[2] entry (1 GiB)")
    ap.add_argument("--streams", type=int, default=1 << 20, help="inflate: zlib streams (configs[3])")
    ap.add_argument("--stream-block", type=int, default=2048, help="inflate: plain bytes per stream")
    ap.add_argument("--zlib-strategy", default="fixed", choices=["fixed", "default"],
                    help="inflate: fixed = Z_FIXED streams (configs[3]); default = stock zlib streams with dynamic trees "
                         "(exercises the second pass k_inflate_tok<true> / k_inflate_dyn, SURVEY 8(f) rank 1)")
    ap.add_argument("--inflate-kernel", default="default", choices=["default", "lane", "wave", "group"],
                    help="inflate: mapping hint (lane = k_inflate_tok, wave = k_inflate_dyn, group = k_inflate_grp)")
    ap.add_argument("--mode", default="compress", choices=["compress", "inflate", "roundtrip", "single", "few", "zlib"],
                    help="compress = BASELINE metric (default); inflate = only the configs[3] side metric (1 GPU)")
    ap.add_argument("--no-archive", dest="archive", action="store_false",
                    help="N=1: skip the archive figure (compress + scan + hdlz_compact_batch) of the headline job")
    ap.add_argument("--no-end-to-end", dest="end_to_end", action="store_false",
                    help="N=1: skip the PCIe-inclusive measurement of the headline job (SURVEY 8(d) Timing)")
    ap.add_argument("--no-t1", dest="t1", action="store_false",
                    help="N>1: do not run the whole job on rank 0 first (T1_ms / speedup_vs_T1 are then absent)")
    ap.add_argument("--shuffle", type=int, default=0, help="N=1 headline: the blocks in a seeded random order (0 = the order of BASELINE configs[1])")
    ap.add_argument("--detail", default=DETAIL_DEFAULT,
                    help="file that receives the FULL result (notes, configs, every sub-measurement); the printed line names it ('' = none)")
    a = ap.parse_args()
    if a.mode == "inflate":
        emit(bench_inflate(a), a)
    elif a.mode == "roundtrip":                               # only the configs[4] round-trip entry (profiling)
        import torch
        import hdl_deflate_amd
        from hdl_deflate_amd.data import make_blocks
        torch.cuda.set_device(0)
        eng = hdl_deflate_amd.Engine(torch.device("cuda", 0))
        d5 = make_blocks(a.cfg5_blocks, CFG5_BLOCK, torch.device("cuda", 0), seed=0)
        r5 = run_compress(torch, eng, d5, 32, 10, 1, 1, 0)
        emit(bench_roundtrip(torch, eng, a, d5, r5), a)
    elif a.mode == "single":                                  # only the one-stream entry (profiling)
        import torch
        import hdl_deflate_amd
        torch.cuda.set_device(0)
        emit(bench_single_stream(torch, hdl_deflate_amd.Engine(torch.device("cuda", 0)), torch.device("cuda", 0), a), a)
    elif a.mode == "zlib":                                    # only the stock-zlib single-stream entry (profiling)
        import torch
        import hdl_deflate_amd
        torch.cuda.set_device(0)
        emit(bench_zlib_stream(torch, hdl_deflate_amd.Engine(torch.device("cuda", 0)), torch.device("cuda", 0), a, with_wave=False), a)
    elif a.mode == "few":                                     # only the few-large-streams entry (profiling)
        import torch
        import hdl_deflate_amd
        torch.cuda.set_device(0)
        emit(bench_few_large(torch, hdl_deflate_amd.Engine(torch.device("cuda", 0)), torch.device("cuda", 0), a, with_wave=False), a)
    elif a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))
    elif a.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1:
        main_sharded(a)
    else:
        main_single(a)


if __name__ == "__main__":
    main()
