#!/usr/bin/env python3
"""profiles/traffic.json from the PMC summaries of tools/evidence_r6.sh: per bench entry the HBM bytes per launch (FETCH_SIZE x 2
per the gfx950 correction of MI355X_MICROARCH.md + WRITE_SIZE, KB -> bytes; per-dispatch means of separate --pmc passes) and the
issue counters (SQ_INSTS_VALU / _SALU, GRBM_GUI_ACTIVE), together with WHAT was measured -- kernel symbol, launch grid, library
version -- so that bench.py can refuse an entry that does not describe the kernel it just launched.
usage: tools/update_traffic.py gpurun_out/ev_r5        (copies the summaries to profiles/r05_<name>_pmc_summary.txt)
Round 5: the issue side is priced per kernel -- the share of its VALU instructions that issue at the full rate comes
from its ISA (tools/valu_mix.py), the clock of the cycle counter from the tile timing build of the same evidence run."""
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from valu_mix import mix
root = sys.argv[1]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hdr = open(os.path.join(REPO, "include", "hdlz.h")).read()
VERSION = int(re.search(r"#define\s+HDLZ_VERSION\s+(0x[0-9a-fA-F]+)", hdr).group(1), 16)
RND = "r06"
# summary file, traffic.json key, kernel symbol (prefix) the counters are taken from, committed copy
# summary file, traffic.json key, kernel symbol(s) the counters are taken from, committed copy, (source file, mangled substring) for the VALU mix
SPEC = [("pmc_cfg1.txt", "k_compress<1>|blocks=1048576|block=2048|data=families", "k_compress<1, true, true>", "cfg1", ("hdlz_compress.hip", "k_compressILi1ELb1ELb1E")),
        ("pmc_cfg5.txt", "k_compress<1>|blocks=131072|block=65536|data=families", "k_compress<1, true, false>", "cfg5", ("hdlz_compress.hip", "k_compressILi1ELb1ELb0E")),
        ("pmc_cfg2.txt", "k_compress<2>|blocks=16384|block=65536|data=text", "k_compress<2, true, false>", "cfg2", ("hdlz_compress.hip", "k_compressILi2ELb1ELb0E")),
        ("pmc_cw256.txt", "k_compress<8>|blocks=16384|block=65536|data=text", "k_compress<8, true, false>", "cw256", ("hdlz_compress.hip", "k_compressILi8ELb1ELb0E")),
        ("pmc_inflate.txt", "k_inflate_tok|streams=1048576|block=2048|fixed", "k_inflate_tok<false, 288u>", "inflate_tok", ("hdlz_inflate_tok.hip", "k_inflate_tokILb0E")),
        ("pmc_inflate_dyn.txt", "k_inflate_tok|streams=262144|block=2048|default", "k_inflate_tok<false, 288u> + k_inflate_tok<true, 144u>", "inflate_tokdyn", ("hdlz_inflate_tok.hip", "k_inflate_tokILb1ELj144E")),
        ("pmc_roundtrip.txt", "k_inflate_tok|roundtrip|streams=131072|block=65536", "k_inflate_tok<false, 288u>", "roundtrip", ("hdlz_inflate_tok.hip", "k_inflate_tokILb0E")),
        ("pmc_inflate_grp.txt", "k_inflate_grp|streams=262144|block=2048|fixed", "k_inflate_grp", "inflate_grp", ("hdlz_inflate_grp.hip", "k_inflate_grp"))]
# the clock of s_memtime under the headline load (tools/exp_tile_timing.py prints it)
CLK = 2.3
tt = os.path.join(root, "tile_timing.txt")
if os.path.exists(tt):
    m_ = re.search(r"the counter runs at ([0-9.]+) GHz", open(tt).read())
    if m_:
        CLK = float(m_.group(1))
    open(os.path.join(REPO, "profiles", "r06_tile_timing.txt"), "w").write(
        "# tools/exp_tile_timing.py on the -DHDLZ_TILE_TIMING build, evidence run of the shipped kernels (tools/evidence_r6.sh)\n" +
        "".join(l for l in open(tt) if "amdgpu.ids" not in l) +
        "# \"first prologue of the wave\": ~45 us in front of a wave's first tile in THIS instrumented build of <1,true,true> (its prologue spills SGPRs to scratch; the shipped kernel has\n"
        "# no scratch, and the wide-window instantiations of the same build show 140 cycles there): an artifact of the stamps, reported apart so that \"block prologue\" is the per-tile cost.\n")
tj = os.path.join(REPO, "profiles", "traffic.json")
T = json.load(open(tj))
for fn, key, kern, tag, mixsrc in SPEC:
    p = os.path.join(root, fn)
    if not os.path.exists(p):
        continue
    txt = "\n".join(l for l in open(p).read().splitlines() if "amdgpu.ids" not in l) + "\n"
    norm = lambda x: x.replace(" ", "")
    grid = [None]

    def val(name):
        # the counter of the named kernel(s) -- "a + b": the sum over both -- (the summaries list every kernel of the command per PMC pass)
        tot = None
        for kn in kern.split(" + "):
            cur = ""
            for ln in txt.splitlines():
                if ln.lstrip().startswith("kernel "):
                    cur = ln
                m = re.search(r"%s\s+n=\d+ mean=([0-9.e+]+)" % name, ln)
                if m and norm(kn) in norm(cur):
                    g = re.search(r"grid/wg=\('[^']*', '[^']*', '[^']*', '[^']*', '[^']*', '(\d+)', '(\d+)'\)", cur)
                    if g and grid[0] is None:
                        grid[0] = int(g.group(1))
                    tot = (tot or 0.0) + float(m.group(1))
                    break
        return tot
    f, w = val("FETCH_SIZE"), val("WRITE_SIZE")
    if f is None or w is None:
        print("no counters for", kern, "in", p)
        continue
    dest = "profiles/%s_%s_pmc_summary.txt" % (RND, tag)
    open(os.path.join(REPO, dest), "w").write(
        "# rocprofv3 evidence (tools/evidence_r6.sh -> tools/profile*.sh: kernel stats + SQ / LDS / FETCH / WRITE (+ TA / TCP / TCC) passes, "
        "separate --pmc passes, per-dispatch means; libhdlz 0x%06x)\n" % VERSION + txt)
    e = {"traffic_bytes": int(f * 2 * 1024 + w * 1024), "source": dest, "fetch_size_kb": f, "write_size_kb": w,
         "kernel": kern, "grid": grid[0], "hdlz_version": VERSION,
         "valu_insts": val("SQ_INSTS_VALU"), "salu_insts": val("SQ_INSTS_SALU"), "gui_active": val("GRBM_GUI_ACTIVE"),
         "valu_fast_frac": None, "shader_clock_ghz": CLK,
         "issue_note": "est_valu_pipe_cycles = SQ_INSTS_VALU x (1.96 f + 3.25 (1 - f)) / 1024 SIMDs, f = the full-rate share of this kernel's VALU "
                       "instructions in its ISA (tools/valu_mix.py; 1.96 / 3.25 shader cycles per wave64 instruction measured in s_memtime ticks, "
                       "profiles/r04_ubench/ubench_valu_cycles*.txt): the time the VALU pipes are HELD, a lower bound of the time they are needed "
                       "(at 2 waves per SIMD a wave cannot issue faster than 2.4 / 4.4, alone 4.9); kernel_cycles = launch duration x the clock of "
                       "the same counter under the headline load (%.3f GHz, profiles/r06_tile_timing.txt)" % CLK,
         "note": "round 6, %s: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE" % kern}
    f_, s_ = mix(mixsrc[0], mixsrc[1])
    e["valu_fast_frac"] = f_ / float(f_ + s_)
    rd, wr = val("TCC_EA0_RDREQ"), val("TCC_EA0_WRREQ")
    if rd is not None:
        e["tcc_ea0_rdreq"] = rd; e["tcc_ea0_wrreq"] = wr
        e["note"] += "; TCC_EA0_RDREQ / WRREQ = the 64-byte requests that left the L2 (the far history of the copies: one sector per token)"
    T[key] = e
    print(key, e["traffic_bytes"], "grid", e["grid"], "valu", e["valu_insts"])
# a few large streams in one call (PROF_MODE=few tools/prof_single.sh: the k_par_* family summed over the kernels of one call)
p = os.path.join(root, "pmc_few.txt")
if os.path.exists(p):
    txt = "\n".join(l for l in open(p).read().splitlines() if "amdgpu.ids" not in l) + "\n"
    dest = "profiles/%s_few_large_pmc_summary.txt" % RND
    open(os.path.join(REPO, dest), "w").write("# rocprofv3 evidence (PROF_MODE=few tools/prof_single.sh): 256 own streams of 1 MiB through ONE hdlz_inflate_batch call (k_par_*, blockIdx.y = the stream, "
                                              "+ the launch that redoes flagged streams), every counter SUMMED over the kernels of one call; the k_stream family here is the setup's compress_batch; libhdlz 0x%06x\n" % VERSION + txt)
    vals, cur = {}, None
    for ln in txt.splitlines():
        m = re.match(r"\s*family (\w+)\s*$", ln)
        if m:
            cur = m.group(1)
        m = re.search(r"(\w+)\s+n=\d+ mean=([0-9.e+]+)", ln)
        if m and cur == "k_par":
            vals[m.group(1)] = float(m.group(2))
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        key = "k_par|streams=256|stream=1048576"
        T[key] = {"traffic_bytes": int(vals["FETCH_SIZE"] * 2 * 1024 + vals["WRITE_SIZE"] * 1024), "source": dest,
                  "fetch_size_kb": vals["FETCH_SIZE"], "write_size_kb": vals["WRITE_SIZE"],
                  "kernel": "k_par_* (STARTD: all kernels of hdlz_inflate_batch, 256 streams)", "grid": None,
                  "hdlz_version": VERSION, "valu_insts": vals.get("SQ_INSTS_VALU"), "salu_insts": vals.get("SQ_INSTS_SALU"),
                  "note": "round 6: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, summed over the kernels of one call"}
        print(key, T[key]["traffic_bytes"])
# the one-stream paths (tools/prof_single.sh: counters summed over the kernels of one call)
p = os.path.join(root, "pmc_single.txt")
if os.path.exists(p):
    txt = "\n".join(l for l in open(p).read().splitlines() if "amdgpu.ids" not in l) + "\n"
    dest = "profiles/%s_single_stream_pmc_summary.txt" % RND
    open(os.path.join(REPO, dest), "w").write("# rocprofv3 evidence (tools/evidence_r6.sh -> tools/prof_single.sh): one 16 MiB stream through hdlz_compress_stream (k_stream_*) and "
                                              "hdlz_inflate_batch(nstreams = 1) (k_par_* + the fall-back launch), every counter SUMMED over the kernels of one call; libhdlz 0x%06x\n" % VERSION + txt)
    for fam, key, kname in (("k_stream", "k_stream|stream=16777216", "k_stream_* (STARTC: all kernels of hdlz_compress_stream)"),
                            ("k_par", "k_par|stream=16777216", "k_par_* (STARTD: all kernels of hdlz_inflate_batch(nstreams = 1))")):
        vals = {}
        cur = None
        for ln in txt.splitlines():
            m = re.match(r"\s*family (\w+)\s*$", ln)
            if m:
                cur = m.group(1)
            m = re.search(r"(\w+)\s+n=\d+ mean=([0-9.e+]+)", ln)
            if m and cur == fam:
                vals[m.group(1)] = float(m.group(2))
        if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
            T[key] = {"traffic_bytes": int(vals["FETCH_SIZE"] * 2 * 1024 + vals["WRITE_SIZE"] * 1024), "source": dest,
                      "fetch_size_kb": vals["FETCH_SIZE"], "write_size_kb": vals["WRITE_SIZE"], "kernel": kname, "grid": None,
                      "hdlz_version": VERSION, "valu_insts": vals.get("SQ_INSTS_VALU"), "salu_insts": vals.get("SQ_INSTS_SALU"),
                      "note": "round 6: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, summed over the kernels of one call"}
            print(key, T[key]["traffic_bytes"])
# one stock-zlib stream (PROF_MODE=zlib tools/prof_single.sh: k_any_* + the k_par_* kernels of the same call, summed)
p = os.path.join(root, "pmc_zlib.txt")
if os.path.exists(p):
    txt = "\n".join(l for l in open(p).read().splitlines() if "amdgpu.ids" not in l) + "\n"
    dest = "profiles/%s_zlib_stream_pmc_summary.txt" % RND
    open(os.path.join(REPO, dest), "w").write("# rocprofv3 evidence (PROF_MODE=zlib tools/prof_single.sh): one 16 MiB zlib level-6 stream through hdlz_inflate_batch_ws(nstreams = 1) "
                                              "(k_any_* + k_par_emit / k_par_jump + the idle fixed-block chain + the fall-back launch), every counter SUMMED over the kernels of one call; libhdlz 0x%06x\n" % VERSION + txt)
    vals, cur = {}, None
    for ln in txt.splitlines():
        m = re.match(r"\s*family (\w+)\s*$", ln)
        if m:
            cur = m.group(1)
        m = re.search(r"(\w+)\s+n=\d+ mean=([0-9.e+]+)", ln)
        if m and cur == "k_par":
            vals[m.group(1)] = float(m.group(2))
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        key = "k_any|stream=16777216|level=6"
        T[key] = {"traffic_bytes": int(vals["FETCH_SIZE"] * 2 * 1024 + vals["WRITE_SIZE"] * 1024), "source": dest,
                  "fetch_size_kb": vals["FETCH_SIZE"], "write_size_kb": vals["WRITE_SIZE"],
                  "kernel": "k_any_* + k_par_emit/jump (STARTD: all kernels of hdlz_inflate_batch(nstreams = 1))", "grid": None,
                  "hdlz_version": VERSION, "valu_insts": vals.get("SQ_INSTS_VALU"), "salu_insts": vals.get("SQ_INSTS_SALU"),
                  "note": "round 6: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, summed over the kernels of one call"}
        print(key, T[key]["traffic_bytes"])
json.dump(T, open(tj, "w"), indent=1)
