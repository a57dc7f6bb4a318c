#!/usr/bin/env python3
"""profiles/traffic.json from the PMC summaries of tools/evidence_r4.sh: per bench entry the HBM bytes per launch (FETCH_SIZE x 2
per the gfx950 correction of MI355X_MICROARCH.md + WRITE_SIZE, KB -> bytes; per-dispatch means of separate --pmc passes) and the
issue counters (SQ_INSTS_VALU / _SALU, GRBM_GUI_ACTIVE), together with WHAT was measured -- kernel symbol, launch grid, library
version -- so that bench.py can refuse an entry that does not describe the kernel it just launched.
usage: tools/update_traffic.py gpurun_out/ev_r4        (copies the summaries to profiles/r04_<name>_pmc_summary.txt)"""
import json, os, re, sys
root = sys.argv[1]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
hdr = open(os.path.join(REPO, "include", "hdlz.h")).read()
VERSION = int(re.search(r"#define\s+HDLZ_VERSION\s+(0x[0-9a-fA-F]+)", hdr).group(1), 16)
RND = "r04"
# summary file, traffic.json key, kernel symbol (prefix) the counters are taken from, committed copy
SPEC = [("pmc_cfg1.txt", "k_compress<1>|blocks=1048576|block=2048|data=families", "k_compress<1, true, true>", "cfg1"),
        ("pmc_cfg5.txt", "k_compress<1>|blocks=131072|block=65536|data=families", "k_compress<1, true, false>", "cfg5"),
        ("pmc_cfg2.txt", "k_compress<2>|blocks=16384|block=65536|data=text", "k_compress<2, true, false>", "cfg2"),
        ("pmc_cw256.txt", "k_compress<8>|blocks=16384|block=65536|data=text", "k_compress<8, true, false>", "cw256"),
        ("pmc_inflate.txt", "k_inflate_tok|streams=1048576|block=2048|fixed", "k_inflate_tok<false, 288u>", "inflate_tok"),
        ("pmc_inflate_dyn.txt", "k_inflate_tok|streams=262144|block=2048|default", "k_inflate_tok<false, 288u> + k_inflate_tok<true, 144u>", "inflate_tokdyn"),
        ("pmc_roundtrip.txt", "k_inflate_tok|roundtrip|streams=131072|block=65536", "k_inflate_tok<false, 288u>", "roundtrip")]
tj = os.path.join(REPO, "profiles", "traffic.json")
T = json.load(open(tj))
for fn, key, kern, tag in SPEC:
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
        "# rocprofv3 evidence (tools/evidence_r4.sh -> tools/profile*.sh: kernel stats + SQ / LDS / FETCH / WRITE (+ TA / TCP / TCC) passes, "
        "separate --pmc passes, per-dispatch means; libhdlz 0x%06x)\n" % VERSION + txt)
    e = {"traffic_bytes": int(f * 2 * 1024 + w * 1024), "source": dest, "fetch_size_kb": f, "write_size_kb": w,
         "kernel": kern, "grid": grid[0], "hdlz_version": VERSION,
         "valu_insts": val("SQ_INSTS_VALU"), "salu_insts": val("SQ_INSTS_SALU"), "gui_active": val("GRBM_GUI_ACTIVE"),
         "cycles_per_valu_inst": 4.0,
         "issue_note": "est_issue_cycles = SQ_INSTS_VALU x 4 cycles / 1024 SIMDs: an UPPER price -- measured in shader cycles "
                       "(profiles/r04_ubench/ubench_valu_cycles*.txt) a wave64 instruction costs 1.96 (add/sub/and/or/xor/lshr/ashr/mov/min_u16/bitop3) "
                       "or 3.25 cycles (everything else) at >= 4 waves per SIMD, 2.4 / 4.4 at two waves, 4.9 at one; the compress tile mixes 59 % / 41 % "
                       "of the two classes (tools/phase_count.py); kernel_cycles = GRBM_GUI_ACTIVE / 8 XCDs (GRBM_GUI_ACTIVE / wall time = 2.4 GHz; "
                       "s_memtime / wall time = 1.8-2.1 GHz under these loads)",
         "note": "round 4, %s: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE" % kern}
    rd, wr = val("TCC_EA0_RDREQ"), val("TCC_EA0_WRREQ")
    if rd is not None:
        e["tcc_ea0_rdreq"] = rd; e["tcc_ea0_wrreq"] = wr
        e["note"] += "; TCC_EA0_RDREQ / WRREQ = the 64-byte requests that left the L2 (the far history of the copies: one sector per token)"
    T[key] = e
    print(key, e["traffic_bytes"], "grid", e["grid"], "valu", e["valu_insts"])
json.dump(T, open(tj, "w"), indent=1)
