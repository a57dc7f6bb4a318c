#!/usr/bin/env python3
"""profiles/traffic.json from the PMC summaries of tools/evidence_r3.sh (FETCH_SIZE x 2 per the gfx950 correction of
MI355X_MICROARCH.md + WRITE_SIZE, KB -> bytes, per-dispatch means of separate --pmc passes).
usage: tools/update_traffic.py gpurun_out/ev_r3 profiles/r03_<name>_pmc.txt ..."""
import json, os, re, sys
root = sys.argv[1]
SPEC = [("pmc_cfg1.txt", "k_compress<1>|blocks=1048576|block=2048|data=families", "k_compress<1, true, true>", "profiles/r03_cfg1_pmc_summary.txt"),
        ("pmc_cfg5.txt", "k_compress<1>|blocks=131072|block=65536|data=families", "k_compress<1, true, false>", "profiles/r03_cfg5_pmc_summary.txt"),
        ("pmc_cfg2.txt", "k_compress<2>|blocks=16384|block=65536|data=text", "k_compress<2, true, false>", "profiles/r03_cfg2_pmc_summary.txt"),
        ("pmc_cw256.txt", "k_compress<8>|blocks=16384|block=65536|data=text", "k_compress<8, true, false>", "profiles/r03_cw256_pmc_summary.txt"),
        ("pmc_inflate.txt", "k_inflate_tok|streams=1048576|block=2048", "k_inflate_tok<false", "profiles/r03_inflate_tok_pmc_summary.txt"),
        ("pmc_inflate_dyn.txt", "k_inflate_tok|streams=262144|block=2048", "k_inflate_tok<true, 144", "profiles/r03_inflate_tokdyn_pmc_summary.txt")]
tj = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
T = json.load(open(tj))
for fn, key, kern, dest in SPEC:
    p = os.path.join(root, fn)
    if not os.path.exists(p):
        continue
    txt = open(p).read()
    def val(name):
        # the counter of the named kernel (the summaries list every kernel of the command per PMC pass); first match otherwise
        cur, first = "", None
        for ln in txt.splitlines():
            if ln.lstrip().startswith("kernel "):
                cur = ln
            m = re.search(r"%s\s+n=\d+ mean=([0-9.e+]+)" % name, ln)
            if m:
                if kern.replace(" ", "") in cur.replace(" ", ""):
                    return float(m.group(1))
                first = first if first is not None else float(m.group(1))
        return first
    f, w = val("FETCH_SIZE"), val("WRITE_SIZE")
    if f is None or w is None:
        print("no counters in", p)
        continue
    open(os.path.join(os.path.dirname(tj), os.path.basename(dest)), "w").write(
        "# rocprofv3 evidence (tools/evidence_r3.sh -> tools/profile.sh: kernel stats + SQ / LDS / FETCH / WRITE passes, separate --pmc passes, per-dispatch means)\n" + txt)
    old = T.get(key, {})
    T[key] = {"traffic_bytes": int(f * 2 * 1024 + w * 1024), "source": dest, "fetch_size_kb": f, "write_size_kb": w,
              "note": "round 3, %s: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE" % kern + ("; " + old["note"] if key.startswith("k_inflate_tok") and "note" in old else "")}
    print(key, T[key]["traffic_bytes"])
json.dump(T, open(tj, "w"), indent=1)
