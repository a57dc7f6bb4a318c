"""Condense the rocprofv3 CSVs of tools/prof_single.sh: the one-stream paths are CHAINS of kernels, so every counter is summed over
all dispatches of a kernel family (k_stream_* = hdlz_compress_stream, k_par_* + k_any_* + k_inflate_dyn = hdlz_inflate_batch(nstreams = 1)) and
divided by the number of calls (= the dispatch count of a kernel that runs exactly once per call)."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
FAM = (("k_stream", ("k_stream_",), "k_stream_place"), ("k_par", ("k_par_", "k_any_", "k_inflate_dyn", "k_zero_words"), "k_par_finish"))


def find(sub, pat):
    return sorted(glob.glob(os.path.join(root, sub, "**", pat), recursive=True))


print("== rocprofv3 --kernel-trace --stats: kernels of the two one-stream paths ==")
for f in find("trace", "*kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f)) if any(x in r.get("Name", "") for fam in FAM for x in fam[1])]
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
    for fam, pats, once in FAM:
        sel = [r for r in rows if any(x in r["Name"] for x in pats)]
        calls = max([int(r["Calls"]) for r in sel if once in r["Name"]] or [1])
        tot = sum(float(r["TotalDurationNs"]) for r in sel)
        print(" family %s: %d kernels, %d calls, sum of kernel durations per call %.1f us" % (fam, len(sel), calls, tot / calls / 1e3))
        for r in sel[:12]:
            print("   %-70s calls=%s avg_ns=%s" % (r["Name"][:70], r["Calls"], r["AverageNs"]))
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    for f in find(sub, "*counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(float))
        ncall = defaultdict(int)
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            for fam, pats, once in FAM:
                if any(x in k for x in pats):
                    acc[fam][r["Counter_Name"]] += float(r["Counter_Value"])
                    if once in k:
                        ncall[(fam, r["Counter_Name"])] += 1
        print("== PMC pass %s (sum over the kernels of ONE call) ==" % sub)
        for fam, cs in acc.items():
            print(" family %s" % fam)
            for c, v in sorted(cs.items()):
                n = max(ncall[(fam, c)], 1)
                print("   %-24s n=%d mean=%.6g" % (c, n, v / n))
