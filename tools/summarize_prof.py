"""Condense the rocprofv3 CSVs written by tools/profile.sh into a small text summary
(kernel stats + per-dispatch PMC averages for our kernels)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
OURS = ("k_compress", "k_inflate", "k_par_", "k_stream_", "k_collect", "k_emit", "k_compact", "k_archive")


def find(sub, pat):
    return sorted(glob.glob(os.path.join(root, sub, "**", pat), recursive=True))


print("== rocprofv3 --kernel-trace --stats (top kernels by total time) ==")
for f in find("trace", "*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", r.get("Total_Duration(ns)", 0)) or 0))
    for r in rows[:8]:
        name = r.get("Name", "?")[:90]
        print("%-90s calls=%s total_ns=%s avg_ns=%s pct=%s" % (
            name, r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage")))

for sub in ("pmc_sq", "pmc_lds", "pmc_fetch", "pmc_write", "pmc_m1", "pmc_m2", "pmc_m3", "pmc_m4"):
    for f in find(sub, "*counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(list))
        meta = {}
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if not any(o in k for o in OURS):
                continue
            kk = k.split("(")[0][-40:]
            acc[kk][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[kk] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"),
                        r.get("Scratch_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
        print("== PMC pass %s (per-dispatch mean over our kernels) ==" % sub)
        for kk, cs in acc.items():
            print(" kernel %s  vgpr/agpr/sgpr/lds/scratch/grid/wg=%s" % (kk, meta[kk]))
            for c, v in sorted(cs.items()):
                print("   %-24s n=%d mean=%.6g" % (c, len(v), sum(v) / len(v)))
