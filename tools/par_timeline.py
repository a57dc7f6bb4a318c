"""kernel timeline of ONE single-stream STARTD call from a rocprofv3 --kernel-trace directory (the last complete chain of k_par_* kernels)
usage: rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python tools/bench_single_stream.py 16; python tools/par_timeline.py DIR"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_par" in r["Kernel_Name"] or "k_inflate_dyn" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = "k_par_head" if any("k_par_head" in r["Kernel_Name"] for r in rows) else "k_par_spec"
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]][-2]
t0 = int(rows[idx]["Start_Timestamp"])
for r in rows[idx:idx + 24]:
    print("%-28s start %8.1f us  dur %8.1f us" % (r["Kernel_Name"].split("(")[0][-28:], (int(r["Start_Timestamp"]) - t0) / 1e3,
                                                  (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    if "k_inflate_dyn" in r["Kernel_Name"]:
        break
