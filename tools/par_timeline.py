import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "k_par" in r["Kernel_Name"] or "k_inflate_dyn" in r["Kernel_Name"]]
idx=[i for i,r in enumerate(rows) if "k_par_spec" in r["Kernel_Name"]][-2]
t0=int(rows[idx]["Start_Timestamp"])
for r in rows[idx:idx+16]:
    print("%-28s start %8.1f us  dur %8.1f us"%(r["Kernel_Name"].split("(")[0][-28:], (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
    if "finish" in r["Kernel_Name"]: break
