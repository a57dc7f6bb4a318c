#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r4_exp3; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/tr -o t -- python tools/exp_inflate_host2.py 128 > $out/log.txt 2>&1
python - <<'PY' > gpurun_out/r4_exp3/timeline.txt 2>&1
import csv, glob
rows=[]
for f in glob.glob('gpurun_out/r4_exp3/tr/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:60], r.get('Stream_Id','?')))
for f in glob.glob('gpurun_out/r4_exp3/tr/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY '+r.get('Direction','?')+' '+r.get('Bytes', r.get('Size','?')), r.get('Stream_Id','?')))
rows.sort()
t0=rows[0][0]
keep=[r for r in rows if ('hdlz' in r[2] or 'COPY' in r[2])]
for r in keep[-140:]:
    print("%10.3f ms  +%8.3f ms  %-62s s=%s" % ((r[0]-t0)/1e6, (r[1]-r[0])/1e6, r[2], r[3]))
PY
find $out/tr -name "*.csv" -delete
tail -5 $out/log.txt
