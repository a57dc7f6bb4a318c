#!/bin/bash
# copies what tools/evidence_r6b.sh left under gpurun_out/ev_r6 into profiles/ (run here, after the gpurun call)
set -euo pipefail
cd "$(dirname "$0")/.."
E=gpurun_out/ev_r6
python tools/update_traffic.py $E | tail -4
(tail -1 $E/default_cmd_bench_line.json; tail -1 $E/default_cmd_bench_line_plain.json) > profiles/r06_default_cmd_bench_lines.txt
cp $E/default_cmd_kernel_stats.csv profiles/r06_default_cmd_kernel_stats.csv
grep -v amdgpu.ids $E/any_streams.txt | cut -c1-400 > profiles/r06_any_streams.txt
grep -v amdgpu.ids $E/single_stream.txt > profiles/r06_single_stream_inflate.txt
grep -v amdgpu.ids $E/fuzz_any.txt > profiles/r06_fuzz_any.txt
cp $E/bench_detail.json profiles/r06_bench_detail.json
(sed -n 1,3p profiles/r06_any_batches.txt; grep -v amdgpu.ids $E/any_batches.txt | grep -v "ALL OK" | awk 'NR==27{print ""; print "# small streams (tools/dev_any.py batch small): the chain takes 0.55 ms whatever the stream; from 4 KiB of compressed bytes on it beats one wave,"; print "# for calls of up to 64 / 256 / 512 streams (< 8 KiB / < 16 KiB / >= 16 KiB of compressed bytes per stream)"} {print}') > /tmp/ab.txt
cp /tmp/ab.txt profiles/r06_any_batches.txt
