#!/bin/bash
# Round-3 evidence on ONE GPU box (from the repo root): rocprofv3 kernel stats of the default bench command, PMC passes (separate
# --pmc passes, kernel trace only) for every kernel whose line carries roofline.traffic, the configs bench lines, the single-stream
# timeline.  Results under gpurun_out/ev_r3/ ; tools/update_traffic.py turns the summaries into profiles/traffic.json.
set -uo pipefail
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/ev_r3
rm -rf "$out"; mkdir -p "$out"
# 1. the default command, traced (kernel stats: one row per workload) -- and its JSON line
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/default_trace" -o t -- python bench.py > "$out/default_cmd_bench_line.json" 2> "$out/default_cmd.err"
cp "$out"/default_trace/*/t_kernel_stats.csv "$out/default_cmd_kernel_stats.csv" 2>/dev/null || find "$out/default_trace" -name "*kernel_stats.csv" -exec cp {} "$out/default_cmd_kernel_stats.csv" \;
find "$out/default_trace" -name "*kernel_trace.csv" -delete; find "$out/default_trace" -name "*agent_info.csv" -delete
# 2. PMC passes per workload
bash tools/profile.sh ev_cfg1 --no-end-to-end > "$out/pmc_cfg1.txt" 2>&1
bash tools/profile.sh ev_cfg5 --no-end-to-end --block-size 65536 --blocks 131072 > "$out/pmc_cfg5.txt" 2>&1
bash tools/profile.sh ev_cfg2 --no-end-to-end --data text --block-size 65536 --blocks 16384 --cwindow 64 > "$out/pmc_cfg2.txt" 2>&1
bash tools/profile.sh ev_cw256 --no-end-to-end --data text --block-size 65536 --blocks 16384 --cwindow 256 > "$out/pmc_cw256.txt" 2>&1
bash tools/profile_inflate.sh ev_inflate > "$out/pmc_inflate.txt" 2>&1
bash tools/profile_inflate.sh ev_inflate_dyn --zlib-strategy default --streams 262144 > "$out/pmc_inflate_dyn.txt" 2>&1
python tools/bench_inflate_mapping.py default > "$out/inflate_mapping.txt" 2>&1
python tools/bench_inflate_mapping.py fixed >> "$out/inflate_mapping.txt" 2>&1
# 3. the configs bench lines
bash tools/run_configs.sh > "$out/configs_bench_lines.txt" 2>&1
# 4. single-stream STARTD: sizes and the timelines of one 1 MiB and one 16 MiB call
python tools/bench_single_stream.py 1 4 16 64 256 > "$out/single_stream.txt" 2>&1
for m in 1 16; do
  rm -rf gpurun_out/tl$m
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl$m -o t -- python tools/bench_single_stream.py $m > /dev/null 2>&1
  echo "# kernel timeline of one $m MiB call (rocprofv3 --kernel-trace, tools/par_timeline.py):" >> "$out/single_stream.txt"
  python tools/par_timeline.py gpurun_out/tl$m >> "$out/single_stream.txt" 2>&1
  rm -rf gpurun_out/tl$m
done
rm -rf gpurun_out/prof_ev_*/trace/*/*.csv.bak 2>/dev/null
echo done
