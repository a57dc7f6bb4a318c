#!/bin/bash
# Round-6 evidence on ONE GPU box (from the repo root): rocprofv3 kernel stats of the default bench command, PMC passes (separate
# --pmc passes, kernel trace only) for every kernel whose line carries roofline.traffic / roofline.issue, the configs bench lines,
# the mapping tables, the single-stream paths, the tile timing build.  Results under gpurun_out/ev_r6/ ; tools/update_traffic.py gpurun_out/ev_r6 turns
# the summaries into profiles/r06_*_pmc_summary.txt + profiles/traffic.json (kernel symbol, grid and library version recorded).
set -uo pipefail
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/ev_r6
rm -rf "$out"; mkdir -p "$out"
# 1. the default command, traced (kernel stats: one row per workload) -- and its JSON line
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/default_trace" -o t -- python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/default_cmd_bench_line.json" 2> "$out/default_cmd.err"
find "$out/default_trace" -name "*kernel_stats.csv" -exec cp {} "$out/default_cmd_kernel_stats.csv" \;
find "$out/default_trace" -name "*kernel_trace.csv" -delete; find "$out/default_trace" -name "*agent_info.csv" -delete
python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/default_cmd_bench_line_plain.json" 2>> "$out/default_cmd.err"
# 2. PMC passes per workload
STEPS=20 WARMUP=5 bash tools/profile.sh ev_cfg1 --no-end-to-end > "$out/pmc_cfg1.txt" 2>&1
bash tools/profile.sh ev_cfg5 --no-end-to-end --block-size 65536 --blocks 131072 > "$out/pmc_cfg5.txt" 2>&1
bash tools/profile.sh ev_cfg2 --no-end-to-end --data text --block-size 65536 --blocks 16384 --cwindow 64 > "$out/pmc_cfg2.txt" 2>&1
bash tools/profile.sh ev_cw256 --no-end-to-end --data text --block-size 65536 --blocks 16384 --cwindow 256 > "$out/pmc_cw256.txt" 2>&1
bash tools/profile_inflate.sh ev_inflate > "$out/pmc_inflate.txt" 2>&1
bash tools/profile_mem.sh ev_inflate_mem >> "$out/pmc_inflate.txt" 2>&1
bash tools/profile_inflate.sh ev_inflate_dyn --zlib-strategy default --streams 262144 > "$out/pmc_inflate_dyn.txt" 2>&1
bash tools/profile_inflate.sh ev_rt --mode roundtrip > "$out/pmc_roundtrip.txt" 2>&1
bash tools/profile_inflate.sh ev_grp --inflate-kernel group --streams 262144 > "$out/pmc_inflate_grp.txt" 2>&1
bash tools/profile_mem.sh ev_grp_mem --inflate-kernel group --streams 262144 >> "$out/pmc_inflate_grp.txt" 2>&1
bash tools/prof_single.sh ev_single > "$out/pmc_single.txt" 2>&1
PROF_MODE=few bash tools/prof_single.sh ev_few > "$out/pmc_few.txt" 2>&1
PROF_MODE=zlib bash tools/prof_single.sh ev_zlib > "$out/pmc_zlib.txt" 2>&1
# 2b. the tile timing build of the headline kernel (built here if it did not travel: hipcc is on the GPU box too)
[ -f hdl_deflate_amd/lib/libhdlz_tiletime.so ] || HDLZ_VARIANT=tiletime HDLZ_DEFS="-DHDLZ_TILE_TIMING" HDLZ_ONLY="hdlz_compress" bash hdl_deflate_amd/csrc/build.sh > /dev/null 2>&1
# (wave time per phase + the clock of the cycle counter under this load)
HDLZ_LIB=$PWD/hdl_deflate_amd/lib/libhdlz_tiletime.so python tools/exp_tile_timing.py > "$out/tile_timing.txt" 2>&1
# 3. mapping tables, the configs bench lines
python tools/bench_inflate_mapping.py fixed > "$out/inflate_mapping.txt" 2>&1
python tools/bench_inflate_mapping.py default >> "$out/inflate_mapping.txt" 2>&1
python tools/bench_inflate_mapping.py own >> "$out/inflate_mapping.txt" 2>&1
bash tools/run_configs.sh > "$out/configs_bench_lines.txt" 2>&1
# 4. single-stream STARTD
python tools/bench_single_stream.py 1 4 16 64 256 > "$out/single_stream.txt" 2>&1
python tools/dev_any.py all > "$out/any_streams.txt" 2>&1
for d in gpurun_out/prof_ev_*; do find "$d" -name "*.csv" -not -name "*kernel_stats.csv" -delete; done
echo done
