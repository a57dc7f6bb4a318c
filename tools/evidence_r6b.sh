#!/bin/bash
# Round-6 evidence, second part (after the last changes to the whole-GPU inflate chains; the compress / batch-inflate kernels and their PMC
# passes are those of tools/evidence_r6.sh): the default command's lines, the one-stream entries, the any-block-types table, a soak.
set -uo pipefail
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/ev_r6
mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/default_trace" -o t -- python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/default_cmd_bench_line.json" 2> "$out/default_cmd.err"
find "$out/default_trace" -name "*kernel_stats.csv" -exec cp {} "$out/default_cmd_kernel_stats.csv" \;
find "$out/default_trace" -name "*kernel_trace.csv" -delete; find "$out/default_trace" -name "*agent_info.csv" -delete
python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/default_cmd_bench_line_plain.json" 2>> "$out/default_cmd.err"
cp profiles/r06_bench_detail.json "$out/bench_detail.json" 2>/dev/null
bash tools/prof_single.sh ev_single > "$out/pmc_single.txt" 2>&1
PROF_MODE=few bash tools/prof_single.sh ev_few > "$out/pmc_few.txt" 2>&1
PROF_MODE=zlib bash tools/prof_single.sh ev_zlib > "$out/pmc_zlib.txt" 2>&1
HDLZ_LIB=$PWD/hdl_deflate_amd/lib/libhdlz_dbg.so python tools/dev_any.py all > "$out/any_streams.txt" 2>&1
HDLZ_LIB=$PWD/hdl_deflate_amd/lib/libhdlz_dbg.so python tools/dev_any.py huge >> "$out/any_streams.txt" 2>&1
(python tools/dev_any.py batch sweep; python tools/dev_any.py batch small) > "$out/any_batches.txt" 2>&1
python tools/bench_single_stream.py 1 4 16 64 256 > "$out/single_stream.txt" 2>&1
HDLZ_LIB=$PWD/hdl_deflate_amd/lib/libhdlz_dbg.so python tools/fuzz_any.py --seconds 200 --seed 7 > "$out/fuzz_any.txt" 2>&1
for d in gpurun_out/prof_ev_*; do find "$d" -name "*.csv" -not -name "*kernel_stats.csv" -delete; done
echo done
