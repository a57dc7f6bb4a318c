#!/bin/bash
# rocprofv3 evidence for the ONE-stream paths (VERDICT r4 #3b): hdlz_compress_stream (k_stream_*) and hdlz_inflate_batch(nstreams = 1)
# (k_par_* + k_inflate_dyn's fall-back launch) on one 16 MiB stream -- kernel stats + FETCH / WRITE / SQ passes (separate --pmc passes).
# The path is a chain of kernels: tools/summarize_single.py sums every counter over the kernels of ONE call.
# usage (GPU box, repo root): tools/prof_single.sh <tag>      (PROF_MODE=few: the few-large-streams entry -- the same kernels, blockIdx.y = the stream)
set -uo pipefail
tag="${1:-r5_single}"
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out="gpurun_out/prof_$tag"
mkdir -p "$out"
BENCH="python bench.py --mode ${PROF_MODE:-single} --steps 5 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- $BENCH > "$out/bench_trace.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d "$out/pmc_sq" -o t -- $BENCH > "$out/bench_pmc_sq.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$out/pmc_fetch" -o t -- $BENCH > "$out/bench_pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$out/pmc_write" -o t -- $BENCH > "$out/bench_pmc_write.log" 2>&1
python tools/summarize_single.py "$out" > "$out/summary.txt" 2>&1
cat "$out/summary.txt"
find "$out" -name "*kernel_trace.csv" -delete; find "$out" -name "*counter_collection.csv" -delete; find "$out" -name "*agent_info.csv" -delete
