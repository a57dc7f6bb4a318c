#!/bin/bash
# rocprofv3 evidence for bench.py: kernel-trace stats + PMC counters in SEPARATE passes
# (never combine --pmc with trace domains other than --kernel-trace on this pool).
# usage (on the GPU box, from the repo root): tools/profile.sh <tag> [bench args...]
set -uo pipefail
tag="${1:-r01}"; shift || true
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out="gpurun_out/prof_$tag"
mkdir -p "$out"
BENCH="python bench.py --no-secondary --steps ${STEPS:-3} --warmup ${WARMUP:-1} --cpu-seconds 0 --verify 0 $*"      # (STEPS / WARMUP: the driver runs 20 / 5)
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o t -- $BENCH > "$out/bench_trace.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d "$out/pmc_sq" -o t -- $BENCH > "$out/bench_pmc_sq.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE -d "$out/pmc_lds" -o t -- $BENCH > "$out/bench_pmc_lds.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d "$out/pmc_fetch" -o t -- $BENCH > "$out/bench_pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d "$out/pmc_write" -o t -- $BENCH > "$out/bench_pmc_write.log" 2>&1
python tools/summarize_prof.py "$out" > "$out/summary.txt" 2>&1
cat "$out/summary.txt"
# keep gpurun_out small (<64 MiB is merged back): drop the raw per-dispatch CSVs, keep stats + summary
find "$out" -name "*kernel_trace.csv" -delete; find "$out" -name "*counter_collection.csv" -delete; find "$out" -name "*agent_info.csv" -delete
