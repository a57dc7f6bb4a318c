#!/bin/bash
# memory-pipeline counters (TA / TCP / TCC) of the inflate bench, one --pmc pass per group (see tools/profile.sh for the method)
# usage (on the GPU box, from the repo root): tools/profile_mem.sh <tag> [bench args...]     (HDLZ_LIB selects the build)
set -uo pipefail
tag="${1:-mem}"; shift || true
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out="gpurun_out/prof_$tag"
mkdir -p "$out"
BENCH="python bench.py --mode inflate --steps 3 --warmup 1 --cpu-seconds 0 --no-end-to-end $*"
i=0
for grp in "TA_TA_BUSY TA_FLAT_READ_WAVEFRONTS GRBM_GUI_ACTIVE" \
           "TA_FLAT_WRITE_WAVEFRONTS TA_ADDR_STALLED_BY_TC_CYCLES" \
           "TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES" \
           "TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_WRREQ"; do   # (at most 2 TA / 4 TCP / 4 TCC counters fit one pass)
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d "$out/pmc_m$i" -o t -- $BENCH > "$out/bench_pmc_m$i.log" 2>&1
done
python tools/summarize_prof.py "$out" > "$out/summary_mem.txt" 2>&1
cat "$out/summary_mem.txt"
find "$out" -name "*kernel_trace.csv" -delete; find "$out" -name "*counter_collection.csv" -delete; find "$out" -name "*agent_info.csv" -delete
