#!/bin/bash
set -uo pipefail
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r4_exp2
rm -rf "$out"; mkdir -p "$out"
timeout 200 tools/ubench/valu_cycles > "$out/ubench_valu_cycles.txt" 2>&1
timeout 900 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_parity.py -x -q -m gpu -k "rccl or bench or round_trip or host or shards" > "$out/pytest_new.txt" 2>&1; echo "pytest rc=$?"
timeout 600 python bench.py --mode roundtrip --steps 5 --warmup 2 > "$out/roundtrip.json" 2> "$out/roundtrip.err"; echo "roundtrip rc=$?"
timeout 600 python bench.py --mode inflate --steps 5 --warmup 2 > "$out/inflate.json" 2> "$out/inflate.err"; echo "inflate rc=$?"
tail -3 "$out/pytest_new.txt"
