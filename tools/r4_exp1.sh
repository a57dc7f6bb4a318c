#!/bin/bash
# round 4, GPU call 1: ubench raw outputs, ring A/B of k_inflate_tok, D2H experiment, baseline bench line
set -uo pipefail
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r4_exp1
rm -rf "$out"; mkdir -p "$out"
for u in valu_rate valu_rate2 valu_rate3 valu_rate4 valu_rate5 salu_rate scatter_rate lds_ops lds_dma; do
  if [ -x tools/ubench/$u ]; then
    echo "== $u" ; timeout 120 tools/ubench/$u > "$out/ubench_$u.txt" 2>&1; echo "rc=$?"
  fi
done
echo "== ring A/B (configs[3] inflate)" > "$out/ring_ab.txt"
bash tools/ab_inflate.sh hdl_deflate_amd/lib/libhdlz.so hdl_deflate_amd/lib/libhdlz_ring256.so hdl_deflate_amd/lib/libhdlz_ring64.so >> "$out/ring_ab.txt" 2>&1
timeout 300 python tools/exp_d2h.py 512 > "$out/d2h.txt" 2>&1
timeout 600 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.err"
echo done
