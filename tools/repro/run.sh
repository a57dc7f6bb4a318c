#!/bin/bash
# the evidence behind profiles/r06_graph_abort_cause.txt; writes gpurun_out/repro_graph.txt
cd "$(dirname "$0")/../.."
o=gpurun_out/repro_graph.txt; mkdir -p gpurun_out; : > $o
run() { echo "== $*" >> $o; timeout 300 "$@" >> $o 2>&1; echo "   rc=$?" >> $o; }
for m in none pool default; do for k in 1 2; do run tools/repro/graph_scratch $m $k 60 16 512 1; done; done
for i in 1 2 3; do HDLZ_LIB=hdl_deflate_amd/lib/libhdlz_poolcap.so run python tools/repro/graph_abort.py pool 30; done
for i in 1 2 3; do run python tools/repro/graph_abort.py ws 200; done
run python tools/repro/graph_abort.py pool 3
tail -5 $o
