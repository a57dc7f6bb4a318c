#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r4_exp12; mkdir -p $out
export HDLZ_LIB=$PWD/hdl_deflate_amd/lib/libhdlz_timing2.so
{
FAM=1,2,4 python tools/exp_tok_timing.py 262144 2048 zdefault
FAM=1,2,4 python tools/exp_tok_timing.py 32768 16384 zdefault
FAM=1,2,4 python tools/exp_tok_timing.py 4096 2048 zdefault
} 2>&1 | grep -v amdgpu.ids > $out/tok_timing_dyn.txt
cat $out/tok_timing_dyn.txt
