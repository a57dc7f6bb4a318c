#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r4_exp5; mkdir -p $out
export HDLZ_LIB=$PWD/hdl_deflate_amd/lib/libhdlz_timing.so
{
python tools/exp_tok_timing.py 4096 2048 zfixed
python tools/exp_tok_timing.py 1048576 2048 zfixed
FAM=3 python tools/exp_tok_timing.py 4096 2048 own
python tools/exp_tok_timing.py 16384 65536 own
python tools/exp_tok_timing.py 131072 65536 own
FAM=3 python tools/exp_tok_timing.py 131072 65536 own
FAM=1 python tools/exp_tok_timing.py 131072 65536 own
} > $out/tok_timing.txt 2>&1
cat $out/tok_timing.txt
