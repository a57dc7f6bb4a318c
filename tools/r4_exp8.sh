#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r4_exp8; mkdir -p $out
{
HDLZ_LIB=$PWD/hdl_deflate_amd/lib/libhdlz_timing.so FAM=1,2,4 python tools/exp_tok_timing.py 1048576 2048 zfixed
HDLZ_LIB=$PWD/hdl_deflate_amd/lib/libhdlz_timing.so python tools/exp_tok_timing.py 131072 65536 own
HDLZ_LIB=$PWD/hdl_deflate_amd/lib/libhdlz_timing.so FAM=1,2,4 python tools/exp_tok_timing.py 4096 2048 zfixed
} > $out/tok_timing.txt 2>&1
bash tools/profile_inflate.sh r4i > $out/pmc_inflate.txt 2>&1
bash tools/profile_mem.sh r4m > $out/pmc_mem.txt 2>&1
cat $out/tok_timing.txt; tail -60 $out/pmc_inflate.txt; tail -40 $out/pmc_mem.txt
