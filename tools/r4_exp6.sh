#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r4_exp6; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "inflate or round_trip or roundtrip" 2>&1 | tail -3 > $out/pytest.txt
python bench.py --mode inflate --steps 5 --warmup 2 --cpu-seconds 0 > $out/inflate.json 2>$out/inflate.err
python bench.py --mode roundtrip --steps 5 --warmup 2 --cpu-seconds 0 > $out/rt.json 2>$out/rt.err
{
HDLZ_LIB=$PWD/hdl_deflate_amd/lib/libhdlz_timing.so python tools/exp_tok_timing.py 131072 65536 own
HDLZ_LIB=$PWD/hdl_deflate_amd/lib/libhdlz_timing.so FAM=1,2,4 python tools/exp_tok_timing.py 1048576 2048 zfixed
} > $out/tok_timing.txt 2>&1
cat $out/pytest.txt; tail -c 1500 $out/inflate.json; echo; tail -c 1500 $out/rt.json; cat $out/tok_timing.txt
