#!/bin/bash
# the hash finder's fallback (values of the returning atomics put in order by a readlane loop) is never taken on this hardware:
# a -DHDLZ_HASH_FORCE_REORDER build takes it for every group; all compress parity tests + the adversarial inputs must pass on it
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r4_exp13; mkdir -p $out
export HDLZ_LIB=$PWD/hdl_deflate_amd/lib/libhdlz_reorder.so
{
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "compress or stream or variants or sweep or fixture" 2>&1 | tail -3
timeout 900 python tools/adversarial_cw256.py 512 2>&1 | grep -v amdgpu.ids
} > $out/reorder_forced.txt 2>&1
cat $out/reorder_forced.txt
