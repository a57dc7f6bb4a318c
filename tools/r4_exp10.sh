#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r4_exp10; mkdir -p $out
bash tools/profile_inflate.sh r4i > $out/pmc_inflate.txt 2>&1
bash tools/profile_mem.sh r4m > $out/pmc_mem.txt 2>&1
grep -v amdgpu.ids $out/pmc_inflate.txt | tail -32; grep -v amdgpu.ids $out/pmc_mem.txt | tail -24
