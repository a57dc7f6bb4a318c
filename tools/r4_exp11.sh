#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r4_exp11; mkdir -p $out
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $out/pytest.txt
bash tools/profile.sh r4cw256 --no-end-to-end --data text --block-size 65536 --blocks 16384 --cwindow 256 > $out/pmc_cw256.txt 2>&1
bash tools/profile.sh r4cw64 --no-end-to-end --data text --block-size 65536 --blocks 16384 --cwindow 64 > $out/pmc_cw64.txt 2>&1
cat $out/pytest.txt; grep -v amdgpu.ids $out/pmc_cw256.txt | tail -30; grep -v amdgpu.ids $out/pmc_cw64.txt | tail -30
