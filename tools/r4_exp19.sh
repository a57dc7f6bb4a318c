#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/${1:-r4_exp19}; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "compress or stream or variants or sweep or fixture or port or chain" 2>&1 | tail -3 > $out/pytest.txt
timeout 600 python tools/adversarial_cw256.py 1024 2>&1 | grep -v amdgpu.ids | tail -14 >> $out/pytest.txt
{
for r in 1 2; do
echo "## text CW256";  AB_ARGS="--data text --block-size 65536 --blocks 16384 --cwindow 256" bash tools/ab.sh hdl_deflate_amd/lib/libhdlz.so
echo "## families CW256";  AB_ARGS="--block-size 65536 --blocks 16384 --cwindow 256" bash tools/ab.sh hdl_deflate_amd/lib/libhdlz.so
echo "## text CW64";  AB_ARGS="--data text --block-size 65536 --blocks 16384 --cwindow 64" bash tools/ab.sh hdl_deflate_amd/lib/libhdlz.so
echo "## families CW64";  AB_ARGS="--block-size 65536 --blocks 16384 --cwindow 64" bash tools/ab.sh hdl_deflate_amd/lib/libhdlz.so
echo "## headline";  AB_ARGS="" bash tools/ab.sh hdl_deflate_amd/lib/libhdlz.so
done
} > $out/lines.txt 2>&1
cat $out/pytest.txt $out/lines.txt
