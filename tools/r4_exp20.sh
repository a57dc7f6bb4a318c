#!/bin/bash
# inflate A/B over several libhdlz builds: parity subset (default lib = the build under test), then configs[3] / dynamic / configs[4] round trip per library
# usage: tools/r4_exp20.sh outdir lib1.so lib2.so ...
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/${1:-r4_exp20}; shift; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "inflate or round_trip or roundtrip or chain or port" 2>&1 | tail -3 > $out/pytest.txt
for lib in "$@"; do
echo "== $lib"
export HDLZ_LIB="$PWD/$lib"
python bench.py --mode inflate --steps 5 --warmup 2 --cpu-seconds 0 --no-end-to-end 2>$out/inflate.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('configs[3]   value_MBps',d['value'],'kernel_ms',d['roofline']['kernel_ms_avg'],'min',d['roofline']['kernel_ms_min'])"
python bench.py --mode inflate --steps 5 --warmup 2 --cpu-seconds 0 --no-end-to-end --zlib-strategy default --streams 262144 2>$out/dyn.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dynamic 2KiB value_MBps',d['value'],'ms',d['ms_per_step'])"
python bench.py --mode roundtrip --steps 5 --warmup 2 --cpu-seconds 0 2>$out/rt.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('configs[4]RT value_MBps',d['value'],'kernel_ms',d['roofline']['kernel_ms_avg'])"
done > $out/lines.txt 2>&1
cat $out/pytest.txt $out/lines.txt
