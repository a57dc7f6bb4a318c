#!/bin/bash
# A/B of libhdlz builds on the three inflate lines: tools/r4_ab3.sh lib1 lib2 ...
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for lib in "$@"; do
echo "== $lib"
export HDLZ_LIB=$PWD/$lib
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "inflate or round_trip" 2>&1 | tail -1
for r in 1 2; do
python bench.py --mode inflate --steps 5 --warmup 2 --cpu-seconds 0 --no-end-to-end 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('configs[3]   value_MBps',d['value'],'kernel_ms',d['roofline']['kernel_ms_avg'],'min',d['roofline']['kernel_ms_min'])"
python bench.py --mode inflate --steps 5 --warmup 2 --cpu-seconds 0 --no-end-to-end --zlib-strategy default --streams 262144 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dynamic 2KiB value_MBps',d['value'],'ms',d['ms_per_step'])"
python bench.py --mode roundtrip --steps 5 --warmup 2 --cpu-seconds 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('configs[4]RT value_MBps',d['value'],'kernel_ms',d['roofline']['kernel_ms_avg'])"
done
done
