#!/bin/bash
for lib in "$@"; do
  echo "== $lib"
  HDLZ_LIB="$PWD/$lib" python bench.py --mode inflate --steps 3 --warmup 1 --cpu-seconds 0 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value_MBps',d['value'],'kernel_ms',d['roofline']['kernel_ms_avg'])"
done
