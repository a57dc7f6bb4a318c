#!/bin/bash
# A/B several libhdlz builds on the configs[3] inflate line: tools/ab_inflate.sh lib1.so lib2.so ...   (extra bench args: AB_ARGS)
for lib in "$@"; do
  echo "== $lib"
  HDLZ_LIB="$PWD/$lib" python bench.py --mode inflate --steps 5 --warmup 2 --cpu-seconds 0 ${AB_ARGS:-} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value_MBps',d['value'],'kernel_ms',d['roofline']['kernel_ms_avg'],'min',d['roofline']['kernel_ms_min'],'frac',d['roofline']['frac'])"
done
