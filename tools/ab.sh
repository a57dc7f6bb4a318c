#!/bin/bash
# A/B several libhdlz builds with the same bench line: tools/ab.sh lib1.so lib2.so ...
for lib in "$@"; do
  echo "== $lib"
  HDLZ_LIB="$PWD/$lib" python bench.py --steps 5 --warmup 2 --cpu-seconds 0 --verify ${AB_VERIFY:-64} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value_MBps',d['value'],'kernel_ms',d['roofline']['kernel_ms_avg'],'frac',d['roofline']['frac'])"
done
