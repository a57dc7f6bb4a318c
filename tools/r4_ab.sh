#!/bin/bash
# A/B of libhdlz builds: tools/r4_ab.sh <compress|inflate> lib1 lib2 ...   (parity: a subset of the GPU tests on the FIRST library)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mode=$1; shift
if [ "$mode" = compress ]; then
  timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "compress" 2>&1 | tail -2
  for r in 1 2; do AB_ARGS="--steps 10 --warmup 3 $AB_EXTRA" bash tools/ab.sh "$@"; done
else
  timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "inflate or round_trip" 2>&1 | tail -2
  for r in 1 2; do bash tools/ab_inflate.sh "$@"; done
fi
