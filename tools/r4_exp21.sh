#!/bin/bash
# round timing of k_inflate_tok<false> per family of own 64 KiB streams, for the libraries given (timing builds)
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r4_exp21; mkdir -p $out
for lib in "$@"; do
export HDLZ_LIB=$PWD/$lib
echo "== $lib"
for f in ${FAMS:-3 1 2 4 1,2,3,4}; do FAM=$f python tools/exp_tok_timing.py 131072 65536 own; done
done > $out/tok_timing.txt 2>&1
cat $out/tok_timing.txt
