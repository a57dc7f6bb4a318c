#!/bin/bash
# Builds hdl_deflate_amd/lib/libhdlz.so for gfx950 (cross-compiles without a GPU).
# A/B builds: HDLZ_VARIANT=name HDLZ_DEFS="-DX=1 ..." build.sh  ->  lib/libhdlz_name.so (own object directory);
# select it at run time with HDLZ_LIB=hdl_deflate_amd/lib/libhdlz_name.so (see _lib.py, tools/ab.sh).
# HDLZ_ONLY="hdlz_compress hdlz_compress_small": only these sources are compiled with HDLZ_DEFS, the other objects are the main build's
# (which must exist).  `build.sh forced` builds lib/libhdlz_forced.so: the compress kernels with every fallback that this hardware never
# asks for FORCED (tests/test_gpu_forced_paths.py runs the compress parity tests on it).
if [ "${1:-}" = "forced" ]; then
  HDLZ_VARIANT=forced HDLZ_DEFS="-DHDLZ_HASH_FORCE_REORDER -DHDLZ_CHAIN_FORCE_SERIAL" \
    HDLZ_ONLY="hdlz_compress hdlz_compress_small hdlz_compress_stream hdlz_compress_chunk" exec "${BASH_SOURCE[0]}"
fi
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../lib"
var="${HDLZ_VARIANT:-}"
objdir="$here/_obj${var:+_$var}"
lib="$out/libhdlz${var:+_$var}.so"
mkdir -p "$out" "$objdir"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function ${HDLZ_DEFS:-}"
srcs="hdlz_compress hdlz_compress_small hdlz_compress_stream hdlz_compress_chunk hdlz_inflate_tok hdlz_inflate_grp hdlz_inflate_par hdlz_inflate_any hdlz_inflate_dyn hdlz_compact hdlz_api"
only="${HDLZ_ONLY:-$srcs}"
pids=()
for f in $only; do
  src="$here/$f.hip"; obj="$objdir/$f.o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$here/hdlz_device.h" -nt "$obj" ] || [ "$here/hdlz_compress_common.h" -nt "$obj" ] || [ "$here/hdlz_inflate_tables.h" -nt "$obj" ] || [ "$here/hdlz_inflate_par.h" -nt "$obj" ] || [ "$here/../../include/hdlz.h" -nt "$obj" ] || [ "${BASH_SOURCE[0]}" -nt "$obj" ]; then
    ( "$HIPCC" $FLAGS -c "$src" -o "$obj" ) &
    pids+=($!)
  fi
done
rc=0
for p in "${pids[@]:-}"; do [ -n "$p" ] && { wait "$p" || rc=1; }; done
[ $rc -eq 0 ] || { echo "compile failed" >&2; exit 1; }
objs=""; for f in $srcs; do
  case " $only " in *" $f "*) objs="$objs $objdir/$f.o";; *) objs="$objs $here/_obj/$f.o";; esac
done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$lib" $objs
echo "built $lib"
