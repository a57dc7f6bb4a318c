#!/bin/bash
# Builds hdl_deflate_amd/lib/libhdlz.so for gfx950 (cross-compiles without a GPU).
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="$here/../lib"
mkdir -p "$out" "$here/_obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function"
pids=()
for f in hdlz_compress hdlz_compress_small hdlz_compress_stream hdlz_inflate hdlz_inflate_dyn hdlz_compact hdlz_api; do
  src="$here/$f.hip"; obj="$here/_obj/$f.o"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ "$here/hdlz_device.h" -nt "$obj" ] || [ "$here/hdlz_compress_common.h" -nt "$obj" ] || [ "$here/../../include/hdlz.h" -nt "$obj" ]; then
    ( "$HIPCC" $FLAGS -c "$src" -o "$obj" ) &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -o "$out/libhdlz.so" "$here/_obj/hdlz_compress.o" "$here/_obj/hdlz_compress_small.o" "$here/_obj/hdlz_compress_stream.o" "$here/_obj/hdlz_inflate.o" "$here/_obj/hdlz_inflate_dyn.o" "$here/_obj/hdlz_compact.o" "$here/_obj/hdlz_api.o"
echo "built $out/libhdlz.so"
