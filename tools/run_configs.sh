#!/bin/bash
# the BASELINE.json configs that fit one GPU, one JSON line each (tools, not the driver contract)
set -u
B="python bench.py --no-secondary --steps 5 --warmup 2"
echo "## cfg[1] 2^20 x 2 KiB, CW32/M10";            $B 2>&1 | tail -1
echo "## cfg[1] the same blocks in a shuffled order (seed 1)"; $B --cpu-seconds 0 --no-end-to-end --no-archive --shuffle 1 2>&1 | tail -1
echo "## cfg[2] 64 KiB text, CW64/M10";             $B --cpu-seconds 0 --data text --block-size 65536 --blocks 16384 --cwindow 64 2>&1 | tail -1
echo "## cfg[2] 64 KiB text, CW32/M10 (same data)"; $B --cpu-seconds 0 --data text --block-size 65536 --blocks 16384 --cwindow 32 2>&1 | tail -1
echo "## cfg[2] 64 KiB text, CW256/M10";            $B --cpu-seconds 0 --data text --block-size 65536 --blocks 16384 --cwindow 256 2>&1 | tail -1
echo "## cfg[4] shape per GPU: 16384 x 64 KiB families, CW32/M10"; $B --cpu-seconds 0 --block-size 65536 --blocks 16384 2>&1 | tail -1
echo "## cfg[0] shape: 2^20 x 256 B, CW32/M10";     $B --cpu-seconds 0 --block-size 256 --blocks 1048576 2>&1 | tail -1
echo "## cfg[3] inflate 2^20 zlib Z_FIXED streams (token rounds)";  $B --mode inflate --steps 3 --warmup 1 2>&1 | tail -1
echo "## next-row: inflate 262144 stock-zlib default-strategy (dynamic trees) streams over 2 KiB blocks"; $B --mode inflate --steps 3 --warmup 1 --cpu-seconds 0 --zlib-strategy default --streams 262144 2>&1 | tail -1
echo "## next-row: inflate 32768 stock-zlib default-strategy streams over 16 KiB blocks"; $B --mode inflate --steps 3 --warmup 1 --cpu-seconds 0 --zlib-strategy default --stream-block 16384 --streams 32768 2>&1 | tail -1
