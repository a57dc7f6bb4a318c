python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "compress or golden or large or cwindow or window" 2>&1 | tail -3
B="python bench.py --no-secondary --steps 5 --warmup 2 --cpu-seconds 0 --no-end-to-end"
for lib in "" wh2; do
export HDLZ_LIB=${lib:+hdl_deflate_amd/lib/libhdlz_$lib.so}
echo "#### lib=$lib"
for cw in 64 256; do
  echo "## text 64 KiB CW$cw"; $B --data text --block-size 65536 --blocks 16384 --cwindow $cw 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print(r['value'], r['compression_ratio_out_over_in'], r['roofline']['kernel_ms_avg'])"
  echo "## families 64 KiB CW$cw"; $B --block-size 65536 --blocks 16384 --cwindow $cw 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print(r['value'], r['compression_ratio_out_over_in'], r['roofline']['kernel_ms_avg'])"
done
done
