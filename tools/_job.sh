python -m pytest tests/test_gpu_parity.py tests/test_gpu_single_stream.py -m gpu -x -q 2>&1 | tail -3
B="python bench.py --no-secondary --steps 5 --warmup 2 --cpu-seconds 0 --no-end-to-end"
for cw in 256; do
  echo "## text 64 KiB CW$cw"; $B --data text --block-size 65536 --blocks 16384 --cwindow $cw 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print(r['value'], r['compression_ratio_out_over_in'], r['roofline']['kernel_ms_avg'])"
  echo "## families 64 KiB CW$cw"; $B --block-size 65536 --blocks 16384 --cwindow $cw 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print(r['value'], r['compression_ratio_out_over_in'], r['roofline']['kernel_ms_avg'])"
done
python tools/bench_single_stream.py 1 4 16 64 256 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in 1 16; do rm -rf gpurun_out/tl$m; rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl$m -o t -- python tools/bench_single_stream.py $m > /dev/null 2>&1; echo "== $m MiB"; python tools/par_timeline.py gpurun_out/tl$m | head -12; rm -rf gpurun_out/tl$m; done
