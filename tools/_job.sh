python -m pytest tests -m gpu -x -q 2>&1 | tail -3
B="python bench.py --no-secondary --steps 5 --warmup 2 --cpu-seconds 0 --no-end-to-end"
for cw in 64 256; do
  echo "## text 64 KiB CW$cw"; $B --data text --block-size 65536 --blocks 16384 --cwindow $cw 2>&1 | tail -1 | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print(r['value'], r['compression_ratio_out_over_in'], r['roofline']['kernel_ms_avg'])"
done
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from hdl_deflate_amd import Engine
from hdl_deflate_amd.data import make_text_blocks
e = Engine()
for mib in (1, 16, 256):
    n = mib << 20
    d = make_text_blocks(mib, 1 << 20, "cuda", seed=3).reshape(-1)
    d = torch.cat([d, torch.zeros(16, dtype=torch.uint8, device="cuda")])
    for cw in (32, 64, 256):
        fn = lambda: e.compress_stream(d, n, cwindow=cw)
        o, ol, st = fn(); torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(5): o, ol, st = fn()
        ev1.record(); torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / 5
        print("one stream %4d MiB cw=%-3d %9.3f ms  %8.2f GB/s  ratio %.3f st=%d" % (mib, cw, ms, n / ms / 1e6, int(ol.item()) / n, int(st.item())), flush=True)
PY
