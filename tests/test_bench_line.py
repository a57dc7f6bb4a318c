"""CPU: the ONE JSON line bench.py prints stays parseable by the driver (the round-5 line grew to 20.4 KB and
`BENCH_r05.json.parsed` was None).  `bench.slim_line` is fed a worst-case result built from canned dicts -- every entry with the
longest notes, workload texts and kernel names bench.py can produce, and more secondary entries than it has -- and must keep the
contract's keys, `roofline` and `cpu_baseline` under `bench.LINE_MAX` bytes."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import bench  # noqa: E402

PROSE = "x" * 700


def _roofline(kernel="k_compress<1, true, true>"):
    return {"bound": "hbm", "kernel": kernel, "achieved": 749.5, "peak": 8000.0, "unit": "GB/s", "frac": 0.0937, "traffic": 3463833600,
            "traffic_source": "profiles/r06_a_rather_long_file_name_pmc_summary.txt", "algorithmic_bytes_per_launch": 13454487491,
            "kernel_ms_avg": 18.0816, "kernel_ms_median": 18.08, "kernel_ms_min": 18.01, "launches_timed": 20, "note": PROSE,
            "issue": {"valu_insts_per_launch": 12834600000, "salu_insts_per_launch": 1234567890, "valu_insts_per_byte": 1.494,
                      "valu_fast_frac": 0.605, "cycles_per_valu_inst": 2.47, "est_valu_pipe_cycles": 7800000, "kernel_cycles": 10300000,
                      "shader_clock_ghz": 2.29, "frac": 0.746, "source": "profiles/r06_a_rather_long_file_name_pmc_summary.txt", "note": PROSE}}


def _entry(k):
    return {"name": "secondary entry number %d with a long name" % k, "metric": "inflate_output_throughput (" + "y" * 120 + ")",
            "value": 594341.9, "unit": "MB/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 3.6132, "higher_is_better": True,
            "dtype": "u8", "data": "synthetic", "config": {"workload": PROSE, "streams": 1 << 20, "block_bytes": 2048},
            "compression_ratio_out_over_in": 0.2589, "input_MBps": 153864.8, "inflate_MBps": 1.0, "compress_MBps": 2.0,
            "roofline": _roofline("k_par_* (STARTD: all kernels of hdlz_inflate_batch, 256 streams) and then some"),
            **({"compress_roofline": _roofline("k_stream_* (STARTC: all kernels of hdlz_compress_stream)")} if k % 4 == 0 else {}),
            "end_to_end": {"ms_median": 1.0, "note": PROSE, "pipelined": {"ms_all": [1.0] * 8, "note": PROSE}}, "note": PROSE}


def _worst_case(nsec):
    return {"metric": bench.METRIC_C32, "value": 477323.7, "unit": "MB/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 4.499,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": PROSE, "cwindow": 32, "maxmatch": 10, "blocks_per_gpu": 1 << 20, "block_bytes": 2048, "parallelism": PROSE},
            "per_gpu_MBps": 477323.7, "compression_ratio_out_over_in": 0.5713, "roofline": dict(_roofline(), device_copy_GBps=4960.3,
                                                                                                 frac_of_device_copy=0.1511),
            "archive": {"archive_ms": 0.611, "compress_archive_ms": 5.1, "input_MBps": 4.0e5, "archive_bytes": 1226951725,
                        "two_pass": {"scan_plus_compact_ms": 0.83, "compress_scan_compact_ms": 5.35, "input_MBps": 4.0e5}, "note": PROSE},
            "end_to_end": {"ms_median": 85.7, "ms_min": 85.6, "input_MBps": 25059.4, "h2d_ms": 37.5, "h2d_GBps": 57.3, "d2h_ms": 42.8,
                           "d2h_GBps": 56.8, "d2h_bytes": 2436890624, "reps": 5, "note": PROSE,
                           "pipelined": {"ms_median": 45.5, "ms_min": 45.4, "ms_all": [45.4] * 5, "input_MBps": 47225.7, "note": PROSE}},
            "cpu_baseline": {"value": 835.5, "unit": "MB/s", "cores": 256, "kind": "port", "stock_zlib_level1_zfixed_single_core_MBps": 123.2,
                             "sample": PROSE, "single_thread_MBps": 67.3, "note": PROSE,
                             "reference_constants": {"fpga_100MHz_3cyc_per_byte_MBps": 33, "standin_sim_KBps": "0.5-1 (BASELINE.md)"}},
            "secondary": [_entry(k) for k in range(nsec)]}


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline")


def test_worst_case_line_fits_and_keeps_the_contract():
    for nsec in (0, 9, 14, 40):                                  # bench.py has 9 secondary entries in round 5; 40: the shedding steps
        line = bench.slim_line(_worst_case(nsec), bench.DETAIL_DEFAULT)
        assert len(line) < bench.LINE_MAX and "\n" not in line
        r = json.loads(line)
        for k in CONTRACT:
            assert k in r, k
        assert r["roofline"]["frac"] == 0.0937 and r["roofline"]["traffic"] == 3463833600 and r["roofline"]["issue"]["frac"] == 0.746
        assert r["cpu_baseline"]["kind"] == "port" and r["cpu_baseline"]["cores"] == 256 and len(r["cpu_baseline"]["sample"]) <= 160
        assert r["detail"] == bench.DETAIL_DEFAULT and PROSE not in line
        if nsec <= 14:                                           # every secondary entry keeps its numbers
            assert len(r.get("secondary", [])) == nsec and "secondary_truncated" not in r
            for e in r.get("secondary", []):
                assert e["value"] == 594341.9 and e["ms_per_step"] == 3.6132
                assert e["roofline"]["frac"] == 0.0937 and e["roofline"]["traffic"] == 3463833600
                assert e["roofline"]["algorithmic_bytes_per_launch"] == 13454487491 and e["roofline"]["kernel_ms_avg"] == 18.0816
                assert e["roofline"]["issue"]["frac"] == 0.746


def test_the_round_5_line_that_was_not_parsed_now_fits():
    """the very line of profiles/r05_default_cmd_bench_lines.txt (20 420 bytes, BENCH_r05.parsed == None) through the slimmer"""
    path = os.path.join(REPO, "profiles", "r05_default_cmd_bench_lines.txt")
    lines = [ln for ln in open(path) if ln.startswith("{")]
    assert lines and len(lines[0]) > 16384
    full = json.loads(lines[0])
    r = json.loads(bench.slim_line(full, bench.DETAIL_DEFAULT))
    assert len(bench.slim_line(full, bench.DETAIL_DEFAULT)) < 8192
    assert r["value"] == full["value"] and r["ms_per_step"] == full["ms_per_step"] and r["roofline"]["frac"] == full["roofline"]["frac"]
    assert [e["name"] for e in r["secondary"]] == [e["name"] for e in full["secondary"]]
    assert [e["value"] for e in r["secondary"]] == [e["value"] for e in full["secondary"]]


def test_emit_writes_the_detail_file(tmp_path, capsys):
    class A:
        detail = str(tmp_path / "detail.json")
    res = _worst_case(3)
    bench.emit(res, A)
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1 and len(out[0]) < bench.LINE_MAX
    assert json.loads(out[0])["detail"] == A.detail
    assert json.load(open(A.detail)) == res                      # nothing is lost: the prose lives in the detail file
