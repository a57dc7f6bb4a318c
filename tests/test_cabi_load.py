"""CPU: the C-ABI library loads, exports every symbol include/hdlz.h declares, and refuses to
compute without a GPU (there is no CPU fallback in the product)."""
import ctypes
import os
import re

from conftest import REPO


def _declared():
    src = open(os.path.join(REPO, "include", "hdlz.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hdlz_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from hdl_deflate_amd import _lib
    names = _declared()
    assert set(names) == set(_lib.EXPORTS)
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), n


def test_version_bound_and_strings():
    from hdl_deflate_amd import _lib, out_bound
    L = _lib.load()
    assert L.hdlz_version() == 0x000600
    for n in (0, 5, 256, 2048, 65536, 1 << 24):
        assert L.hdlz_out_bound(n) == out_bound(n) == 6 + (9 * n + 10 + 7) // 8
    assert L.hdlz_status_string(0) == b"OK" and b"SHORT" in L.hdlz_status_string(1)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        return
    from hdl_deflate_amd import _lib, E_HIP, E_BAD_PARAM
    import hdl_deflate_amd
    L = _lib.load()
    assert L.hdlz_device_count() == 0
    buf = (ctypes.c_uint8 * 64)()
    rc = L.hdlz_compress_batch(buf, None, 64, 64, 1, 32, 10, buf, 64, buf, buf, None)
    assert rc == E_HIP and b"no CPU path" in L.hdlz_last_error() or rc == E_HIP
    assert L.hdlz_compress_batch(buf, None, 64, 64, 1, 999, 10, buf, 64, buf, buf, None) == E_BAD_PARAM
    assert L.hdlz_inflate_batch(buf, None, 64, 64, 1, 0, 0, buf, 64, buf, buf, None) == E_HIP
    # hdlz_archive_batch: parameter checks come before the device is looked at; with good parameters: no device, no CPU path
    off = (ctypes.c_uint64 * 4)()
    assert L.hdlz_archive_batch(buf, 64, buf, 1 << 31, buf, 64, off, None) == E_BAD_PARAM and b"2^31" in L.hdlz_last_error()
    assert L.hdlz_archive_batch(buf, 64, buf, 1, buf, 64, None, None) == E_BAD_PARAM
    assert L.hdlz_archive_batch(buf, 64, buf, 1, buf, 64, off, None) == E_HIP
    # the entry points with caller-owned scratch: sizes are host arithmetic (no device needed), the calls have no CPU path either
    assert L.hdlz_archive_work_bytes(0) == 0 and L.hdlz_archive_work_bytes(1) >= 16 and L.hdlz_archive_work_bytes(1 << 20) >= 8 * 4096
    assert L.hdlz_archive_batch_ws(buf, 64, buf, 1, buf, 64, off, None, 0, None) == E_BAD_PARAM and b"hdlz_archive_work_bytes" in L.hdlz_last_error()
    assert L.hdlz_archive_batch_ws(buf, 64, buf, 1, buf, 64, off, buf, 4096, None) == E_HIP
    assert L.hdlz_inflate_work_bytes(0, 0, 0, 0, 0) == 0
    lanes = L.hdlz_inflate_work_bytes(1 << 20, 0, 2048, 0, 1)             # ragged lane mapping: the ordered lists, ~8 bytes per stream
    assert 8 << 20 <= lanes <= 9 << 20
    one = L.hdlz_inflate_work_bytes(1, 1 << 24, 1 << 26, 0, 0)            # ONE 16 MiB stream: the whole-GPU path's markers and lists
    assert one >= 1 << 26 and L.hdlz_inflate_work_bytes(1, 1 << 24, 1 << 26, 2, 0) < 1 << 16      # (a mapping hint keeps the batch kernels)
    assert L.hdlz_inflate_work_bytes(4096, 1 << 20, 1 << 22, 0, 0) <= (8 << 30) + (1 << 20)         # a batch: bounded by the 8 GiB budget
    assert L.hdlz_inflate_batch_ws(buf, None, 64, 64, 1, 0, 0, buf, 64, buf, buf, None, 0, None) == E_HIP
    assert L.hdlz_inflate_batch_ws(buf, None, 64, 64, 1, 16, 0, buf, 64, buf, buf, None, 0, None) == E_BAD_PARAM      # flags 16 / 32 left with round 6
    try:
        hdl_deflate_amd.Engine()
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("Engine() must fail loudly without a GPU")


def test_product_never_imports_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/"""
    pkg = os.path.join(REPO, "hdl_deflate_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".sh")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert "oracle" not in txt.lower(), (root, f)


def test_tile_phases_have_one_source():
    """match search / extension / parse / token bits / bit scatter live ONCE, in hdlz_compress_common.h; the three
    compress kernels only call them (round 1 had the text three times, one copy generated)"""
    csrc = os.path.join(REPO, "hdl_deflate_amd", "csrc")
    common = open(os.path.join(csrc, "hdlz_compress_common.h")).read()
    for fn in ("match_search", "make_tokens", "run_transfer", "chain_skips", "token_codes", "scatter_codes", "adler_run"):
        assert common.count("void %s(" % fn) + common.count("uint32_t %s(" % fn) + common.count("uint64_t %s(" % fn) == 1, fn
    for f in ("hdlz_compress.hip", "hdlz_compress_small.hip", "hdlz_compress_stream.hip"):
        txt = open(os.path.join(csrc, f)).read()
        assert "match_search<" in txt and "make_tokens<" in txt and "run_transfer(" in txt, f
        assert "umin3(m[i]" not in txt and "v_lshl_add_u32" not in txt, f      # no pasted phase bodies
    assert not os.path.exists(os.path.join(REPO, "tools", "gen_stream_kernel.py"))


def test_inflate_group_decode_has_one_source():
    """the fast path's group decode of the lane kernel (up to three literals + the match behind them, fixed blocks) lives ONCE -- the
    TOK_FIXED_GROUP text of hdlz_inflate_tok.hip, used by both decode steps of a round --, the kernel file is not included twice
    anywhere, and it keeps few preprocessor switches"""
    csrc = os.path.join(REPO, "hdl_deflate_amd", "csrc")
    txt = open(os.path.join(csrc, "hdlz_inflate_tok.hip")).read()
    assert txt.count("#define TOK_FIXED_GROUP(") == 1 and txt.count("TOK_FIXED_GROUP(false,") == 1 and txt.count("TOK_FIXED_GROUP(true,") == 1
    # the distance look-up behind a length: the group decode, the DYN fast path has its own (x_decode), the slow path
    assert txt.count("dst_at((uint32_t)") == 2
    assert sum(1 for ln in txt.splitlines() if ln.startswith("#if")) <= 5
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")):
            assert '#include "hdlz_inflate_tok.hip"' not in open(os.path.join(csrc, f)).read(), f
