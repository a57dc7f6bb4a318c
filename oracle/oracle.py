"""ctypes front-end for oracle/hdlz_oracle.c (the CPU restatement of deflate.py's STARTC/STARTD
paths).  TEST INFRASTRUCTURE ONLY: the checker, never the product."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libhdlz_oracle.so")

OK, E_SHORT_INPUT, E_OUT_CAPACITY, E_BAD_BTYPE, E_BAD_DISTANCE, E_NO_EOF, E_DYNAMIC_UNSUPPORTED, \
    E_BAD_SYMBOL, E_BAD_PARAM = range(9)
E_HIP, E_BAD_TREE = 9, 10
INFLATE_ASSUME_FIXED = 1
INFLATE_ONEBLOCK = 8


def build(force=False):
    src = os.path.join(_HERE, "hdlz_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        u8p = ctypes.c_void_p
        L.hdlz_oracle_out_bound.restype = ctypes.c_size_t
        L.hdlz_oracle_out_bound.argtypes = [ctypes.c_size_t]
        L.hdlz_oracle_compress.restype = ctypes.c_int
        L.hdlz_oracle_compress.argtypes = [u8p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, u8p,
                                           ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        L.hdlz_oracle_inflate.restype = ctypes.c_int
        L.hdlz_oracle_inflate.argtypes = [u8p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_uint32, u8p,
                                          ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        L.hdlz_oracle_tokens.restype = ctypes.c_int
        L.hdlz_oracle_tokens.argtypes = [u8p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, u8p, u8p, u8p,
                                         ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        for name in ("hdlz_oracle_compress_batch",):
            f = getattr(L, name)
            f.restype = ctypes.c_int
            f.argtypes = [u8p, u8p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, u8p, ctypes.c_size_t,
                          u8p, u8p, ctypes.c_int]
        L.hdlz_oracle_inflate_batch.restype = ctypes.c_int
        L.hdlz_oracle_inflate_batch.argtypes = [u8p, u8p, ctypes.c_size_t, ctypes.c_uint, u8p,
                                                ctypes.c_size_t, u8p, u8p, ctypes.c_int]
        L.hdlz_oracle_out_codes.restype = ctypes.POINTER(ctypes.c_uint16)
        L.hdlz_oracle_stat_leaves.restype = ctypes.POINTER(ctypes.c_uint16)
        _lib = L
    return _lib


def out_bound(n):
    return int(lib().hdlz_oracle_out_bound(n))


def compress(data, cwindow=32, maxmatch=10):
    """-> (status, bytes)"""
    data = bytes(data)
    cap = out_bound(len(data)) + 8
    out = ctypes.create_string_buffer(cap)
    ol = ctypes.c_size_t(0)
    rc = lib().hdlz_oracle_compress(data, len(data), cwindow, maxmatch, out, cap, ctypes.byref(ol))
    return rc, out.raw[:ol.value]


def inflate(z, flags=0, obsize=0, out_cap=None):
    """-> (status, bytes)"""
    z = bytes(z)
    cap = out_cap if out_cap is not None else min(1 << 24, max(1 << 16, 1032 * len(z) + 258))   # deflate expands <= 1032:1
    out = ctypes.create_string_buffer(cap)
    ol = ctypes.c_size_t(0)
    rc = lib().hdlz_oracle_inflate(z, len(z), flags, obsize, out, cap, ctypes.byref(ol))
    return rc, out.raw[:ol.value]


def tokens(data, cwindow=32, maxmatch=10):
    """-> list of (pos, len, dist_or_literal); len==0 means literal"""
    data = bytes(data)
    n = len(data)
    pos = np.zeros(n + 1, np.uint32)
    ln = np.zeros(n + 1, np.uint16)
    ds = np.zeros(n + 1, np.uint16)
    nt = ctypes.c_size_t(0)
    rc = lib().hdlz_oracle_tokens(data, n, cwindow, maxmatch, pos.ctypes.data, ln.ctypes.data,
                                  ds.ctypes.data, n + 1, ctypes.byref(nt))
    if rc != OK:
        raise ValueError("oracle status %d" % rc)
    k = nt.value
    return list(zip(pos[:k].tolist(), ln[:k].tolist(), ds[:k].tolist()))


def compress_batch(in_u8, in_off, cwindow=32, maxmatch=10, out_pitch=None, nthreads=1, out=None):
    """numpy batch driver: in_u8 uint8[total], in_off uint64[B+1] -> (out uint8[B,pitch], out_len, status).
    Pass a pre-touched `out` to keep first-touch page faults out of a timed region."""
    in_u8 = np.ascontiguousarray(in_u8, np.uint8)
    in_off = np.ascontiguousarray(in_off, np.uint64)
    B = len(in_off) - 1
    if out_pitch is None:
        out_pitch = out.shape[1] if out is not None else out_bound(int((in_off[1:] - in_off[:-1]).max()) if B else 0)
    if out is None:
        out = np.zeros((B, out_pitch), np.uint8)
    assert out.shape == (B, out_pitch) and out.dtype == np.uint8 and out.flags.c_contiguous
    out_len = np.zeros(B, np.uint32)
    status = np.zeros(B, np.uint32)
    lib().hdlz_oracle_compress_batch(in_u8.ctypes.data, in_off.ctypes.data, B, cwindow, maxmatch,
                                     out.ctypes.data, out_pitch, out_len.ctypes.data, status.ctypes.data, nthreads)
    return out, out_len, status


def inflate_batch(in_u8, in_off, out_pitch, flags=0, nthreads=1):
    in_u8 = np.ascontiguousarray(in_u8, np.uint8)
    in_off = np.ascontiguousarray(in_off, np.uint64)
    B = len(in_off) - 1
    out = np.zeros((B, out_pitch), np.uint8)
    out_len = np.zeros(B, np.uint32)
    status = np.zeros(B, np.uint32)
    lib().hdlz_oracle_inflate_batch(in_u8.ctypes.data, in_off.ctypes.data, B, flags, out.ctypes.data,
                                    out_pitch, out_len.ctypes.data, status.ctypes.data, nthreads)
    return out, out_len, status
