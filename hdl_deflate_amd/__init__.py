"""hdl_deflate_amd -- MI355X-native deflate engine behind the HDL-deflate command surface.

Only the hot path of tomtor/HDL-deflate is here: STARTC (fixed-Huffman LZ77 compress) and STARTD
(inflate) as hand-written HIP kernels for gfx950 (csrc/, built into lib/libhdlz.so, C-ABI in
include/hdlz.h), plus the host-side mirror of the reference's port interface (port.py).
There is NO CPU fallback: every compute entry point raises if the HIP library or a GPU is missing.
"""
from .errors import Error, HdlzStatusError, HdlzRangeError                         # noqa: F401
from .constants import (IDLE, WRITE, READ, STARTC, STARTD, OK, E_SHORT_INPUT, E_OUT_CAPACITY,   # noqa: F401
                        E_BAD_BTYPE, E_BAD_DISTANCE, E_NO_EOF, E_DYNAMIC_UNSUPPORTED, E_BAD_SYMBOL,
                        E_BAD_PARAM, E_HIP, E_BAD_TREE, INFLATE_ASSUME_FIXED, INFLATE_LANE_PER_STREAM, INFLATE_WAVE_PER_STREAM, INFLATE_GROUP_PER_STREAM, INFLATE_ONEBLOCK, INFLATE_ONE_FIXED_BLOCK,
                        STATUS_NAMES, out_bound)
from .port import Sig, DeflatePort, deflate                         # noqa: F401


def Engine(*a, **kw):
    """The HIP batch engine (imports torch lazily)."""
    from .engine import Engine as _E
    return _E(*a, **kw)
