"""ctypes loader for lib/libhdlz.so (the C-ABI of include/hdlz.h).  Fails loudly: there is no
Python or CPU implementation to fall back to."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HDLZ_LIB") or os.path.join(_HERE, "lib", "libhdlz.so")   # HDLZ_LIB: A/B builds
EXPORTS = ("hdlz_version", "hdlz_status_string", "hdlz_last_error", "hdlz_device_count", "hdlz_out_bound",
           "hdlz_compress_batch", "hdlz_inflate_batch", "hdlz_compact_batch", "hdlz_archive_batch",
           "hdlz_stream_work_bytes", "hdlz_compress_stream", "hdlz_streams_work_bytes", "hdlz_compress_streams",
           "hdlz_compress_chunk", "hdlz_inflate_chunk", "hdlz_release_scratch",
           "hdlz_inflate_work_bytes", "hdlz_inflate_batch_ws", "hdlz_archive_work_bytes", "hdlz_archive_batch_ws")
_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libhdlz.so is not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or hdl_deflate_amd/csrc/build.sh -- there is no CPU fallback" % LIB_PATH)
    # torch ships its own libamdhip64.so (SONAME libamdhip64.so.7).  Import it FIRST so that libhdlz's
    # NEEDED libamdhip64.so.7 resolves to the runtime torch already loaded; the other order would put
    # two HIP runtimes in one process and torch's streams/pointers would be foreign to ours.
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    vp, u64, u32, ci = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
    L.hdlz_version.restype = ci
    L.hdlz_status_string.restype = ctypes.c_char_p
    L.hdlz_status_string.argtypes = [ci]
    L.hdlz_last_error.restype = ctypes.c_char_p
    L.hdlz_device_count.restype = ci
    L.hdlz_out_bound.restype = ctypes.c_size_t
    L.hdlz_out_bound.argtypes = [ctypes.c_size_t]
    L.hdlz_compress_batch.restype = ci
    L.hdlz_compress_batch.argtypes = [vp, vp, u64, u32, u64, ci, ci, vp, u64, vp, vp, vp]
    L.hdlz_inflate_batch.restype = ci
    L.hdlz_inflate_batch.argtypes = [vp, vp, u64, u32, u64, u32, u32, vp, u64, vp, vp, vp]
    L.hdlz_compact_batch.restype = ci
    L.hdlz_compact_batch.argtypes = [vp, u64, vp, vp, u64, vp, vp]
    L.hdlz_archive_batch.restype = ci
    L.hdlz_archive_batch.argtypes = [vp, u64, vp, u64, vp, u64, vp, vp]
    L.hdlz_stream_work_bytes.restype = ctypes.c_size_t
    L.hdlz_stream_work_bytes.argtypes = [ctypes.c_size_t]
    L.hdlz_compress_stream.restype = ci
    L.hdlz_compress_stream.argtypes = [vp, u32, ci, ci, vp, u64, vp, vp, vp, ctypes.c_size_t, vp]
    L.hdlz_streams_work_bytes.restype = ctypes.c_size_t
    L.hdlz_streams_work_bytes.argtypes = [ctypes.c_size_t, u64]
    L.hdlz_compress_streams.restype = ci
    L.hdlz_compress_streams.argtypes = [vp, u64, u32, u64, ci, ci, vp, u64, vp, vp, vp, ctypes.c_size_t, vp]
    L.hdlz_compress_chunk.restype = ci
    L.hdlz_compress_chunk.argtypes = [vp, u32, u32, ci, ci, ci, vp, u64, vp, vp]
    L.hdlz_inflate_chunk.restype = ci
    L.hdlz_inflate_chunk.argtypes = [vp, u32, ci, u32, u32, vp, u64, u32, vp, vp]
    L.hdlz_release_scratch.restype = ci
    L.hdlz_inflate_work_bytes.restype = ctypes.c_size_t
    L.hdlz_inflate_work_bytes.argtypes = [u64, u32, u64, u32, ci]
    L.hdlz_inflate_batch_ws.restype = ci
    L.hdlz_inflate_batch_ws.argtypes = [vp, vp, u64, u32, u64, u32, u32, vp, u64, vp, vp, vp, ctypes.c_size_t, vp]
    L.hdlz_archive_work_bytes.restype = ctypes.c_size_t
    L.hdlz_archive_work_bytes.argtypes = [u64]
    L.hdlz_archive_batch_ws.restype = ci
    L.hdlz_archive_batch_ws.argtypes = [vp, u64, vp, u64, vp, u64, vp, vp, ctypes.c_size_t, vp]
    _lib = L
    return L
