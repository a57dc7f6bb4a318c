"""Constants shared with include/hdlz.h (command codes: /root/reference/deflate.py:18)."""
IDLE, WRITE, READ, STARTC, STARTD = range(5)

(OK, E_SHORT_INPUT, E_OUT_CAPACITY, E_BAD_BTYPE, E_BAD_DISTANCE, E_NO_EOF, E_DYNAMIC_UNSUPPORTED,
 E_BAD_SYMBOL, E_BAD_PARAM, E_HIP) = range(10)
E_BAD_TREE = 10
INFLATE_ASSUME_FIXED = 1
INFLATE_LANE_PER_STREAM = 2      # mapping hints of hdlz_inflate_batch (results are identical)
INFLATE_WAVE_PER_STREAM = 4
INFLATE_GROUP_PER_STREAM = 64   # 16 lanes per stream, history in LDS (hdlz_inflate_grp.hip)
INFLATE_ONE_FIXED_BLOCK = 128   # hint: the streams are single fixed blocks (what STARTC writes): only that whole-GPU chain is launched
INFLATE_ONEBLOCK = 8             # ONEBLOCK=True build: stop at the end of the first block (deflate.py:40-41,678,1542,1617)

STATUS_NAMES = {OK: "OK", E_SHORT_INPUT: "SHORT_INPUT", E_OUT_CAPACITY: "OUT_CAPACITY", E_BAD_BTYPE: "BAD_BTYPE",
                E_BAD_DISTANCE: "BAD_DISTANCE", E_NO_EOF: "NO_EOF", E_DYNAMIC_UNSUPPORTED: "DYNAMIC_UNSUPPORTED",
                E_BAD_SYMBOL: "BAD_SYMBOL", E_BAD_PARAM: "BAD_PARAM", E_HIP: "HIP_ERROR", E_BAD_TREE: "BAD_TREE"}

# reference defaults (deflate.py:34-76)
CWINDOW = 32
MAXMATCH = 10      # MATCH10 = True
LMAX = 24


def out_bound(n):
    """worst-case compressed size: 6 + ceil((9n+10)/8)  (same as hdlz_out_bound)"""
    return 6 + (9 * n + 10 + 7) // 8


def pitch_for(n, align=16):
    """an out_pitch >= out_bound(n) rounded up to `align` bytes"""
    b = out_bound(n)
    return (b + align - 1) // align * align
