"""Block chaining for inputs beyond the reference's counter range.

One START of the reference handles one stream of at most 2^LMAX bytes in and out: i_waddr, isize, o_iprogress and
o_oprogress are LMAX = 24 bits wide (deflate.py:73-76) -- 16 MiB; the port adapter raises HdlzRangeError beyond that, as
MyHDL does.  A hardware user with more data issues one START per piece; `compress_chained` is that loop for a
device-resident input: the input is cut into blocks whose INPUT and worst-case OUTPUT both fit the counters, every block
becomes a complete zlib stream (own header, own Adler-32 -- the same bytes STARTC would produce for that piece), and
the streams are compacted into one archive with an offset index (hdlz_compact_batch).  `inflate_chained` is the inverse.
All blocks go through the multi-wave stream passes together (hdlz_compress_streams), not one START after the other."""
import torch

from .constants import LMAX, OK, out_bound, pitch_for
from .errors import Error

# largest block whose worst-case output (out_bound) still fits an LMAX-bit o_oprogress, rounded down to 1 MiB
MAX_BLOCK = ((((1 << LMAX) - 1 - 8) * 8 // 9) >> 20) << 20


def plan_blocks(n, block):
    """[(offset, length)]: full blocks of `block` bytes; a tail shorter than 5 bytes (which the reference cannot compress,
    deflate.py:429-431) is avoided by shortening the last full block"""
    if n < 5:
        raise Error("an input of %d bytes cannot be compressed: the reference never starts below 5 (deflate.py:429-431)" % n)
    cuts = list(range(0, n, block)) + [n]
    if len(cuts) > 2 and cuts[-1] - cuts[-2] < 5:
        cuts[-2] -= 16
    return [(a, b - a) for a, b in zip(cuts, cuts[1:])]


def compress_chained(engine, d_in, block=8 << 20, cwindow=32, maxmatch=10):
    """d_in: flat uint8 device tensor (any length >= 5) -> (archive uint8[total], offsets int64[B+1], lengths int64[B]):
    stream b is archive[offsets[b]:offsets[b+1]] and holds input[b*block ...]."""
    assert d_in.is_cuda and d_in.dtype == torch.uint8 and d_in.dim() == 1 and d_in.is_contiguous()
    if block % 16 or block < (1 << 16) or block > MAX_BLOCK:
        raise ValueError("block must be a multiple of 16 in [64 KiB, %d]" % MAX_BLOCK)
    n = d_in.numel()
    plan = plan_blocks(n, block)
    dev = d_in.device
    rows, lens = [], []
    nfull = sum(1 for _, ln in plan if ln == block)
    if nfull:
        out, ol, st = engine.compress_batch(d_in[:nfull * block].view(nfull, block), cwindow=cwindow, maxmatch=maxmatch)
        if int((st != OK).sum().item()):
            raise Error("compress_chained: a block failed")
        rows.append(out)
        lens.append(ol)
    pitch = pitch_for(block)
    for off, ln in plan[nfull:]:                            # at most two shorter blocks at the end
        piece = torch.zeros((ln + 15) // 16 * 16 + 16, dtype=torch.uint8, device=dev)
        piece[:ln] = d_in[off:off + ln]
        if ln >= engine.STREAM_MIN:
            o, ol, st = engine.compress_stream(piece, ln, cwindow=cwindow, maxmatch=maxmatch)
        else:
            o, ol, st = engine.compress_batch(piece.view(1, -1), in_len=ln, cwindow=cwindow, maxmatch=maxmatch)
            o = o[0]
        if int(st.item()) != OK:
            raise Error("compress_chained: the tail block failed")
        row = torch.zeros((1, pitch), dtype=torch.uint8, device=dev)
        k = int(ol.item())
        row[0, :k] = o[:k]
        rows.append(row)
        lens.append(ol.view(1))
    allrows = torch.cat(rows) if len(rows) > 1 else rows[0]
    alllens = torch.cat(lens).to(torch.int32)
    archive, offs = engine.compact(allrows, alllens)
    l64 = alllens.to(torch.int64)
    offsets = torch.cat([offs, (offs[-1:] + l64[-1:])])
    return archive, offsets, l64


def inflate_chained(engine, archive, offsets, block=8 << 20, flags=0):
    """inverse of compress_chained -> flat uint8 device tensor (every stream must inflate to at most `block` bytes)"""
    B = offsets.numel() - 1
    padded = torch.cat([archive, torch.zeros(64, dtype=torch.uint8, device=archive.device)])
    offs = offsets.tolist()
    pitch = (block + 15) // 16 * 16
    parts = []
    # stream by stream: ONE large stream is decoded by the whole GPU (hdlz_inflate_par.hip); a batch of a few large streams would
    # get one wave each (9 MB/s per stream)
    for b in range(B):
        zn = offs[b + 1] - offs[b]
        out, ol, st = engine.inflate_batch(padded[offs[b]:offs[b] + zn + 64].view(1, -1), in_len=zn, out_pitch=pitch, flags=flags)
        if int(st.item()) != OK:
            raise Error("inflate_chained: a stream failed")
        parts.append(out[0, :int(ol.item())])
    return torch.cat(parts) if parts else torch.empty(0, dtype=torch.uint8, device=archive.device)
