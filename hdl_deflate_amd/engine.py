"""Batch engine: thin Python over the C-ABI (include/hdlz.h).  torch is used for device memory and
streams only.  All tensors live on the GPU; nothing here computes on the CPU."""
import functools

import torch

from . import _lib
from .constants import OK, LMAX, INFLATE_ONE_FIXED_BLOCK, pitch_for
from .errors import Error


def _on_device(fn):
    """run an Engine method with the engine's GPU current: the C-ABI launches on the CURRENT HIP device and on torch's
    current stream of that device, so an engine built for cuda:1 must not enqueue on cuda:0 because the caller forgot
    torch.cuda.set_device"""
    @functools.wraps(fn)
    def wrapper(self, *a, **kw):
        with torch.cuda.device(self.device):
            return fn(self, *a, **kw)
    return wrapper


class Engine(object):
    """One engine per process/GPU.  Methods enqueue on torch's current stream and return device tensors."""

    def __init__(self, device=None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("hdl_deflate_amd.Engine needs a HIP device (gfx950); there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.lib.hdlz_device_count() < 1:
            raise RuntimeError("no gfx950 device visible to libhdlz")

    # -- helpers
    def _check(self, rc, what):
        if rc != OK:
            raise Error("%s failed: %s (%s)" % (what, self.lib.hdlz_status_string(rc).decode(),
                                                self.lib.hdlz_last_error().decode()))

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def _prep(self, d_in, in_off, in_len, nblocks):
        assert d_in.is_cuda and d_in.dtype == torch.uint8 and d_in.is_contiguous()
        assert d_in.device == self.device, "tensor on %s, engine on %s" % (d_in.device, self.device)
        if in_off is not None:
            assert in_off.is_cuda and in_off.dtype == torch.int64 and in_off.is_contiguous()
            nb = in_off.numel() - 1
            return in_off.data_ptr(), 0, (0 if in_len is None else in_len), nb
        assert d_in.dim() == 2 or nblocks is not None
        if d_in.dim() == 2:
            nb, pitch = d_in.shape
            return None, pitch, (pitch if in_len is None else in_len), nb
        return None, in_len, in_len, nblocks

    # -- STARTC for a batch
    @_on_device
    def compress_batch(self, d_in, in_off=None, in_len=None, nblocks=None, cwindow=32, maxmatch=10,
                       out=None, out_pitch=None, max_len=None):
        """d_in: uint8 [B, pitch] (fixed-size blocks) or flat uint8 with in_off int64[B+1] (ascending).
        Returns (out uint8[B, out_pitch], out_len int32[B], status int32[B]); per-block failures are statuses, never
        exceptions.  Ragged batches: pass `max_len` (an upper bound on the block lengths) or `out_pitch` -- without
        either the bound is computed from in_off with ONE host sync per call."""
        off_ptr, pitch, ilen, nb = self._prep(d_in, in_off, in_len, nblocks)
        if out_pitch is None:
            if max_len is None:
                max_len = ilen if in_off is None else int((in_off[1:] - in_off[:-1]).max().item()) if nb else 0
            out_pitch = pitch_for(max_len)
        if in_off is not None:
            # ragged: the bound the library gets is `max_len` (it packs small blocks per wave from it and gives a LONGER block
            # HDLZ_E_BAD_PARAM); `in_len` is not a bound here, as before round 5
            ilen = max_len if max_len is not None else 0
        if out is None:
            out = torch.empty((nb, out_pitch), dtype=torch.uint8, device=d_in.device)
        out_len = torch.empty(nb, dtype=torch.int32, device=d_in.device)
        status = torch.empty(nb, dtype=torch.int32, device=d_in.device)
        if in_off is None and ilen >= self.LARGE_BLOCK and pitch % 16 == 0 and nb <= self.MANY_WAVES and \
                out_pitch >= (self.lib.hdlz_out_bound(ilen) + 3) // 4 * 4 and self.lib.hdlz_streams_work_bytes(ilen, nb) != 0:
            # (a too small out_pitch stays on the batch path, which reports E_OUT_CAPACITY per block instead of failing the call)
            # few large blocks: one wave per block (hdlz_compress_batch) would leave the GPU idle (a 1 MiB block is 5.5 ms
            # on a wave); all tiles of all blocks go through the stream passes together instead (same bytes)
            work = torch.empty((self.lib.hdlz_streams_work_bytes(ilen, nb) + 7) // 8, dtype=torch.int64, device=d_in.device)
            rc = self.lib.hdlz_compress_streams(d_in.data_ptr(), pitch, ilen, nb, cwindow, maxmatch, out.data_ptr(), out_pitch,
                                                out_len.data_ptr(), status.data_ptr(), work.data_ptr(),
                                                work.numel() * 8, self._stream())
            self._check(rc, "hdlz_compress_streams")
            return out, out_len, status
        rc = self.lib.hdlz_compress_batch(d_in.data_ptr(), off_ptr, pitch, ilen, nb, cwindow, maxmatch,
                                          out.data_ptr(), out_pitch, out_len.data_ptr(), status.data_ptr(),
                                          self._stream())
        self._check(rc, "hdlz_compress_batch")
        return out, out_len, status

    # -- STARTC for ONE large stream, spread over the whole GPU (same bytes as compress_batch with one block)
    STREAM_MIN = 1 << 14          # measured crossover with the single-wave batch path: ~8 KiB
    LARGE_BLOCK, MANY_WAVES = 1 << 16, 1200   # compress_batch: up to this many blocks of at least this size -> stream passes
                                              # (a wave does ~180 MB/s, the stream passes ~250 GB/s: crossover ~1400 blocks)

    @_on_device
    def compress_stream(self, d_in, n, cwindow=32, maxmatch=10, out=None, work=None):
        """d_in: flat uint8 device tensor, readable up to n rounded up to 16.
        Returns (out uint8[cap], out_len int32[1], status int32[1])."""
        assert d_in.is_cuda and d_in.dtype == torch.uint8 and d_in.is_contiguous() and d_in.numel() >= (n + 15) // 16 * 16
        cap = pitch_for(n)
        if out is None:
            out = torch.empty(cap, dtype=torch.uint8, device=d_in.device)
        wb = self.lib.hdlz_stream_work_bytes(n)
        if work is None:
            work = torch.empty((wb + 7) // 8, dtype=torch.int64, device=d_in.device)
        out_len = torch.empty(1, dtype=torch.int32, device=d_in.device)
        status = torch.empty(1, dtype=torch.int32, device=d_in.device)
        rc = self.lib.hdlz_compress_stream(d_in.data_ptr(), n, cwindow, maxmatch, out.data_ptr(), out.numel(),
                                           out_len.data_ptr(), status.data_ptr(), work.data_ptr(),
                                           work.numel() * work.element_size(), self._stream())
        self._check(rc, "hdlz_compress_stream")
        return out, out_len, status

    # -- STARTD for a batch
    @_on_device
    def inflate_batch(self, d_in, in_off=None, in_len=None, nblocks=None, out_pitch=None, flags=0, obsize=0,
                      out=None, work=None):
        """STARTD for a batch: d_in uint8 [B, pitch] (a stream per row, `in_len` bytes of it: default the pitch) or flat uint8 with in_off
        int64[B + 1] -- then `in_len`, if given, is an UPPER BOUND on the stream lengths: with it a batch of large streams can take
        the whole-GPU path (a stream longer than the bound is still decoded, by the serial pass).
        `work`: the call's scratch, a uint8 device tensor (hdlz_inflate_batch_ws: the library allocates nothing); default: a tensor of
        hdlz_inflate_work_bytes(...) bytes from torch's allocator, stream-ordered like every other tensor of the call; a smaller one
        (or an empty one) gives the same results through mappings that need less.
        -> (out uint8[B, out_pitch], out_len int32[B], status int32[B]); per-stream failures are statuses."""
        off_ptr, pitch, ilen, nb = self._prep(d_in, in_off, in_len, nblocks)
        assert out_pitch is not None and out_pitch % 4 == 0
        if out is None:
            out = torch.empty((nb, out_pitch), dtype=torch.uint8, device=d_in.device)
        out_len = torch.empty(nb, dtype=torch.int32, device=d_in.device)
        status = torch.empty(nb, dtype=torch.int32, device=d_in.device)
        if work is None:
            work = torch.empty(self.lib.hdlz_inflate_work_bytes(nb, ilen, out_pitch, flags, 0 if in_off is None else 1),
                               dtype=torch.uint8, device=d_in.device)
        assert work.is_cuda and work.dtype == torch.uint8 and work.is_contiguous() and work.device == self.device
        rc = self.lib.hdlz_inflate_batch_ws(d_in.data_ptr(), off_ptr, pitch, ilen, nb, flags, obsize,
                                            out.data_ptr(), out_pitch, out_len.data_ptr(), status.data_ptr(),
                                            work.data_ptr() if work.numel() else None, work.numel(), self._stream())
        self._check(rc, "hdlz_inflate_batch_ws")
        return out, out_len, status

    # -- archive compaction
    @_on_device
    def compact(self, rows, lens, offsets=None, archive=None):
        """rows uint8[B, pitch], lens int32[B] -> (archive uint8[total], offsets int64[B]).
        `offsets` (exclusive scan, e.g. global offsets after the multi-GPU length all-gather) and `archive`
        may be supplied; otherwise they are computed / allocated here (one host sync for the size)."""
        assert rows.is_cuda and rows.dtype == torch.uint8 and rows.dim() == 2 and rows.is_contiguous()
        assert rows.device == self.device and lens.is_cuda and lens.dtype == torch.int32 and lens.numel() == rows.shape[0]
        B, pitch = rows.shape
        lens = lens.contiguous()
        if offsets is None:
            l64 = lens.to(torch.int64)
            offsets = torch.cumsum(l64, 0) - l64
        assert offsets.is_cuda and offsets.dtype == torch.int64 and offsets.numel() == B, "offsets must be int64 [B] on the GPU"
        offsets = offsets.contiguous()
        if archive is None:
            total = int((offsets[-1] + lens[-1]).item()) if B else 0
            archive = torch.empty(total, dtype=torch.uint8, device=rows.device)
        rc = self.lib.hdlz_compact_batch(rows.data_ptr(), pitch, lens.data_ptr(), offsets.data_ptr(), B,
                                         archive.data_ptr(), self._stream())
        self._check(rc, "hdlz_compact_batch")
        return archive, offsets

    @_on_device
    def archive(self, rows, lens, archive=None, offsets=None, work=None):
        """rows uint8[B, pitch], lens int32[B] -> (archive uint8[cap], offsets int64[B + 1]) in ONE launch (hdlz_archive_batch): the
        exclusive scan of the lengths is made on the device, offsets[B] is the archive's length (no host sync: read it when needed),
        offsets is directly the `in_off` of a ragged inflate / compress call.  `archive` defaults to a buffer of B * pitch bytes."""
        assert rows.is_cuda and rows.dtype == torch.uint8 and rows.dim() == 2 and rows.is_contiguous() and rows.device == self.device
        assert lens.is_cuda and lens.dtype == torch.int32 and lens.numel() == rows.shape[0]
        B, pitch = rows.shape
        lens = lens.contiguous()
        if archive is None:
            archive = torch.empty(B * pitch, dtype=torch.uint8, device=rows.device)
        if offsets is None:
            offsets = torch.empty(B + 1, dtype=torch.int64, device=rows.device)
        assert archive.is_cuda and archive.dtype == torch.uint8 and archive.is_contiguous()
        assert offsets.is_cuda and offsets.dtype == torch.int64 and offsets.numel() == B + 1 and offsets.is_contiguous()
        if work is None:                 # the call's scratch (8 bytes per 256 rows), caller-owned like every other buffer of the call
            work = torch.empty(max(8, self.lib.hdlz_archive_work_bytes(B)) // 8, dtype=torch.int64, device=rows.device)
        rc = self.lib.hdlz_archive_batch_ws(rows.data_ptr(), pitch, lens.data_ptr(), B, archive.data_ptr(), archive.numel(),
                                            offsets.data_ptr(), work.data_ptr(), work.numel() * work.element_size(), self._stream())
        self._check(rc, "hdlz_archive_batch_ws")
        return archive, offsets

    # -- the job from HOST buffers: the PCIe hops overlapped with the kernels
    @_on_device
    def compress_host(self, h_in, cwindow=32, maxmatch=10, chunk_blocks=None, h_archive=None, h_len=None, keep_buffers=True):
        """h_in: PINNED host uint8 [B, n].  Compresses every block on the GPU and returns (h_archive, h_len, total, status_bad):
        the blocks' zlib streams back to back in pinned host memory (`h_archive[:total]`; block b at the exclusive scan of
        `h_len`), their lengths, and the number of blocks whose status is not OK (0 unless n < 5 / the capacity is wrong).
        The batch goes through the GPU in chunks of `chunk_blocks` blocks on three streams -- H2D of chunk k + 1, compress +
        hdlz_archive_batch (scan + gather) of chunk k, D2H of chunk k - 1's ARCHIVE (not its pitched rows) -- so the job costs about
        the slower PCIe direction instead of H2D + kernel + D2H.  One host sync per chunk (its archive size).
        The three side streams and the staging buffers are kept on the Engine between calls (`keep_buffers=False` or
        release_host_buffers() drops them); compress_host / inflate_host of ONE Engine must not run from two threads at once."""
        assert h_in.dtype == torch.uint8 and h_in.dim() == 2 and h_in.is_contiguous() and h_in.is_pinned()
        B, n = h_in.shape
        pitch = pitch_for(n)
        if chunk_blocks is None:
            # 48 MiB of input per chunk: its archive stays below 64 MiB -- D2H copies of 64 MiB and more were seen to cost 20 ms per job
            # (73 MiB: 60 ms instead of 40) or to block the host (inflate_host); 16 .. 48 MiB chunks measure the same 39.6 .. 40.0 ms
            chunk_blocks = max(1, min(B, (48 << 20) // max(n, 1)))
        C = chunk_blocks
        if h_archive is None:
            h_archive = torch.empty(B * self.lib.hdlz_out_bound(n), dtype=torch.uint8, pin_memory=True)
        if h_len is None:
            h_len = torch.empty(B, dtype=torch.int32, pin_memory=True)
        assert h_archive.is_pinned() and h_len.is_pinned() and h_len.numel() == B and h_len.dtype == torch.int32
        assert h_archive.dtype == torch.uint8 and h_archive.dim() == 1 and h_archive.numel() >= B * self.lib.hdlz_out_bound(n), \
            "h_archive must hold B * hdlz_out_bound(n) bytes (the size of the archive is only known at the end)"
        dev = self.device
        cur = torch.cuda.current_stream()
        # streams and staging buffers are kept between calls (a fresh stream has a fresh allocator pool: device mallocs in the job)
        ctx = getattr(self, "_host_ctx", None)
        if ctx is None or ctx["key"] != (C, n):
            ctx = {"key": (C, n), "streams": [torch.cuda.Stream(dev) for _ in range(3)],
                   "d_in": [torch.empty((C, n), dtype=torch.uint8, device=dev) for _ in range(2)],
                   "d_arch": [torch.empty(C * pitch, dtype=torch.uint8, device=dev) for _ in range(2)],
                   "d_len": [torch.empty(C, dtype=torch.int32, device=dev) for _ in range(2)],
                   "d_rows": torch.empty((C, pitch), dtype=torch.uint8, device=dev),
                   "d_tot": [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(2)]}     # archive bytes, failed blocks
            self._host_ctx = ctx
        s_in, s_k, s_out = ctx["streams"]
        d_in, d_arch, d_len, d_rows, d_tot = ctx["d_in"], ctx["d_arch"], ctx["d_len"], ctx["d_rows"], ctx["d_tot"]
        for st in (s_in, s_k, s_out):
            st.wait_stream(cur)
        ev_k = [None, None]        # compute of the chunk that last used buffer pair j
        ev_out = [None, None]      # D2H of the chunk that last used buffer pair j
        base, bad = 0, 0
        pending = None             # (j, first block, blocks) of the chunk whose D2H is still to be issued

        def drain(p):
            nonlocal base, bad
            j, b0, nb = p
            ev_k[j].synchronize()
            tot = d_tot[j].tolist()                       # (the two words were written before ev_k[j])
            total, nbad = int(tot[0]), int(tot[1])
            with torch.cuda.stream(s_out):
                s_out.wait_event(ev_k[j])
                h_archive[base:base + total].copy_(d_arch[j][:total], non_blocking=True)
                h_len[b0:b0 + nb].copy_(d_len[j][:nb], non_blocking=True)
                ev_out[j] = torch.cuda.Event()
                ev_out[j].record(s_out)
            base += total
            bad += nbad

        k = 0
        try:
            for b0 in range(0, B, C):
                nb = min(C, B - b0)
                j = k & 1
                with torch.cuda.stream(s_in):
                    if ev_k[j] is not None:
                        s_in.wait_event(ev_k[j])              # the compress that read d_in[j] two chunks ago
                    d_in[j][:nb].copy_(h_in[b0:b0 + nb], non_blocking=True)
                    ev_in = torch.cuda.Event()
                    ev_in.record(s_in)
                with torch.cuda.stream(s_k):
                    s_k.wait_event(ev_in)
                    if ev_out[j] is not None:
                        s_k.wait_event(ev_out[j])             # the D2H that read d_arch[j] / d_len[j] two chunks ago
                    _, ol, st = self.compress_batch(d_in[j][:nb], cwindow=cwindow, maxmatch=maxmatch, out=d_rows[:nb], out_pitch=pitch)
                    _, off = self.archive(d_rows[:nb], ol, archive=d_arch[j])      # scan + gather in one launch (hdlz_archive_batch)
                    d_len[j][:nb].copy_(ol)
                    d_tot[j][0] = off[nb]
                    d_tot[j][1] = (st != 0).sum()
                    ev_k[j] = torch.cuda.Event()
                    ev_k[j].record(s_k)
                if pending is not None:
                    drain(pending)                            # chunk k - 1: its size is known now, its D2H runs beside chunk k's kernels
                pending = (j, b0, nb)
                k += 1
            if pending is not None:
                drain(pending)
        finally:
            # also on an exception (a failed C-ABI call in the loop): nothing stays queued on the side streams behind the caller's back
            cur.wait_stream(s_out)
            cur.wait_stream(s_k)
            cur.wait_stream(s_in)
            s_out.synchronize()
            if not keep_buffers:
                self.release_host_buffers()
        return h_archive, h_len, base, bad

    def release_host_buffers(self):
        """drop the streams and the device / pinned staging buffers compress_host and inflate_host keep between calls (about
        2 C n + 3 C pitch bytes of HBM for chunks of C blocks: ~250 MB at the default 48 MiB chunk) and give the library's cached
        scratch back to the device"""
        self._host_ctx = None
        self._ihost_ctx = None
        with torch.cuda.device(self.device):
            self._check(self.lib.hdlz_release_scratch(), "hdlz_release_scratch")

    @_on_device
    def inflate_host(self, h_z, h_off, out_pitch, flags=0, obsize=0, chunk_streams=None, h_out=None, h_len=None, h_status=None,
                     keep_buffers=True, d2h="copy"):
        """STARTD for a batch held in HOST memory: h_z PINNED flat uint8 (the zlib streams back to back), h_off int64 [B + 1] on the
        host (ascending; stream b = h_z[h_off[b]:h_off[b + 1]]).  Returns pinned (h_out uint8 [B, out_pitch], h_len int32 [B],
        h_status int32 [B]).  Chunks of `chunk_streams` streams (the first one a quarter of that: the pipeline fills sooner) on three
        streams, nothing but enqueues on the host -- no synchronisation before the end: H2D of chunk k + 1 | hdlz_inflate_batch of
        chunk k | D2H of chunk k - 1's rows in copies of at most 32 MiB (larger hipMemcpyAsync D2H copies were seen to block the host
        on ROCm 7.2).  Measured on BASELINE configs[3] (profiles/r04_inflate_host.txt): 42.5 ms against 37.7 ms for the rows alone at
        the link's 57 GB/s (round 3: 51.9 ms -- it waited on the host for a chunk's kernel before issuing its D2H, and its 48 MiB
        chunks ran the inflate kernel 43 times at the ~0.6 ms a lane-per-stream launch costs whatever its size).  What is left above the
        floor: a kernel that runs BESIDE a D2H -- the runtime's blit or a kernel of ours storing into the mapped rows, d2h="kernel" --
        completes only when the D2H's writes have drained (0.63 -> 3.1 ms in the kernel trace; its end-of-kernel write-back queues
        behind the PCIe-bound stores), so inflate and D2H alternate per chunk pair instead of overlapping freely.
        Buffers and streams are kept on the Engine between calls (see compress_host); one call at a time per Engine."""
        assert h_z.dtype == torch.uint8 and h_z.dim() == 1 and h_z.is_pinned() and out_pitch % 4 == 0
        h_off = torch.as_tensor(h_off, dtype=torch.int64)
        assert not h_off.is_cuda and h_off.dim() == 1 and h_off.numel() >= 1
        B = h_off.numel() - 1
        if h_out is None:
            h_out = torch.empty((B, out_pitch), dtype=torch.uint8, pin_memory=True)
        if h_len is None:
            h_len = torch.empty(B, dtype=torch.int32, pin_memory=True)
        if h_status is None:
            h_status = torch.empty(B, dtype=torch.int32, pin_memory=True)
        assert h_out.is_pinned() and h_len.is_pinned() and h_status.is_pinned() and tuple(h_out.shape) == (B, out_pitch)
        assert h_out.dtype == torch.uint8 and h_out.is_contiguous() and h_len.numel() == B and h_status.numel() == B
        assert h_len.dtype == torch.int32 and h_status.dtype == torch.int32
        if B == 0:
            return h_out, h_len, h_status
        if chunk_streams is None:
            # 256 MiB of rows per chunk: a lane-per-stream batch needs ~10^5 streams to fill the GPU (131 072 streams of 2 KiB take
            # 0.73 ms, 16 384 take 0.6 ms as well), so small chunks only multiply the kernel time (profiles/r04_inflate_host.txt)
            chunk_streams = max(1, min(B, (256 << 20) // max(out_pitch, 1)))
        C = chunk_streams
        piece = max(1, (32 << 20) // max(out_pitch, 1))
        starts = [0] + list(range(max(1, C // 4), B, C))               # chunk k = streams [starts[k], starts[k + 1])
        ends = starts[1:] + [B]
        zlo = [int(h_off[b0]) & ~255 for b0 in starts]                  # (the H2D copies start at aligned host addresses)
        zhi = [min((int(h_off[b1]) + 255) & ~255, h_z.numel()) for b1 in ends]
        zmax = max(b - a for a, b in zip(zlo, zhi))                      # compressed bytes of the largest chunk
        dev = self.device
        cur = torch.cuda.current_stream()
        ctx = getattr(self, "_ihost_ctx", None)
        if ctx is None or ctx["key"] != (C, out_pitch) or ctx["zcap"] < zmax or ctx["h_off"].numel() < B + 1:
            # (the inflate stream has the higher priority: its workgroups are placed before the pending ones of the row copy beside it)
            ctx = {"key": (C, out_pitch), "zcap": zmax,
                   "streams": [torch.cuda.Stream(dev), torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev)],
                   "d_z": [torch.zeros(zmax + 1024, dtype=torch.uint8, device=dev) for _ in range(2)],
                   "d_len": [torch.empty((2, C), dtype=torch.int32, device=dev) for _ in range(2)],
                   "d_off": [torch.empty(C + 1, dtype=torch.int64, device=dev) for _ in range(2)],
                   "h_off": torch.empty(B + 1, dtype=torch.int64, pin_memory=True),
                   "row_off": torch.arange(C, dtype=torch.int64, device=dev) * out_pitch,
                   "d_out": [torch.empty((C, out_pitch), dtype=torch.uint8, device=dev) for _ in range(2)]}
            self._ihost_ctx = ctx
        ctx["h_off"][:B + 1].copy_(h_off)                               # pinned copy of the offsets: the per-chunk H2D reads it asynchronously
        s_in, s_k, s_out = ctx["streams"]
        for st in (s_in, s_k, s_out):
            st.wait_stream(cur)
        ev_k = [None, None]                # inflate of the chunk that last used staging pair j (read d_z / d_off, wrote d_out / d_len)
        ev_out = [None, None]              # its rows are in host memory
        try:
            for k, b0 in enumerate(starts):
                nb = ends[k] - b0
                j = k & 1
                za, zb = zlo[k], zhi[k]
                with torch.cuda.stream(s_in):
                    if ev_k[j] is not None:
                        s_in.wait_event(ev_k[j])
                    ctx["d_z"][j][:zb - za].copy_(h_z[za:zb], non_blocking=True)
                    ctx["d_off"][j][:nb + 1].copy_(ctx["h_off"][b0:b0 + nb + 1], non_blocking=True)
                    ctx["d_off"][j][:nb + 1].sub_(za)                   # offsets relative to the staged piece
                    ev_in = torch.cuda.Event()
                    ev_in.record(s_in)
                with torch.cuda.stream(s_k):
                    s_k.wait_event(ev_in)
                    if ev_out[j] is not None:
                        s_k.wait_event(ev_out[j])
                    _, ol, st = self.inflate_batch(ctx["d_z"][j], in_off=ctx["d_off"][j][:nb + 1], out_pitch=out_pitch, flags=flags,
                                                   obsize=obsize, out=ctx["d_out"][j][:nb])
                    ctx["d_len"][j][0, :nb].copy_(ol)
                    ctx["d_len"][j][1, :nb].copy_(st)
                    ev_k[j] = torch.cuda.Event()
                    ev_k[j].record(s_k)
                with torch.cuda.stream(s_out):
                    s_out.wait_event(ev_k[j])
                    if d2h == "kernel":
                        # rows -> pinned host rows by a kernel (row b of the chunk to h_out[b0 + b]; out_len bytes each)
                        rc = self.lib.hdlz_compact_batch(ctx["d_out"][j].data_ptr(), out_pitch, ctx["d_len"][j][0].data_ptr(),
                                                         ctx["row_off"].data_ptr(), nb, h_out[b0:].data_ptr(), s_out.cuda_stream)
                        self._check(rc, "hdlz_compact_batch (rows -> pinned host memory)")
                    else:
                        for r0 in range(0, nb, piece):                  # copy-engine transfers of at most 32 MiB (see the docstring)
                            r1 = min(nb, r0 + piece)
                            h_out[b0 + r0:b0 + r1].copy_(ctx["d_out"][j][r0:r1], non_blocking=True)
                    h_len[b0:b0 + nb].copy_(ctx["d_len"][j][0, :nb], non_blocking=True)
                    h_status[b0:b0 + nb].copy_(ctx["d_len"][j][1, :nb], non_blocking=True)
                    ev_out[j] = torch.cuda.Event()
                    ev_out[j].record(s_out)
        finally:
            for st_ in (s_out, s_k, s_in):
                cur.wait_stream(st_)
            s_out.synchronize()
            if not keep_buffers:
                self.release_host_buffers()
        return h_out, h_len, h_status

    # -- streaming sessions (hdlz_compress_chunk / hdlz_inflate_chunk): the port adapter's streaming mode
    def compress_session(self, cwindow=32, maxmatch=10):
        return CompressSession(self, cwindow, maxmatch)

    def inflate_session(self, flags=0, obsize=0):
        return InflateSession(self, flags, obsize)

    # -- single-stream conveniences used by the port adapter (one START = one block)
    @_on_device
    def compress_bytes(self, data, cwindow=32, maxmatch=10):
        """-> (status, bytes)"""
        n = len(data)
        pad = (n + 15) // 16 * 16 + 16
        host = torch.zeros(pad, dtype=torch.uint8)
        if n:
            host[:n] = torch.frombuffer(bytearray(data), dtype=torch.uint8)
        d = host.to(self.device)
        if n >= self.STREAM_MIN:
            out, ol, st = self.compress_stream(d, n, cwindow=cwindow, maxmatch=maxmatch)
            return int(st.item()), bytes(out[:int(ol.item())].cpu().numpy().tobytes())
        out, ol, st = self.compress_batch(d.view(1, pad), in_len=n, cwindow=cwindow, maxmatch=maxmatch)
        st = int(st.item())
        return st, bytes(out[0, :int(ol.item())].cpu().numpy().tobytes())

    @_on_device
    def inflate_bytes(self, z, out_cap=None, flags=0, obsize=0):
        """-> (status, bytes).  Default capacity: deflate expands at most 1032:1 (a 258-byte match costs 2 bits), so
        1032 n + 258 bytes hold any stream, capped at the reference's 2^LMAX counter range (deflate.py:73-76)."""
        n = len(z)
        pad = (n + 15) // 16 * 16 + 16
        host = torch.zeros(pad, dtype=torch.uint8)
        if n:
            host[:n] = torch.frombuffer(bytearray(z), dtype=torch.uint8)
        d = host.to(self.device).view(1, pad)
        cap = out_cap if out_cap is not None else min(1 << LMAX, max(1 << 16, 1032 * n + 258))
        cap = (cap + 15) // 16 * 16
        if n >= 5 and (z[2] & 7) == 3:           # the bytes are here: BFINAL = 1, BTYPE = 01 -- a single fixed block takes that chain only
            flags |= INFLATE_ONE_FIXED_BLOCK
        out, ol, st = self.inflate_batch(d, in_len=n, out_pitch=cap, flags=flags, obsize=obsize)
        st = int(st.item())
        return st, bytes(out[0, :int(ol.item())].cpu().numpy().tobytes())


class _Session(object):
    """device-resident buffers of one streaming session: the input seen so far (linear, grows), the output (linear, grows)
    and the kernel's state words; see hdlz_compress_chunk / hdlz_inflate_chunk in include/hdlz.h"""

    def __init__(self, engine, state_words, in_cap=1 << 12, out_cap=1 << 13):
        self.eng, self.lib, self.dev = engine, engine.lib, engine.device
        with torch.cuda.device(self.dev):
            self.d_in = torch.zeros(in_cap, dtype=torch.uint8, device=self.dev)
            self.d_out = torch.zeros(out_cap, dtype=torch.uint8, device=self.dev)
            self.d_state = torch.zeros(state_words, dtype=torch.int32, device=self.dev)
        self.n = 0                    # input bytes on the device

    def write(self, data):
        """append input bytes (host -> device)"""
        k = len(data)
        if not k:
            return
        with torch.cuda.device(self.dev):
            if self.n + k + 64 > self.d_in.numel():
                grown = torch.zeros(max(2 * self.d_in.numel(), self.n + k + 4096), dtype=torch.uint8, device=self.dev)
                grown[:self.n] = self.d_in[:self.n]
                self.d_in = grown
            self.d_in[self.n:self.n + k] = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(self.dev)
        self.n += k

    def _need_out(self, cap):
        if cap > self.d_out.numel():
            with torch.cuda.device(self.dev):
                grown = torch.zeros(max(2 * self.d_out.numel(), cap), dtype=torch.uint8, device=self.dev)
                grown[:self.d_out.numel()] = self.d_out
                self.d_out = grown

    def _state(self):
        return self.d_state.cpu().tolist()           # (the sync point of a step)

    def output(self, a, b):
        """output bytes [a, b) (must be below what the last step reported as produced)"""
        return bytes(self.d_out[a:b].cpu().numpy().tobytes()) if b > a else b""


class CompressSession(_Session):
    """STARTC for a stream that arrives in pieces: write() bytes as they come, step() encodes what can be encoded (the
    reference's rule: a position needs ten known bytes behind it, deflate.py:768-770, unless the input has ended).  Output
    is ONE zlib stream, bit-identical to compress_batch on the whole input."""

    def __init__(self, engine, cwindow=32, maxmatch=10):
        super().__init__(engine, 16)
        self.cwindow, self.maxmatch = cwindow, maxmatch
        self.pos = 0                  # positions [0, pos) are encoded
        self.out_len = 0              # complete output bytes readable
        self.done = False

    def encodable(self, final=False):
        """positions step() could encode now (a multiple of 32 unless final)"""
        if final:
            return self.n - self.pos
        return max(0, (self.n - 11 - self.pos) // 32 * 32)

    def step(self, final=False, max_positions=None):
        """encode up to max_positions (rounded down to a multiple of 32) of the pending positions; with final=True and no
        cap left over, finish the stream.  Returns the status code (OK also when there was nothing to do)."""
        if self.done:
            return OK
        from .constants import E_SHORT_INPUT
        nonfinal = max(0, (self.n - 11 - self.pos) // 32 * 32)
        cap = None if max_positions is None else max_positions // 32 * 32
        if final and self.n < 5:
            return E_SHORT_INPUT                              # R0: the reference never starts
        if final and (max_positions is None or self.n - self.pos <= max_positions):
            k, fin = self.n - self.pos, True
        else:
            k, fin = (nonfinal if cap is None else min(nonfinal, cap)), False
        if k <= 0:
            return OK
        q_end = self.pos + k
        self._need_out(self.lib.hdlz_out_bound(self.n + 64) + 4096)
        with torch.cuda.device(self.dev):
            rc = self.lib.hdlz_compress_chunk(self.d_in.data_ptr(), self.n, q_end, 1 if fin else 0, self.cwindow, self.maxmatch,
                                              self.d_out.data_ptr(), self.d_out.numel(), self.d_state.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream)
        self.eng._check(rc, "hdlz_compress_chunk")
        st = self._state()
        self.pos, self.done, self.out_len = st[0], bool(st[8]), st[9]
        return st[10]


class InflateSession(_Session):
    """STARTD for a stream that arrives in pieces and whose output is drained through a bounded window: step() decodes until
    the stream ends, the input known so far runs out (need == 1) or out_limit bytes are produced (need == 2)."""

    def __init__(self, engine, flags=0, obsize=0, out_cap=1 << 16):
        super().__init__(engine, 16 + 80, out_cap=out_cap)
        self.flags, self.obsize = flags, obsize
        self.out_pos, self.done, self.need = 0, False, 1

    def step(self, final=False, out_limit=None):
        if self.done:
            return OK
        limit = (1 << LMAX) if out_limit is None else out_limit
        self._need_out(min(limit, 1 << LMAX) + 64)
        with torch.cuda.device(self.dev):
            rc = self.lib.hdlz_inflate_chunk(self.d_in.data_ptr(), self.n, 1 if final else 0, self.flags, self.obsize,
                                             self.d_out.data_ptr(), self.d_out.numel(), min(limit, 0xFFFFFFFF),
                                             self.d_state.data_ptr(), torch.cuda.current_stream().cuda_stream)
        self.eng._check(rc, "hdlz_inflate_chunk")
        st = self._state()
        self.out_pos, self.done, self.need = st[1], bool(st[9]), st[11]
        return st[10]
