"""Port-protocol adapter: the reference's command surface on top of the HIP engine.

Mirrors `deflate(i_mode, o_done, i_data, o_iprogress, o_oprogress, o_byte, i_waddr, i_raddr, clk,
reset)` (/root/reference/deflate.py:219-221) -- same port names, same command codes
(IDLE/WRITE/READ/STARTC/STARTD = 0..4, deflate.py:18), one `cycle()` per clock -- so the harness of
test_deflate.py:115-286 (START-then-stream) and the bench of test_deflate.py:513-560 (preload-then-START)
run unchanged against it.  The reference moves one byte per clock through a 28-state FSM; here the bytes go
to the GPU engine in two possible ways:

BATCH mode (default).  WRITEs are buffered, the whole stream is handed to the engine when the caller signals
end of input (first IDLE after START, cf. deflate.py:768-770 / :1529), READs are served from the result.

STREAMING mode (`streaming=True`).  The reference's own mode of operation: bounded
circular memories on both sides and work that overlaps WRITE and READ.
  * input: `iram` is the reference's circular buffer of IBSIZE bytes (`iram[i_waddr & IBS]`, deflate.py:604); every
    clock the engine CONSUMES what was written, up to the reference's own limit of ten bytes behind the writer (copies
    it into its device-resident session), and `o_iprogress` is the last position consumed -- so the harness throttle
    `o_iprogress > i - CWINDOW` (test_deflate.py:159,250) sees the progress it waits for, and a writer that laps the
    ring by more than IBSIZE before a clock edge loses data exactly as on the hardware;
  * processing: whenever a window of `window` consumed-but-unprocessed bytes is filled (and at the end of input) one
    resumable kernel call is launched on it (hdlz_compress_chunk / hdlz_inflate_chunk): compress encodes a position
    only when ten more bytes are known (the reference's stall, deflate.py:768-770), inflate decodes a token only
    when its bits are there (deflate.py:1529-1530, :1600-1602);
  * output: `oram` is the circular buffer of OBSIZE bytes (`o_byte = oram[i_raddr & OBS]`, deflate.py:601) and the
    engine is HELD while it could overwrite unread bytes: nothing is produced beyond `i_raddr + OBSIZE`
    (deflate.py:1531-1534, :1597-1599 for inflate; the reference's compress side has no such hold and simply
    overwrites -- here both directions hold by default, the safe superset; `compress_hold=False` selects the reference's
    overwrite behaviour);
  * `o_oprogress` counts the bytes produced so far, `o_done` rises when the last one is produced.
Addresses and counters are LMAX bits wide (deflate.py:73-76: 24, or 16 in a LOWLUT build): ports wrap like the
reference's modbv signals, and a stream whose input or output does not fit raises HdlzRangeError where MyHDL
raises "intbv value out of range" -- longer inputs are chained block by block (hdl_deflate_amd/chain.py).

Protocol facts honoured in both modes:
  * WRITE stores i_data at i_waddr and sets isize = i_waddr (deflate.py:602-605); the input memory
    persists across runs (the SHORT-INPUT hack of test_deflate.py:239-248 depends on stale bytes 0..3);
  * o_byte mirrors oram[i_raddr] one clock later, whatever i_mode is (deflate.py:601);
  * STARTC/STARTD are honoured in IDLE state only and clear o_done/progress (deflate.py:616-651);
  * o_done rises only when all output is readable; final o_oprogress = output length (deflate.py:814);
  * where the reference raises myhdl.Error (or hangs: N < 5), cycle() raises hdl_deflate_amd.Error.
"""
from .constants import IDLE, WRITE, STARTC, STARTD, OK, CWINDOW, MAXMATCH, LMAX
from .errors import HdlzStatusError, HdlzRangeError


class Sig(object):
    """Minimal stand-in for a MyHDL Signal: `.next` is committed by the adapter's cycle()."""
    __slots__ = ("val", "next")

    def __init__(self, val=0):
        self.val = val
        self.next = val

    def __int__(self):
        return int(self.val)

    __index__ = __int__

    def __bool__(self):
        return bool(self.val)

    def __eq__(self, o):
        return self.val == (o.val if isinstance(o, Sig) else o)

    def __ne__(self, o):
        return not self.__eq__(o)

    def __lt__(self, o):
        return self.val < (o.val if isinstance(o, Sig) else o)

    def __le__(self, o):
        return self.val <= (o.val if isinstance(o, Sig) else o)

    def __gt__(self, o):
        return self.val > (o.val if isinstance(o, Sig) else o)

    def __ge__(self, o):
        return self.val >= (o.val if isinstance(o, Sig) else o)

    def __hash__(self):
        return id(self)

    def __repr__(self):
        return "Sig(%r)" % (self.val,)


class DeflatePort(object):
    """The DUT.  `engine` must provide compress_bytes / inflate_bytes (batch mode) and compress_session /
    inflate_session (streaming mode); the default is the HIP engine."""

    ST_IDLE, ST_COMPRESS, ST_INFLATE = range(3)

    def __init__(self, i_mode, o_done, i_data, o_iprogress, o_oprogress, o_byte, i_waddr, i_raddr,
                 clk=None, reset=None, engine=None, cwindow=CWINDOW, maxmatch=MAXMATCH,
                 inflate_flags=0, obsize=0, stream_obsize=None, lmax=LMAX, streaming=False, ibsize=None, window=None,
                 compress_hold=True):
        self.i_mode, self.o_done, self.i_data = i_mode, o_done, i_data
        self.o_iprogress, self.o_oprogress, self.o_byte = o_iprogress, o_oprogress, o_byte
        self.i_waddr, self.i_raddr, self.clk, self.reset = i_waddr, i_raddr, clk, reset
        if engine is None:
            from .engine import Engine
            engine = Engine()
        self.engine = engine
        self.cwindow, self.maxmatch = cwindow, maxmatch
        self.inflate_flags, self.obsize = inflate_flags, obsize
        if streaming and stream_obsize is None:
            stream_obsize = 512                                      # the reference's default OBSIZE (deflate.py:62)
        if stream_obsize is not None and (stream_obsize < 64 or stream_obsize & (stream_obsize - 1)):
            raise ValueError("stream_obsize must be a power of two >= 64")
        self.stream_obsize = stream_obsize
        # streaming STARTC and a reader that lags: True (default) = the engine is held like the inflate side, the reader always gets
        # the stream; False = the reference's behaviour -- put / do_flush write oram[do & OBS] unconditionally (deflate.py:535-567,
        # no counterpart of the inflate hold deflate.py:1531-1534), so a reader more than OBSIZE behind reads overwritten bytes
        # (fixture lagging_reader of tests/golden/streaming_r3_vectors.json: 654 bytes ahead of a 512-byte memory, not a stream)
        self.compress_hold = bool(compress_hold)
        self.lmax = lmax
        self.mask = (1 << lmax) - 1
        self.streaming = bool(streaming)
        # the reference's IBSIZE: 16 * CWINDOW in a FAST build (deflate.py:64-68)
        self.ibsize = ibsize if ibsize is not None else 16 * cwindow
        if self.ibsize & (self.ibsize - 1) or self.ibsize < 32:
            raise ValueError("ibsize must be a power of two >= 32")
        self.window = window if window is not None else max(32, self.ibsize // 2)
        self.pending_done = False
        self.iram = bytearray(self.ibsize) if self.streaming else bytearray()
        self.oram = b""
        self.isize = 0
        self.state = self.ST_IDLE
        self.cycles = 0
        self.launches = 0            # kernel calls of the current / last run (streaming mode)
        self.session = None
        self.consumed = 0
        self.n_at_step = 0
        self.ended = False

    # -- one clock: commit the caller's .next values, then act on them
    def cycle(self):
        for s in (self.i_mode, self.i_data, self.i_waddr, self.i_raddr, self.reset):
            if s is not None:
                s.val = s.next
        self.cycles += 1
        if self.reset is not None and self.reset.val:
            self.state = self.ST_IDLE
            self.session = None
            self._set(self.o_done, False)
            return
        mode = int(self.i_mode.val)
        ra = int(self.i_raddr.val) & self.mask
        self._drive_o_byte(ra)
        if self.streaming:
            return self._cycle_streaming(mode, ra)
        if self.stream_obsize is not None and self.pending_done:
            # release output with the reference's hold `do >= i_raddr + OBSIZE` (deflate.py:1531-1534)
            vis = min(len(self.oram), ra + self.stream_obsize)
            if vis > int(self.o_oprogress.val):
                self._set(self.o_oprogress, vis)
            if int(self.o_oprogress.val) == len(self.oram):
                self.pending_done = False
                self._set(self.o_done, True)
        if mode == WRITE:
            wa = int(self.i_waddr.val) & self.mask
            if wa >= len(self.iram):
                self.iram.extend(bytes(wa + 1 - len(self.iram)))
            self.iram[wa] = int(self.i_data.val) & 0xFF
            self.isize = wa
            if self.state != self.ST_IDLE:
                self._set(self.o_iprogress, wa)
        # logic (deflate.py:607-1664), collapsed
        if self.state == self.ST_IDLE:
            if mode == STARTC or mode == STARTD:
                self._start(mode)
        elif mode == IDLE:
            self._run()

    # -- io_logic's read side (deflate.py:601): o_byte = oram[i_raddr & OBS], one clock later
    def _drive_o_byte(self, ra):
        if self.stream_obsize is None:
            self._set(self.o_byte, self.oram[ra] if ra < len(self.oram) else 0)
            return
        # circular output memory: the newest produced byte whose address is congruent to i_raddr
        N = self.stream_obsize
        vis = int(self.o_oprogress.val)
        r = ra & (N - 1)
        p = r + ((vis - 1 - r) // N) * N if vis > r else -1
        self._set(self.o_byte, self.oram[p] if 0 <= p < len(self.oram) else 0)

    def _start(self, mode):
        self.state = self.ST_COMPRESS if mode == STARTC else self.ST_INFLATE
        self._set(self.o_done, False)
        self._set(self.o_iprogress, 0)
        self._set(self.o_oprogress, 0)
        self.oram = b""
        self.pending_done = False
        self.launches = 0

    # ------------------------------------------------------------------------------------------ batch mode
    def _run(self):
        n = self.isize + 1                      # R0: isize = last written address
        data = bytes(self.iram[:n]) + bytes(max(0, n - len(self.iram)))
        if self.state == self.ST_COMPRESS:
            st, res = self.engine.compress_bytes(data, cwindow=self.cwindow, maxmatch=self.maxmatch)
            what = "STARTC"
        else:
            # a stream may expand 1032:1; the counters are LMAX bits wide (deflate.py:73-76)
            st, res = self.engine.inflate_bytes(data, flags=self.inflate_flags, obsize=self.obsize,
                                                out_cap=min(1 << self.lmax, max(1 << 16, 1032 * n + 258)))
            what = "STARTD"
            if st == 2 and 1032 * n + 258 > (1 << self.lmax):      # E_OUT_CAPACITY at the counter range
                self.state = self.ST_IDLE
                raise HdlzRangeError("STARTD: output does not fit the %d-bit progress counters (deflate.py:73-76)" % self.lmax)
        self.state = self.ST_IDLE
        if st != OK:
            # the reference raises myhdl.Error from inside Simulation.run (or never finishes)
            self._set(self.o_done, True)
            raise HdlzStatusError(st, what)
        if len(res) > self.mask:
            raise HdlzRangeError("%s: %d output bytes do not fit the %d-bit progress counters (deflate.py:73-76)"
                                 % (what, len(res), self.lmax))
        self.oram = res
        self._set(self.o_iprogress, self.isize)
        if self.stream_obsize is None:
            self._set(self.o_oprogress, len(res))
            self._set(self.o_done, True)
        else:
            self._set(self.o_oprogress, min(len(res), (int(self.i_raddr.val) & self.mask) + self.stream_obsize))
            self.pending_done = int(self.o_oprogress.val) < len(res)
            if not self.pending_done:
                self._set(self.o_done, True)

    # ------------------------------------------------------------------------------------------ streaming mode
    def _cycle_streaming(self, mode, ra):
        if mode == WRITE:
            wa = int(self.i_waddr.val) & self.mask
            self.iram[wa & (self.ibsize - 1)] = int(self.i_data.val) & 0xFF      # deflate.py:604
            self.isize = wa                                                      # deflate.py:605
        if self.state == self.ST_IDLE:
            if mode == STARTC or mode == STARTD:
                self._start(mode)
                self.consumed = 0
                self.n_at_step = 0
                if mode == STARTC and self.stream_obsize < 128:
                    # progress needs room for either a non-final call (32 positions: >= 43 pending bytes) or the final one
                    # (<= 42 pending: ceil(9*42/8) + 8 + 11 = 67 bytes): 64 bytes of output memory can dead-lock
                    self.state = self.ST_IDLE
                    raise ValueError("streaming STARTC needs stream_obsize >= 128")
                elif mode == STARTC:
                    self.session = self.engine.compress_session(cwindow=self.cwindow, maxmatch=self.maxmatch)
                elif self.stream_obsize < 512:
                    # a copy is only started when all of it fits (deflate.py:1597: `do + length >= i_raddr + OBSIZE` holds), so
                    # a 258-byte match can never start in a smaller buffer: 512 is the reference's minimum (deflate.py:62)
                    self.state = self.ST_IDLE
                    raise ValueError("streaming STARTD needs stream_obsize >= 512 (the reference's minimal OBSIZE)")
                else:
                    self.session = self.engine.inflate_session(flags=self.inflate_flags, obsize=self.obsize)
                self.ended = False
            return
        s = self.session
        if mode == IDLE:
            self.ended = True                           # end of input (deflate.py:768-770 / :1529: the stall ends at IDLE)
        # consume: bytes move from the ring into the engine's session -- while the input is still coming only the positions
        # the reference itself may touch (di < isize - 10, deflate.py:768-770: the last ten written bytes can still change,
        # e.g. address 0 is written twice by the harness: "CLEAR OLD INPUT", then the first byte), everything at the end
        hi = self.isize + 1 if self.ended else self.isize - 10
        if hi > self.consumed:
            lo = self.consumed
            if hi - lo > self.ibsize:                   # the writer lapped the ring: those bytes are gone, as on the hardware
                lo = hi - self.ibsize
                s.write(bytes(lo - self.consumed))
            ib = self.ibsize - 1
            s.write(bytes(self.iram[p & ib] for p in range(lo, hi)))
            self.consumed = hi
            self._set(self.o_iprogress, hi - 1)
        room = ra + self.stream_obsize - len(self.oram)  # bytes that may still be produced: the hold of deflate.py:1531-1534
        if self.state == self.ST_COMPRESS:
            self._step_compress(s, room if self.compress_hold else 1 << 30)
        else:
            self._step_inflate(s, ra, room)

    def _publish(self, produced, done, what):
        """new output bytes [len(oram), produced) become readable"""
        if produced > self.mask:
            self.state = self.ST_IDLE
            raise HdlzRangeError("%s: output does not fit the %d-bit progress counters (deflate.py:73-76)" % (what, self.lmax))
        if produced > len(self.oram):
            self.oram += self.session.output(len(self.oram), produced)
            self._set(self.o_oprogress, produced)
        if done:
            self.state = self.ST_IDLE
            self._set(self.o_done, True)

    def _step_compress(self, s, room):
        # a call over k positions writes at most ceil(9k/8) + 8 bytes (+ 11 for EOB, padding and the trailer at the end)
        bound = (room - 24) * 8 // 9 if room > 24 else 0     # positions whose output certainly fits the room (unrounded)
        fit = bound // 32 * 32                               # ... as a non-final call may take them (whole lanes)
        pending = s.n - s.pos
        if self.ended:
            if s.n < 5:                                  # R0: the reference never starts (deflate.py:429-431) -- it hangs; we say so
                self.state = self.ST_IDLE
                self._set(self.o_done, True)
                raise HdlzStatusError(1, "STARTC")
            if pending <= bound:
                st = s.step(final=True)                  # everything that is left, EOB and the trailer fit
            elif fit >= 32 and s.encodable() >= 32:
                st = s.step(max_positions=fit)           # a non-final piece; the rest when the reader has made room
            else:
                return                                   # HOLD: the reader must advance first
        else:
            if pending - 11 < self.window or fit < 32:
                return                                   # window not filled yet / held by the reader
            st = s.step(max_positions=min(fit, self.window))
        self.launches += 1
        if st != OK:
            self.state = self.ST_IDLE
            self._set(self.o_done, True)
            raise HdlzStatusError(st, "STARTC")
        self._publish(s.out_len, s.done, "STARTC")

    def _step_inflate(self, s, ra, room):
        if room <= 0:
            return                                       # HOLD: the reader must advance first (deflate.py:1531-1534)
        if s.need == 2:
            # the engine stopped for output room: wake it when half the buffer is free again, or everything is read
            if room < self.stream_obsize // 2 and ra < len(self.oram):
                return
        elif not self.ended and s.n - self.n_at_step < self.window:
            return                                       # it stopped for input and the next window is not filled yet
        self.n_at_step = s.n
        st = s.step(final=self.ended, out_limit=ra + self.stream_obsize)
        self.launches += 1
        if st != OK:
            self.state = self.ST_IDLE
            self._set(self.o_done, True)
            raise HdlzStatusError(st, "STARTD")
        self._publish(s.out_pos, s.done, "STARTD")

    @staticmethod
    def _set(sig, v):
        sig.val = v
        sig.next = v


def deflate(i_mode, o_done, i_data, o_iprogress, o_oprogress, o_byte, i_waddr, i_raddr, clk=None, reset=None,
            **kw):
    """Factory with the reference's signature (deflate.py:219-221).  Signals are `Sig` objects."""
    return DeflatePort(i_mode, o_done, i_data, o_iprogress, o_oprogress, o_byte, i_waddr, i_raddr, clk, reset,
                       **kw)
