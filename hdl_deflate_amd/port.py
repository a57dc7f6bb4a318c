"""Port-protocol adapter: the reference's command surface on top of the batch engine.

Mirrors `deflate(i_mode, o_done, i_data, o_iprogress, o_oprogress, o_byte, i_waddr, i_raddr, clk,
reset)` (/root/reference/deflate.py:219-221) -- same port names, same command codes
(IDLE/WRITE/READ/STARTC/STARTD = 0..4, deflate.py:18), one `cycle()` per clock -- so the harness of
test_deflate.py:115-286 (START-then-stream) and the bench of test_deflate.py:513-560 (preload-then-START)
run unchanged against it.  What differs by design: the reference moves one byte per clock through a
28-state FSM; here WRITEs are buffered, the whole stream is handed to the HIP engine when the caller
signals end of input (first IDLE after START, cf. deflate.py:768-770 / :1529), and READs are served
from the result.

Protocol facts honoured:
  * WRITE stores i_data at i_waddr and sets isize = i_waddr (deflate.py:602-605); the input memory
    persists across runs (the SHORT-INPUT hack of test_deflate.py:239-248 depends on stale bytes 0..3);
  * o_byte mirrors oram[i_raddr] one clock later, whatever i_mode is (deflate.py:601);
  * STARTC/STARTD are honoured in IDLE state only and clear o_done/progress (deflate.py:616-651);
  * o_iprogress follows accepted input so the harness throttle `o_iprogress > i - CWINDOW`
    (test_deflate.py:159,250) never dead-locks;
  * o_done rises only when all output is readable; final o_oprogress = output length (deflate.py:814);
  * where the reference raises myhdl.Error (or hangs: N < 5), cycle() raises hdl_deflate_amd.Error.

Streaming mode: with `obsize=N` (a power of two, the reference's OBSIZE,
deflate.py:61-62) the output memory is the reference's circular buffer of N bytes -- `o_byte` returns
oram[i_raddr & (N-1)] (deflate.py:601) -- and output is released with the reference's back-pressure:
the engine's result becomes visible only up to `i_raddr + N` (the reference holds while
`do >= i_raddr + OBSIZE`, deflate.py:1531-1534, :1597-1599), so a harness that reads too slowly sees
exactly the stall it would see on the hardware, and `o_done` rises only when the last byte is released.
"""
from .constants import (IDLE, WRITE, READ, STARTC, STARTD, OK, CWINDOW, MAXMATCH, LMAX, STATUS_NAMES)
from .errors import Error, HdlzStatusError


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
    """The DUT.  `engine` must provide compress_bytes(data, cwindow, maxmatch) -> (status, bytes) and
    inflate_bytes(z, flags=, obsize=) -> (status, bytes); the default is the HIP engine."""

    ST_IDLE, ST_COMPRESS, ST_INFLATE = range(3)

    def __init__(self, i_mode, o_done, i_data, o_iprogress, o_oprogress, o_byte, i_waddr, i_raddr,
                 clk=None, reset=None, engine=None, cwindow=CWINDOW, maxmatch=MAXMATCH,
                 inflate_flags=0, obsize=0, stream_obsize=None):
        self.i_mode, self.o_done, self.i_data = i_mode, o_done, i_data
        self.o_iprogress, self.o_oprogress, self.o_byte = o_iprogress, o_oprogress, o_byte
        self.i_waddr, self.i_raddr, self.clk, self.reset = i_waddr, i_raddr, clk, reset
        if engine is None:
            from .engine import Engine
            engine = Engine()
        self.engine = engine
        self.cwindow, self.maxmatch = cwindow, maxmatch
        self.inflate_flags, self.obsize = inflate_flags, obsize
        if stream_obsize is not None and (stream_obsize < 64 or stream_obsize & (stream_obsize - 1)):
            raise ValueError("stream_obsize must be a power of two >= 64")
        self.stream_obsize = stream_obsize
        self.pending_done = False
        self.iram = bytearray()
        self.oram = b""
        self.isize = 0
        self.state = self.ST_IDLE
        self.cycles = 0

    # -- one clock: commit the caller's .next values, then act on them
    def cycle(self):
        for s in (self.i_mode, self.i_data, self.i_waddr, self.i_raddr, self.reset):
            if s is not None:
                s.val = s.next
        self.cycles += 1
        mask = (1 << LMAX) - 1
        if self.reset is not None and self.reset.val:
            self.state = self.ST_IDLE
            self._set(self.o_done, False)
            return
        mode = int(self.i_mode.val)
        # io_logic (deflate.py:599-605)
        ra = int(self.i_raddr.val) & mask
        if self.stream_obsize is None:
            self._set(self.o_byte, self.oram[ra] if ra < len(self.oram) else 0)
        else:
            # circular output memory: the newest released byte whose address is congruent to i_raddr
            N = self.stream_obsize
            vis = int(self.o_oprogress.val)
            p = (ra & (N - 1)) + ((vis - 1 - (ra & (N - 1))) // N) * N if vis > (ra & (N - 1)) else -1
            self._set(self.o_byte, self.oram[p] if 0 <= p < len(self.oram) else 0)
            if self.pending_done:
                # release output with the reference's hold `do >= i_raddr + OBSIZE` (deflate.py:1531-1534)
                vis = min(len(self.oram), ra + N)
                if vis > int(self.o_oprogress.val):
                    self._set(self.o_oprogress, vis)
                if int(self.o_oprogress.val) == len(self.oram):
                    self.pending_done = False
                    self._set(self.o_done, True)
        if mode == WRITE:
            wa = int(self.i_waddr.val) & mask
            if wa >= len(self.iram):
                self.iram.extend(bytes(wa + 1 - len(self.iram)))
            self.iram[wa] = int(self.i_data.val) & 0xFF
            self.isize = wa
            if self.state != self.ST_IDLE:
                self._set(self.o_iprogress, wa)
        # logic (deflate.py:607-1664), collapsed
        if self.state == self.ST_IDLE:
            if mode == STARTC or mode == STARTD:
                self.state = self.ST_COMPRESS if mode == STARTC else self.ST_INFLATE
                self._set(self.o_done, False)
                self._set(self.o_iprogress, 0)
                self._set(self.o_oprogress, 0)
                self.oram = b""
                self.pending_done = False
        elif mode == IDLE:
            self._run()

    def _run(self):
        n = self.isize + 1                      # R0: isize = last written address
        data = bytes(self.iram[:n]) + bytes(max(0, n - len(self.iram)))
        if self.state == self.ST_COMPRESS:
            st, res = self.engine.compress_bytes(data, cwindow=self.cwindow, maxmatch=self.maxmatch)
            what = "STARTC"
        else:
            st, res = self.engine.inflate_bytes(data, flags=self.inflate_flags, obsize=self.obsize)
            what = "STARTD"
        self.state = self.ST_IDLE
        if st != OK:
            # the reference raises myhdl.Error from inside Simulation.run (or never finishes)
            self._set(self.o_done, True)
            raise HdlzStatusError(st, what)
        self.oram = res
        self._set(self.o_iprogress, self.isize)
        if self.stream_obsize is None:
            self._set(self.o_oprogress, len(res))
            self._set(self.o_done, True)
        else:
            self._set(self.o_oprogress, min(len(res), (int(self.i_raddr.val) & ((1 << LMAX) - 1)) + self.stream_obsize))
            self.pending_done = int(self.o_oprogress.val) < len(res)
            if not self.pending_done:
                self._set(self.o_done, True)

    @staticmethod
    def _set(sig, v):
        sig.val = v
        sig.next = v


def deflate(i_mode, o_done, i_data, o_iprogress, o_oprogress, o_byte, i_waddr, i_raddr, clk=None, reset=None,
            **kw):
    """Factory with the reference's signature (deflate.py:219-221).  Signals are `Sig` objects."""
    return DeflatePort(i_mode, o_done, i_data, o_iprogress, o_oprogress, o_byte, i_waddr, i_raddr, clk, reset,
                       **kw)
