"""Clocked-only stand-in for the subset of the MyHDL API that
/root/reference/deflate.py and /root/reference/test_deflate.py use.

TEST INFRASTRUCTURE ONLY (container-only).  MyHDL is not installed in the build
image and cannot be installed (no network), so the reference's Python sources
cannot be imported as they are.  This module is our own code (it is NOT MyHDL
and shares no source with it); it exists so that oracle/gen_golden.py can
execute the UNMODIFIED reference sources *by path* and record golden vectors.
Nothing in the product, in tests/ or on the GPU box imports it.

Semantics implemented (enough for a single-clock synchronous design):
  * Signal.next assignments are queued and committed together at the end of a
    delta; a rising edge on a signal runs every @always(sig.posedge) body once.
  * @always_comb bodies are re-run after every delta in which anything changed
    until a fixpoint is reached (they are pure functions of their inputs, so
    over-triggering is harmless).
  * ConcatSignal is evaluated on read.
  * intbv bounds are enforced on assignment, modbv wraps.
Caveat recorded next to every fixture generated with it: "reference source
executed under a clocked-only stand-in kernel, not under MyHDL 0.10".
"""
import sys

__all__ = ["always", "always_comb", "always_seq", "block", "Signal", "intbv", "modbv",
           "Error", "ResetSignal", "enum", "concat", "ConcatSignal", "instances",
           "instance", "delay", "now", "Simulation", "Cosimulation", "StopSimulation"]


class Error(Exception):
    pass


class StopSimulation(Exception):
    pass


# --------------------------------------------------------------------------- values
class intbv(object):
    _wrap = False

    def __init__(self, val=0, min=None, max=None, _nrbits=0):
        if isinstance(val, intbv):
            val = val._val
        self._val = int(val)
        self._min = min
        self._max = max
        self._nrbits = _nrbits
        if min is not None and max is not None and not _nrbits:
            n = int(max - 1).bit_length() if max > 0 else 0
            if min < 0:
                n = (n if n > int(-min - 1).bit_length() else int(-min - 1).bit_length()) + 1
            self._nrbits = n

    def __getitem__(self, key):
        if isinstance(key, slice):
            hi = key.start
            lo = key.stop if key.stop is not None else 0
            if hi is None:
                raise ValueError("open upper slice")
            n = hi - lo
            r = self.__class__((self._val >> lo) & ((1 << n) - 1))
            r._nrbits = n
            r._min = 0
            r._max = 1 << n
            return r
        return (self._val >> int(key)) & 1

    def __int__(self):
        return self._val

    __index__ = __int__

    def __len__(self):
        return self._nrbits


class modbv(intbv):
    _wrap = True


class _EnumItem(object):
    __slots__ = ("_name", "_index", "_type")

    def __init__(self, name, index, typ):
        self._name, self._index, self._type = name, index, typ

    def __repr__(self):
        return self._name

    def __int__(self):
        return self._index

    __index__ = __int__

    def __hash__(self):
        return hash((id(self._type), self._index))

    def __eq__(self, other):
        if isinstance(other, Signal):
            other = other._val
        return self is other

    def __ne__(self, other):
        return not self.__eq__(other)


class _EnumType(object):
    def __init__(self, names):
        self._names = names
        for i, n in enumerate(names):
            setattr(self, n, _EnumItem(n, i, self))

    def __len__(self):
        return len(self._names)


def enum(*names, **kwargs):
    return _EnumType(names)


# --------------------------------------------------------------------------- kernel
class _Kernel(object):
    def __init__(self):
        self.pending = []
        self.time = 0

    def reset(self):
        self.pending = []
        self.time = 0


_K = _Kernel()


def now():
    return _K.time


class _Edge(object):
    __slots__ = ("sig", "rising")

    def __init__(self, sig, rising):
        self.sig, self.rising = sig, rising


def _v(x):
    """plain python value of a Signal / intbv / ConcatSignal / int."""
    if isinstance(x, Signal):
        return x._val
    if isinstance(x, intbv):
        return x._val
    if isinstance(x, ConcatSignal):
        return x.val
    return x


class Signal(object):
    __slots__ = ("_val", "_next", "_dirty", "_nrbits", "_min", "_max", "_wrap", "_kind",
                 "posedge", "negedge", "_used", "_rose", "_fell")

    def __init__(self, val=None):
        self._dirty = False
        self._used = False
        self._rose = False
        self._fell = False
        self._min = None
        self._max = None
        self._nrbits = 0
        self._wrap = False
        if isinstance(val, intbv):
            self._kind = 1
            self._val = val._val
            self._nrbits = val._nrbits
            self._wrap = val._wrap
            if val._min is not None:
                self._min, self._max = val._min, val._max
            elif val._nrbits:
                self._min, self._max = 0, 1 << val._nrbits
        elif isinstance(val, bool):
            self._kind = 0
            self._val = val
            self._nrbits = 1
        elif isinstance(val, _EnumItem):
            self._kind = 2
            self._val = val
        elif val is None:
            self._kind = 1
            self._val = 0
        else:
            self._kind = 1
            self._val = int(val)
        self._next = self._val
        self.posedge = _Edge(self, True)
        self.negedge = _Edge(self, False)

    # -- next / val
    @property
    def next(self):
        return self._next

    @next.setter
    def next(self, v):
        v = _v(v)
        k = self._kind
        if k == 1:
            v = int(v)
            if self._wrap:
                if self._nrbits:
                    v &= (1 << self._nrbits) - 1
            elif self._max is not None:
                if v < self._min or v >= self._max:
                    raise ValueError("intbv value %d out of range [%d,%d)" % (v, self._min, self._max))
        elif k == 0:
            if v not in (0, 1):
                raise ValueError("bool signal assigned %r" % (v,))
            v = bool(v)
        self._next = v
        if not self._dirty:
            self._dirty = True
            _K.pending.append(self)

    @property
    def val(self):
        return self._val

    def _markUsed(self):
        self._used = True

    # -- conversions
    def __bool__(self):
        return bool(self._val)

    def __int__(self):
        return int(self._val)

    __index__ = __int__

    def __len__(self):
        return self._nrbits

    def __hash__(self):
        return id(self)

    def __repr__(self):
        return repr(self._val)

    __str__ = __repr__

    def __format__(self, spec):
        return format(self._val, spec)

    def __getitem__(self, key):
        if isinstance(key, slice):
            hi = key.start
            lo = key.stop if key.stop is not None else 0
            return (self._val >> lo) & ((1 << (hi - lo)) - 1)
        return (self._val >> int(key)) & 1

    # -- comparisons
    def __eq__(self, o):
        return self._val == _v(o)

    def __ne__(self, o):
        return self._val != _v(o)

    def __lt__(self, o):
        return self._val < _v(o)

    def __le__(self, o):
        return self._val <= _v(o)

    def __gt__(self, o):
        return self._val > _v(o)

    def __ge__(self, o):
        return self._val >= _v(o)

    # -- arithmetic (results are plain ints)
    def __add__(self, o):
        return self._val + _v(o)

    __radd__ = __add__

    def __sub__(self, o):
        return self._val - _v(o)

    def __rsub__(self, o):
        return _v(o) - self._val

    def __mul__(self, o):
        return self._val * _v(o)

    __rmul__ = __mul__

    def __floordiv__(self, o):
        return self._val // _v(o)

    def __rfloordiv__(self, o):
        return _v(o) // self._val

    def __mod__(self, o):
        return self._val % _v(o)

    def __rmod__(self, o):
        return _v(o) % self._val

    def __and__(self, o):
        return self._val & _v(o)

    __rand__ = __and__

    def __or__(self, o):
        return self._val | _v(o)

    __ror__ = __or__

    def __xor__(self, o):
        return self._val ^ _v(o)

    __rxor__ = __xor__

    def __lshift__(self, o):
        return self._val << _v(o)

    def __rlshift__(self, o):
        return _v(o) << self._val

    def __rshift__(self, o):
        return self._val >> _v(o)

    def __rrshift__(self, o):
        return _v(o) >> self._val

    def __neg__(self):
        return -self._val

    def __invert__(self):
        return ~self._val

    def __abs__(self):
        return abs(self._val)


def ResetSignal(val, active, isasync=True, **kw):
    s = Signal(bool(val))
    return s


class ConcatSignal(object):
    """MSB-first concatenation evaluated on read."""

    def __init__(self, *args):
        self._args = args
        self._nrbits = sum(a._nrbits for a in args)

    def _markUsed(self):
        pass

    @property
    def val(self):
        r = 0
        for a in self._args:
            r = (r << a._nrbits) | int(a._val)
        return r

    def __int__(self):
        return self.val

    __index__ = __int__

    def __len__(self):
        return self._nrbits

    def __rshift__(self, o):
        return self.val >> _v(o)

    def __lshift__(self, o):
        return self.val << _v(o)

    def __and__(self, o):
        return self.val & _v(o)

    __rand__ = __and__

    def __or__(self, o):
        return self.val | _v(o)

    __ror__ = __or__

    def __eq__(self, o):
        return self.val == _v(o)

    def __hash__(self):
        return id(self)

    def __repr__(self):
        return repr(self.val)


def concat(*args):
    """MSB-first concatenation of sized operands -> plain int."""
    r = 0
    for a in args:
        n = a._nrbits if not isinstance(a, (bool, int)) else 1
        if not n:
            raise ValueError("concat of unsized operand")
        r = (r << n) | int(_v(a))
    return r


# --------------------------------------------------------------------------- processes
class _Proc(object):
    __slots__ = ("kind", "func", "edge", "gen")

    def __init__(self, kind, func, edge=None):
        self.kind, self.func, self.edge, self.gen = kind, func, edge, None


def always(edge):
    def deco(f):
        return _Proc("edge", f, edge)
    return deco


def always_seq(edge, reset=None):
    def deco(f):
        return _Proc("edge", f, edge)
    return deco


def always_comb(f):
    return _Proc("comb", f)


def instance(f):
    return _Proc("gen", f)


class _Block(object):
    def __init__(self, subs):
        self.subs = subs

    def convert(self, *a, **kw):       # HDL conversion is out of scope: no-op
        return None

    def config_sim(self, *a, **kw):
        return None


def _flatten(x, out):
    if isinstance(x, _Proc):
        out.append(x)
    elif isinstance(x, _Block):
        _flatten(x.subs, out)
    elif isinstance(x, (list, tuple)):
        for y in x:
            _flatten(y, out)


def block(f):
    def wrapper(*a, **kw):
        return _Block(f(*a, **kw))
    wrapper.__name__ = getattr(f, "__name__", "block")
    return wrapper


def instances():
    loc = sys._getframe(1).f_locals
    out = []
    for v in loc.values():
        if isinstance(v, (_Proc, _Block)):
            out.append(v)
        elif isinstance(v, (list, tuple)) and v and isinstance(v[0], (_Proc, _Block)):
            out.append(v)
    return out


class delay(object):
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t


def Cosimulation(*a, **kw):
    raise Error("Cosimulation is not available in the stand-in kernel")


# --------------------------------------------------------------------------- engine
class Design(object):
    """Flattened set of processes plus the delta-cycle engine."""

    def __init__(self, *tops):
        procs = []
        _flatten(list(tops), procs)
        self.edge = {}
        self.combs = []
        self.gens = []
        for p in procs:
            if p.kind == "edge":
                self.edge.setdefault((id(p.edge.sig), p.edge.rising), []).append(p.func)
            elif p.kind == "comb":
                self.combs.append(p.func)
            else:
                self.gens.append(p)
        self.settle(force=True)

    def commit(self):
        """apply queued .next values; return list of changed signals."""
        pend = _K.pending
        _K.pending = []
        changed = []
        for s in pend:
            s._dirty = False
            n = s._next
            if n != s._val:
                if s._kind == 0:
                    s._rose = bool(n) and not s._val
                    s._fell = (not n) and bool(s._val)
                s._val = n
                changed.append(s)
        return changed

    def settle(self, force=False):
        """run delta cycles until nothing is pending."""
        edge = self.edge
        combs = self.combs
        first = force
        while True:
            changed = self.commit()
            if not changed and not first:
                return
            first = False
            for s in changed:
                if s._kind == 0:
                    if s._rose:
                        s._rose = False
                        for f in edge.get((id(s), True), ()):
                            f()
                    elif s._fell:
                        s._fell = False
                        for f in edge.get((id(s), False), ()):
                            f()
            # a delta in which only clock-like signals (those with edge listeners) toggled cannot
            # change any @always_comb input of a synchronous design: skip the comb sweep then.
            for s in changed:
                if (id(s), True) not in edge and (id(s), False) not in edge:
                    for f in combs:
                        f()
                    break

    def cycle(self, clk):
        """one full clock period: rising edge (with whatever .next the caller queued), then falling."""
        clk.next = True
        self.settle()
        clk.next = False
        self.settle()


class Simulation(object):
    """Generator-based driver sufficient for test_deflate.py (yield delay(n) only)."""

    def __init__(self, *args):
        self.design = Design(*[a for a in args if not _isgen(a)])
        self.threads = [a for a in args if _isgen(a)]
        for p in self.design.gens:
            self.threads.append(p.func())

    def run(self, duration=None, quiet=0):
        d = self.design
        wake = [(0, i) for i in range(len(self.threads))]
        alive = set(range(len(self.threads)))
        try:
            while alive:
                t = min(w[0] for w in wake if w[1] in alive)
                _K.time = t
                nxt = []
                for (wt, i) in wake:
                    if i not in alive:
                        continue
                    if wt != t:
                        nxt.append((wt, i))
                        continue
                    try:
                        y = next(self.threads[i])
                    except StopIteration:
                        alive.discard(i)
                        continue
                    if not isinstance(y, delay):
                        raise Error("stand-in Simulation only supports 'yield delay(n)'")
                    nxt.append((t + y.t, i))
                wake = nxt
                d.settle()
                if duration is not None and _K.time >= duration:
                    break
        except StopSimulation:
            pass
        return 0


def _isgen(x):
    return hasattr(x, "__next__") and hasattr(x, "send")
