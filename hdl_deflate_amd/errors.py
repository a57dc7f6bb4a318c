class Error(Exception):
    """Raised where the reference raises myhdl.Error out of Simulation.run (deflate.py, 22 sites)
    or would stall forever (N < 5, README:194)."""


class HdlzStatusError(Error):
    def __init__(self, status, what=""):
        from .constants import STATUS_NAMES
        self.status = int(status)
        super().__init__("%s: status %d (%s)" % (what, self.status, STATUS_NAMES.get(self.status, "?")))


class HdlzRangeError(Error, ValueError):
    """a stream does not fit the LMAX-bit address / progress counters (deflate.py:73-76).  The reference raises MyHDL's
    ValueError("intbv value ... out of range") there; longer inputs are chained block by block (hdl_deflate_amd/chain.py)."""
