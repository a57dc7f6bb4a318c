class Error(Exception):
    """Raised where the reference raises myhdl.Error out of Simulation.run (deflate.py, 22 sites)
    or would stall forever (N < 5, README:194)."""


class HdlzStatusError(Error):
    def __init__(self, status, what=""):
        from .constants import STATUS_NAMES
        self.status = int(status)
        super().__init__("%s: status %d (%s)" % (what, self.status, STATUS_NAMES.get(self.status, "?")))
