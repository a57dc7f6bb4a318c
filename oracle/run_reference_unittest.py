#!/usr/bin/env python3
"""Run the reference's OWN, unmodified unit test (/root/reference/test_deflate.py: modes 0..5, streaming
inflate then streaming compress of 10 000 bytes, checked against stock zlib) under oracle/standin/myhdl.py.

CONTAINER-ONLY validation of the stand-in kernel (not a test of this repository's code): the reference
sources are executed by path, nothing is copied.  Prints the unittest verdict and the simulated time."""
import contextlib
import io
import os
import runpy
import sys
import tempfile
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("HDLZ_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(HERE, "standin"))
sys.path.insert(0, REF)
import myhdl  # noqa: E402  (the stand-in)

t0 = time.time()
os.chdir(tempfile.mkdtemp())                       # the reference writes nothing, but stay out of /root/reference
sys.argv = ["test_deflate.py"]
buf = io.StringIO()
code = 0
try:
    with contextlib.redirect_stdout(buf):           # the reference prints on most state transitions
        runpy.run_path(os.path.join(REF, "test_deflate.py"), run_name="__main__")
except SystemExit as e:                             # unittest.main() exits
    code = int(bool(e.code))
out = buf.getvalue()
print("reference UnitTest under the stand-in kernel: %s  (%.0f s wall, simulated time %d, %d lines of reference output)"
      % ("OK" if code == 0 else "FAILED", time.time() - t0, myhdl.now(), out.count("\n")))
for line in out.splitlines():
    if line.startswith(("IN/OUT/CYCLES/WAIT", "Decompress OK", "zlib test")):
        print("  ", line[:110])
sys.exit(code)
