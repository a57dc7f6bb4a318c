import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLD = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """a fresh checkout has no built library: build it once (hipcc cross-compiles gfx950 without a GPU, ~1 min), the way
    __graft_entry__.build() does, so that the suite does not depend on having been preceded by a build step"""
    lib = os.path.join(REPO, "hdl_deflate_amd", "lib", "libhdlz.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        import subprocess
        subprocess.check_call(["bash", os.path.join(REPO, "hdl_deflate_amd", "csrc", "build.sh")], stdout=subprocess.DEVNULL)


def load_golden(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def large_vectors():
    """tests/golden/compress_large_vectors.json (oracle/gen_golden_large.py) decoded: [(vector dict, input, output)]"""
    import base64
    import zlib
    g = load_golden("compress_large_vectors.json")
    return [(v, zlib.decompress(base64.b64decode(v["in_b64z"])), base64.b64decode(v["out_b64"])) for v in g["vectors"]]


@pytest.fixture(scope="session")
def oracle():
    """the CPU oracle (checker only; see oracle/hdlz_oracle.c header)"""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def engine():
    """the HIP engine through the C-ABI; fails loudly when the extension or the GPU is missing"""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import hdl_deflate_amd
    return hdl_deflate_amd.Engine()
