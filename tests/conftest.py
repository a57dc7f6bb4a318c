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
    forced = os.path.join(REPO, "hdl_deflate_amd", "lib", "libhdlz_forced.so")      # (tests/test_gpu_forced_paths.py)
    if not os.path.exists(forced) and os.path.exists("/opt/rocm/bin/hipcc"):
        import subprocess
        subprocess.check_call(["bash", os.path.join(REPO, "hdl_deflate_amd", "csrc", "build.sh"), "forced"], stdout=subprocess.DEVNULL)


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


def empty_distance_stream():
    """a hand-built zlib stream of b"aaaaa": ONE dynamic block whose HDIST code lengths are all zero (literals only)"""
    import zlib
    # hand-built: BFINAL=1 BTYPE=2, HLIT=257 (0), HDIST=1 (0), HCLEN=19 (15); code-length code: symbols 0,1,2 -> lengths
    # chosen so that literal 'a' (97) and EOB (256) get 1-bit codes and the single distance length is 0
    bits = []

    def put(v, n):
        for k in range(n):
            bits.append((v >> k) & 1)
    put(1, 1); put(2, 2); put(0, 5); put(0, 5); put(15, 4)
    order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
    cl = {0: 1, 1: 2, 18: 2}                                  # code-length code: sym 0 -> '0', sym 1 -> '10', sym 18 -> '11'
    for s in order:
        put(cl.get(s, 0), 3)

    def code(sym):                                            # canonical codes, MSB first
        return {0: (0, 1), 1: (0b10, 2), 18: (0b11, 2)}[sym]

    def emit(sym):
        c, n = code(sym)
        for k in range(n - 1, -1, -1):
            bits.append((c >> k) & 1)
    # lengths: 97 zeros, 'a' = 1, 158 zeros, EOB = 1, then the one distance length = 0
    emit(18); put(97 - 11, 7)
    emit(1)
    emit(18); put(138 - 11, 7)
    emit(18); put(20 - 11, 7)
    emit(1)
    emit(0)
    for _ in range(5):
        bits.append(0)                                        # 'a' x5: code '0'
    bits.append(1)                                            # EOB: code '1'
    while len(bits) % 8:
        bits.append(0)
    body = bytes(sum(bits[i + k] << k for k in range(8)) for i in range(0, len(bits), 8))
    return b"\x78\x9c" + body + zlib.adler32(b"aaaaa").to_bytes(4, "big")
