"""CPU: the C oracle under AddressSanitizer + UndefinedBehaviorSanitizer (the
counterpart of the reference's intbv range checks)."""
import os
import subprocess

from conftest import REPO


def test_oracle_asan_ubsan_selftest():
    r = subprocess.run(["make", "-s", "-C", os.path.join(REPO, "oracle"), "selftest"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "selftest OK" in r.stdout
