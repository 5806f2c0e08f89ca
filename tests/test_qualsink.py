"""Host-side plumbing of the quality filter's device path (mitoflex_amd/csrc/mf_qualsink.h) without a GPU: the per-record arrays the
two mates' pieces exchange counts and keep flags through, the pool of chunks the output comes down through, the writer thread of an
output file taking chunks out of order (a regular file: at their offsets; anything else: in order).  tests/native/qualsink_check.cpp,
plain and under ThreadSanitizer."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("flags", [(), ("-fsanitize=thread",)], ids=["plain", "tsan"])
def test_qualsink_check(tmp_path, flags):
    csrc = os.path.join(ROOT, "mitoflex_amd", "csrc")
    exe = str(tmp_path / "qualsink_check")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", *flags, "-I", csrc, os.path.join(ROOT, "tests", "native", "qualsink_check.cpp"),
                           os.path.join(csrc, "mf_host.cpp"), "-lz", "-lpthread", "-o", exe])
    p = subprocess.run([exe, str(tmp_path)], capture_output=True, timeout=300)
    assert p.returncode == 0 and p.stdout.strip() == b"OK", (p.stdout[-2000:], p.stderr[-4000:])
    assert b"ThreadSanitizer" not in p.stderr, p.stderr[-4000:]
