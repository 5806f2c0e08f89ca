"""Host-side 2-bit packer (row B1): the threaded pack_records -- which sizes its buffer without
zero-filling it -- against a naive single pass, under MALLOC_PERTURB_ so that a word nobody wrote
shows up as garbage."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pack_records_threaded_equals_naive(built_lib, tmp_path):
    csrc = os.path.join(ROOT, "mitoflex_amd", "csrc")
    exe = str(tmp_path / "pack_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", csrc, os.path.join(ROOT, "tests", "native", "pack_check.cpp"),
                           os.path.join(csrc, "build", "mf_host.o"), "-lz", "-lpthread", "-o", exe])
    env = dict(os.environ, MALLOC_PERTURB_="165")
    out = subprocess.check_output([exe], env=env).decode()
    assert out.strip() == "pack ok"
