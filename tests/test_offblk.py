"""The block index over the offsets of a ragged read set -- one 8-byte entry per 128 bases, so that the read of a base is one load in the kernels
(mitoflex_amd/csrc/mf_common.h: OffBlk, offblk_make, offblk_lookup; read_holding and build_off_blk_kernel in mf_kernels.hip call them) -- is plain
host + device code: tests/native/offblk_check.cpp holds it to a search over the offsets on random read-length mixes (empty reads, reads of a few bases,
reads of thousands), under ASan + UBSan.  The GPU suite's ragged-read parity tests run the same functions on the device."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_block_index_against_a_search(tmp_path):
    exe = str(tmp_path / "offblk_check")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", os.path.join(ROOT, "tests", "native", "hipstub"),
                           "-I", os.path.join(ROOT, "mitoflex_amd", "csrc"), os.path.join(ROOT, "tests", "native", "offblk_check.cpp"), "-o", exe])
    r = subprocess.run([exe, "300"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "equal to the model" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])
