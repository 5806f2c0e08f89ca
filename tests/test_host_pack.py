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


def test_outfile_gzip_single_member_round_trip(built_lib, tmp_path):
    """OutFile compresses .gz output in 1 MiB slices on several threads but writes ONE gzip member -- what the
    reference's flate2 GzEncoder writes and its single-member GzDecoder (helper.rs:22) reads.  Whatever goes in comes back
    out of a single-member reader (zlib.decompressobj stops at the end of the first member), of Python's gzip, of
    `gzip -dc`, and (tests/test_pipeline_host.py) of the library's own decoders."""
    import gzip
    import random
    import zlib
    csrc = os.path.join(ROOT, "mitoflex_amd", "csrc")
    exe = str(tmp_path / "outfile_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", csrc, os.path.join(ROOT, "tests", "native", "outfile_check.cpp"),
                           os.path.join(csrc, "build", "mf_host.o"), "-lz", "-lpthread", "-o", exe])
    rng = random.Random(3)
    for size in (0, 1, 70_000, 1 << 20, 9_000_000):
        raw = bytes(rng.choices(b"ACGTN\n@+FI", k=size))
        src, dst, plain = tmp_path / "in.bin", tmp_path / "out.fq.gz", tmp_path / "out.fq"
        src.write_bytes(raw)
        for threads in (1, 5):
            assert subprocess.check_output([exe, str(src), str(dst), str(threads)]).decode().startswith("ok")
            blob = dst.read_bytes()
            assert gzip.decompress(blob) == raw
            d = zlib.decompressobj(wbits=31)                     # one member only, like flate2::read::GzDecoder
            assert d.decompress(blob) + d.flush() == raw and d.eof and d.unused_data == b""
            assert subprocess.check_output(["gzip", "-dc", str(dst)]) == raw
        assert subprocess.check_output([exe, str(src), str(plain), "3"]).decode().startswith("ok")
        assert plain.read_bytes() == raw
