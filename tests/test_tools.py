"""tools/pgzip.py (the parallel single-member gzip writer bench.py and the full-size tests compress their inputs with) and
tools/make_fastq.py (their generator): what they write is what a plain reader expects."""
import gzip
import os
import random
import subprocess
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pgzip_writes_one_member_any_reader_takes(tmp_path):
    rng = random.Random(5)
    data = bytes(rng.getrandbits(8) for _ in range(1000)) * 3000 + b"".join(b"@r%d\nACGT\n+\nIIII\n" % i for i in range(50000))
    src, dst = tmp_path / "in.bin", tmp_path / "out.gz"
    src.write_bytes(data)
    for level, slice_mb, procs in ((6, 1, 3), (1, 8, 1), (9, 2, 2)):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pgzip.py"), str(src), str(dst), "--level", str(level),
                               "--slice-mb", str(slice_mb), "--procs", str(procs)])
        blob = dst.read_bytes()
        assert gzip.decompress(blob) == data
        # one member: a decompressor that stops at the end of the first member has taken every byte of the file
        d = zlib.decompressobj(31)
        assert d.decompress(blob) == data and d.eof and d.unused_data == b""
        assert int.from_bytes(blob[-8:-4], "little") == zlib.crc32(data) and int.from_bytes(blob[-4:], "little") == len(data) & 0xFFFFFFFF
    src.write_bytes(b"")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pgzip.py"), str(src), str(dst)])
    assert gzip.decompress(dst.read_bytes()) == b""


def test_make_fastq_blocks_and_mates(tmp_path):
    base = str(tmp_path / "s")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_fastq.py"), base, "--pairs", "5000", "--mates", "1", "--block", "1700"],
                          stdout=subprocess.DEVNULL)
    lines = open(base + "_1.fq", "rb").read().split(b"\n")
    assert len(lines) == 4 * 5000 + 1 and lines[-1] == b"" and not os.path.exists(base + "_2.fq")
    assert all(lines[4 * i].startswith(b"@") and lines[4 * i + 2].startswith(b"+") and len(lines[4 * i + 1]) == len(lines[4 * i + 3]) == 150 for i in range(0, 5000, 97))
    assert os.path.getsize(base + ".bait.fa") > 10000


def test_scripts_the_documents_name_exist_and_parse():
    """Every tools/ script DESIGN.md, README.md, INTEGRATION.md and round 5's part of profiles/README.md name is in the tree (records must be
    reproducible from the scripts that made them), and every shell script under tools/ parses."""
    import re
    named = set()
    for doc, start in (("DESIGN.md", None), ("README.md", None), ("INTEGRATION.md", None), (os.path.join("profiles", "README.md"), "`r05/`")):
        text = open(os.path.join(ROOT, doc)).read()
        if start:
            text = text[text.index(start):]
        named |= set(re.findall(r"tools/[A-Za-z0-9_]+\.(?:sh|py|cpp)", text))
    missing = sorted(n for n in named if not os.path.exists(os.path.join(ROOT, n)))
    assert not missing, missing
    assert len(named) > 10
    for f in sorted(os.listdir(os.path.join(ROOT, "tools"))):
        if f.endswith(".sh"):
            assert subprocess.run(["bash", "-n", os.path.join(ROOT, "tools", f)]).returncode == 0, f
