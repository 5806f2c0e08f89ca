"""The device ingest path of mf_filter_fastq_files (mitoflex_amd/csrc/mf_devingest.cpp, mf_gzdev.hip, mf_ingest.hip): the
file's bytes go to the GPU as they are; inflate, line indexing, 2-bit packing, the filter and the copy of the survivors run
there.  Held to the oracle (output bytes, kept / total counts) and to the host pipeline (MF_INGEST=host) on inputs that put a
seam everywhere: tiny speculative chunks and slabs, every gzip level incl. stored blocks, several members, CRLF, an
unterminated last line, a partial record at the end, records longer than the deflate window, mates out of step.
Reference conventions: filter/filter_bin/src/helper.rs:14-31 (.gz by extension), main.rs:287-321 (4-line records)."""
import gzip
import os
import random
import zlib

import numpy as np
import pytest

from tests.util_data import make_reads, write_fastq

pytestmark = pytest.mark.gpu
# the library with the test hooks compiled in (MF_FAKE_DEVICES, MF_DEVPOOL_FAIL_AT): the shipped one carries neither
HOOKS_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mitoflex_amd", "libmitofilter_hip_hooks.so")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def device_ingest(monkeypatch):
    """(regular files take the device path by default; forced here all the same, so that a test that means this path cannot end up on the other)"""
    monkeypatch.setenv("MF_INGEST", "device")


@pytest.fixture(scope="module")
def mf():
    from mitoflex_amd import mitofilter
    mitofilter.load()
    return mitofilter


@pytest.fixture(scope="module")
def ol():
    from oracle import oracle_lib
    return oracle_lib


@pytest.fixture(scope="module")
def bait_text():
    from tests.util_data import make_bait
    return make_bait()


def fastq_text(seqs, prefix, crlf=False, tail=b"", last_newline=True, seed=7):
    rng = random.Random(seed)
    nl = b"\r\n" if crlf else b"\n"
    out = []
    for i, s in enumerate(seqs):
        q = "".join(rng.choice("#,:FI") for _ in s)
        out.append(b"@" + f"{prefix}.{i} some comment".encode() + nl + s.encode() + nl + b"+" + (b"anything" if i % 3 == 0 else b"") + nl + q.encode() + nl)
    t = b"".join(out) + tail
    if not last_newline and t.endswith(nl):
        t = t[:-len(nl)]
    return t


def gz_bytes(text, level, members=1):
    """one or several gzip members; level 0 = stored blocks only"""
    if members == 1:
        c = zlib.compressobj(level, zlib.DEFLATED, 31)
        return c.compress(text) + c.flush()
    cut = [len(text) * i // members for i in range(members + 1)]
    return b"".join(gz_bytes(text[cut[i]:cut[i + 1]], level) for i in range(members))


def run_both(mf, ol, bait_path, ks, fq1, fq2, tmp_path, thr=1, pair_mode=0):
    o1, o2 = str(tmp_path / "o1.fq"), str(tmp_path / "o2.fq")
    g1, g2 = str(tmp_path / "g1.fq"), str(tmp_path / "g2.fq")
    ok, ot = ol.filter_fastq_files(bait_path, 31, thr, pair_mode, fq1, fq2, o1, o2 if fq2 else None, threads=2)
    gk, gt = mf.filter_fastq_files(ks, fq1, fq2, g1, g2 if fq2 else None, thr, pair_mode)
    assert (gk, gt) == (ok, ot)
    assert open(g1, "rb").read() == open(o1, "rb").read()
    if fq2:
        assert open(g2, "rb").read() == open(o2, "rb").read()
    return gk, gt


SEAMS = {"MF_GZDEV_CHUNK_BYTES": "4096", "MF_GZDEV_SLAB_CHUNKS": "3", "MF_INGEST_SLAB_BYTES": "70001"}


@pytest.mark.parametrize("level", [0, 1, 6, 9])
@pytest.mark.parametrize("seams", [False, True])
def test_gz_levels_and_seams(mf, ol, bait_text, tmp_path, monkeypatch, level, seams):
    if seams:
        for k, v in SEAMS.items():
            monkeypatch.setenv(k, v)
    bait = str(tmp_path / "bait.fa")
    open(bait, "w").write(bait_text)
    ks = mf.KmerSet.from_fasta(bait, 31)
    s1 = make_reads(bait_text, 4000, seed=31)
    s2 = make_reads(bait_text, 4100, seed=32)
    fq1, fq2 = str(tmp_path / "a_1.fq.gz"), str(tmp_path / "a_2.fq.gz")
    open(fq1, "wb").write(gz_bytes(fastq_text(s1, "a", tail=b"@partial\nACGT\n"), level))
    open(fq2, "wb").write(gz_bytes(fastq_text(s2, "b", crlf=True, last_newline=False), level))
    for pair_mode in (mf.PAIR_EITHER, mf.PAIR_BOTH):
        k, t = run_both(mf, ol, bait, ks, fq1, fq2, tmp_path, 1, pair_mode)
        assert t == 4000 and 0 < k < t
    run_both(mf, ol, bait, ks, fq2, None, tmp_path, 2)


@pytest.mark.parametrize("seams", [False, True])
def test_plain_files_and_seams(mf, ol, bait_text, tmp_path, monkeypatch, seams):
    if seams:
        for k, v in SEAMS.items():
            monkeypatch.setenv(k, v)
        monkeypatch.setenv("MF_INGEST_SLAB_BYTES", "333")       # about one record per slab: every slab carries a partial record
    bait = str(tmp_path / "bait.fa")
    open(bait, "w").write(bait_text)
    ks = mf.KmerSet.from_fasta(bait, 31)
    n = 600 if seams else 5000
    s1 = make_reads(bait_text, n, seed=41)
    s2 = make_reads(bait_text, n + 7, seed=42)
    fq1, fq2 = str(tmp_path / "a_1.fq"), str(tmp_path / "a_2.fq")
    open(fq1, "wb").write(fastq_text(s1, "a", crlf=True, tail=b"@partial\r\nAC"))
    open(fq2, "wb").write(fastq_text(s2, "b", last_newline=False))
    k, t = run_both(mf, ol, bait, ks, fq1, fq2, tmp_path)
    assert t == n
    run_both(mf, ol, bait, ks, fq1, None, tmp_path)
    run_both(mf, ol, bait, ks, fq2, None, tmp_path, 3)


def test_records_longer_than_the_window(mf, ol, bait_text, tmp_path, monkeypatch):
    """A carry longer than 32 KiB: the head of the record comes from the previous slab's buffer."""
    for k, v in SEAMS.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("MF_INGEST_SLAB_BYTES", "50000")
    from tests.util_data import bait_records
    g = bait_records(bait_text)[0]
    rng = random.Random(5)
    seqs = []
    for i in range(40):
        if i % 4 == 0:
            seqs.append("".join(rng.choice("ACGT") for _ in range(rng.randrange(40000, 120000))) + (g[100:400] if i % 8 == 0 else ""))
        else:
            seqs.append(g[rng.randrange(0, 8000):][:150] if i % 3 else "".join(rng.choice("ACGT") for _ in range(150)))
    text = fastq_text(seqs, "long")
    bait = str(tmp_path / "bait.fa")
    open(bait, "w").write(bait_text)
    ks = mf.KmerSet.from_fasta(bait, 31)
    fq, fqgz = str(tmp_path / "l.fq"), str(tmp_path / "l.fq.gz")
    open(fq, "wb").write(text)
    open(fqgz, "wb").write(gz_bytes(text, 6))
    k1, t1 = run_both(mf, ol, bait, ks, fq, None, tmp_path)
    k2, t2 = run_both(mf, ol, bait, ks, fqgz, None, tmp_path)
    assert (k1, t1) == (k2, t2) and t1 == 40 and k1 >= 5


@pytest.mark.parametrize("members", [2, 5])
def test_several_members(mf, ol, bait_text, tmp_path, monkeypatch, members):
    """gzread (and this reader) decode every member of a multi-member file; member boundaries fall inside records."""
    for k, v in SEAMS.items():
        monkeypatch.setenv(k, v)
    bait = str(tmp_path / "bait.fa")
    open(bait, "w").write(bait_text)
    ks = mf.KmerSet.from_fasta(bait, 31)
    s = make_reads(bait_text, 3000, seed=51)
    fq = str(tmp_path / "m.fq.gz")
    open(fq, "wb").write(gz_bytes(fastq_text(s, "m"), 6, members) + b"trailing garbage that is not a member")
    k, t = run_both(mf, ol, bait, ks, fq, None, tmp_path)
    assert t == 3000


@pytest.mark.parametrize("flush", ["sync", "full", "mixed"])
def test_flush_points_are_not_gaps(mf, ol, bait_text, tmp_path, monkeypatch, capfd, flush):
    """pigz, bgzip-less parallel compressors and anything written with Z_SYNC_FLUSH put an empty stored block behind every few
    ten kilobytes of input: the chunk in front decodes across them on the device --
    the host bridges (next to) nothing."""
    monkeypatch.setenv("MF_GZDEV_CHUNK_BYTES", "16384")
    monkeypatch.setenv("MF_GZDEV_SLAB_CHUNKS", "7")
    monkeypatch.setenv("MF_PIPE_TIMING", "1")
    bait = str(tmp_path / "bait.fa")
    open(bait, "w").write(bait_text)
    ks = mf.KmerSet.from_fasta(bait, 31)
    text = fastq_text(make_reads(bait_text, 30000, seed=61), "f")
    rng = random.Random(9)
    c = zlib.compressobj(6, zlib.DEFLATED, 31)
    out, pos, n_flush = [], 0, 0
    while pos < len(text):
        n = rng.randrange(20000, 140000)
        out.append(c.compress(text[pos:pos + n]))
        pos += n
        kind = flush if flush != "mixed" else rng.choice(["sync", "full", "none"])
        if kind != "none":
            out.append(c.flush(zlib.Z_SYNC_FLUSH if kind == "sync" else zlib.Z_FULL_FLUSH))
            n_flush += 1
    out.append(c.flush())
    fq = str(tmp_path / "f.fq.gz")
    open(fq, "wb").write(b"".join(out))
    capfd.readouterr()
    k, t = run_both(mf, ol, bait, ks, fq, None, tmp_path)
    assert t == 30000 and n_flush > 20
    err = capfd.readouterr().err
    import re
    m = re.search(r"(\d+) of (\d+) chunks of \d+ KiB linked, (\d+) gaps bridged on the host, (\d+) bytes decoded there", err)
    assert m, err
    linked, chunks, gaps, gap_bytes = map(int, m.groups())
    assert chunks > 50 and linked > chunks // 2             # (a 16 KiB chunk no block starts in is passed over, not linked)
    assert gaps <= 2 and gap_bytes < 200000, err            # (the end of the file, at most)


def test_same_as_host_pipeline(mf, bait_text, tmp_path, monkeypatch):
    """the two ingest paths of the library write the same bytes (gz output included)"""
    bait = str(tmp_path / "bait.fa")
    open(bait, "w").write(bait_text)
    ks = mf.KmerSet.from_fasta(bait, 31)
    s1, s2 = make_reads(bait_text, 20000, seed=61), make_reads(bait_text, 20000, seed=62)
    fq1, fq2 = str(tmp_path / "a_1.fq.gz"), str(tmp_path / "a_2.fq.gz")
    write_fastq(fq1, s1, "a", gz=True)
    write_fastq(fq2, s2, "b", crlf=True, gz=True)
    d1, d2, h1, h2 = (str(tmp_path / x) for x in ("d1.fq.gz", "d2.fq.gz", "h1.fq.gz", "h2.fq.gz"))
    a = mf.filter_fastq_files(ks, fq1, fq2, d1, d2, 1, mf.PAIR_EITHER)
    monkeypatch.setenv("MF_INGEST", "host")
    b = mf.filter_fastq_files(ks, fq1, fq2, h1, h2, 1, mf.PAIR_EITHER)
    assert a == b and a[1] == 20000
    assert gzip.open(d1).read() == gzip.open(h1).read() and gzip.open(d2).read() == gzip.open(h2).read()


def test_damaged_gz_is_an_error(mf, bait_text, tmp_path):
    bait = str(tmp_path / "bait.fa")
    open(bait, "w").write(bait_text)
    ks = mf.KmerSet.from_fasta(bait, 31)
    s = make_reads(bait_text, 3000, seed=71)
    good = gz_bytes(fastq_text(s, "d"), 6)
    out = str(tmp_path / "o.fq")
    # a flipped bit in the middle (CRC or code error), a wrong CRC, a wrong length, a truncated file
    cases = {"flip": bytearray(good), "crc": bytearray(good), "len": bytearray(good), "cut": bytearray(good[:len(good) * 2 // 3])}
    cases["flip"][len(good) // 2] ^= 0x10
    cases["crc"][-8] ^= 1
    cases["len"][-1] ^= 1
    for name, data in cases.items():
        p = str(tmp_path / (name + ".fq.gz"))
        open(p, "wb").write(bytes(data))
        with pytest.raises(mf.MitoFilterError):
            mf.filter_fastq_files(ks, p, None, out, None)


def test_highly_compressible_input_grows_the_symbol_buffers(mf, ol, bait_text, tmp_path):
    """identical reads compress several hundredfold: the decoder's per-chunk symbol room is enlarged and the slab decoded again"""
    from tests.util_data import bait_records
    g = bait_records(bait_text)[0]
    seqs = [g[500:650]] * 30000 + ["ACGT" * 30] * 30000
    rng = random.Random(3)
    text = b"".join(b"@same\n" + s.encode() + b"\n+\n" + b"F" * len(s) + b"\n" for s in seqs)
    fq = str(tmp_path / "same.fq.gz")
    open(fq, "wb").write(gz_bytes(text, 9))
    assert os.path.getsize(fq) * 100 < len(text)
    bait = str(tmp_path / "bait.fa")
    open(bait, "w").write(bait_text)
    ks = mf.KmerSet.from_fasta(bait, 31)
    k, t = run_both(mf, ol, bait, ks, fq, None, tmp_path)
    assert (k, t) == (30000, 60000)


# ---- round 4: the path streams -- a ring for the compressed bytes, a text buffer per piece, a bounded number of both

STREAMING = {"MF_GZDEV_CHUNK_BYTES": "4096", "MF_GZDEV_SLAB_CHUNKS": "3", "MF_GZDEV_RING_BYTES": "65536", "MF_GZDEV_MARGIN": "8192",
             "MF_INGEST_TEXT_BUFS": "2", "MF_GZDEV_TEXT_PIECE": "20000", "MF_INGEST_SLAB_BYTES": "50001"}


@pytest.mark.parametrize("level", [1, 6])
@pytest.mark.parametrize("carry_room", ["0", "1048576"])
def test_ring_wraps_and_buffers_recycle(mf, ol, bait_text, tmp_path, monkeypatch, capfd, level, carry_room):
    """A 64 KiB ring under files of several hundred kilobytes (it wraps many times), two text buffers per mate, pieces of at most
    20 kB of text, and -- with no carry room -- every piece moved to a buffer that holds the carry too: same bytes as the oracle."""
    for k, v in STREAMING.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("MF_INGEST_CARRY_ROOM", carry_room)
    monkeypatch.setenv("MF_PIPE_TIMING", "1")
    bait = str(tmp_path / "bait.fa")
    open(bait, "w").write(bait_text)
    ks = mf.KmerSet.from_fasta(bait, 31)
    s1 = make_reads(bait_text, 6000, seed=81)
    s2 = make_reads(bait_text, 5900, seed=82)
    fq1, fq2 = str(tmp_path / "a_1.fq.gz"), str(tmp_path / "a_2.fq.gz")
    open(fq1, "wb").write(gz_bytes(fastq_text(s1, "a", tail=b"@partial\nACGT\n"), level))
    open(fq2, "wb").write(gz_bytes(fastq_text(s2, "b", crlf=True, last_newline=False), level))
    assert os.path.getsize(fq1) > 4 * 65536
    capfd.readouterr()
    k, t = run_both(mf, ol, bait, ks, fq1, fq2, tmp_path, 1, mf.PAIR_EITHER)
    assert t == 5900 and 0 < k < t
    err = capfd.readouterr().err
    assert "ring 0 MiB" in err and "declined" not in err, err           # (65536 bytes print as 0 MiB: the knob was honoured)
    run_both(mf, ol, bait, ks, fq1, None, tmp_path, 1)
    # plain text through the same buffers
    p1 = str(tmp_path / "p_1.fq")
    open(p1, "wb").write(fastq_text(s1, "a", crlf=True))
    run_both(mf, ol, bait, ks, p1, fq2, tmp_path, 1, mf.PAIR_BOTH)


def test_long_blocks_past_the_margin_are_bridged(mf, ol, bait_text, tmp_path, monkeypatch, capfd):
    """A chunk may only read what has been copied up: the bytes of its slab and a margin behind it.  With a margin of 1 KiB
    nearly every slab's last chunk runs into it (a deflate block of level 9 is tens of kilobytes long) -- those chunks stop,
    nothing of them is accepted, and the host decodes across the stretch."""
    for k, v in STREAMING.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("MF_GZDEV_MARGIN", "1024")
    monkeypatch.setenv("MF_PIPE_TIMING", "1")
    bait = str(tmp_path / "bait.fa")
    open(bait, "w").write(bait_text)
    ks = mf.KmerSet.from_fasta(bait, 31)
    s = make_reads(bait_text, 5000, seed=83)
    fq = str(tmp_path / "m.fq.gz")
    open(fq, "wb").write(gz_bytes(fastq_text(s, "m"), 9))
    capfd.readouterr()
    k, t = run_both(mf, ol, bait, ks, fq, None, tmp_path)
    assert t == 5000
    import re
    m = re.search(r"(\d+) gaps bridged on the host, (\d+) bytes decoded there", capfd.readouterr().err)
    assert m and int(m.group(1)) >= 1 and int(m.group(2)) > 200000          # (most of the text: hardly a chunk gets to the end of its block)


def _gzip_tools():
    import shutil
    tools = [("gzip", lvl) for lvl in (1, 6, 9) if shutil.which("gzip")]
    tools += [(t, 6) for t in ("pigz", "bgzip") if shutil.which(t)]
    return tools


@pytest.mark.parametrize("tool,level", _gzip_tools())
def test_files_written_by_real_compressors(mf, ol, bait_text, tmp_path, tool, level):
    """gzip -1 / -6 / -9 (and pigz, bgzip where the box has them) write the input themselves -- not zlib.compressobj, not this
    repository's tools/pgzip.py.  (bgzip writes BGZF, which this library decodes on the host: the call must still be right.)"""
    import subprocess
    bait = str(tmp_path / "bait.fa")
    open(bait, "w").write(bait_text)
    ks = mf.KmerSet.from_fasta(bait, 31)
    s1, s2 = make_reads(bait_text, 150000, seed=91), make_reads(bait_text, 150000, seed=92)
    fq1, fq2 = str(tmp_path / "r_1.fq"), str(tmp_path / "r_2.fq")
    open(fq1, "wb").write(fastq_text(s1, "a"))
    open(fq2, "wb").write(fastq_text(s2, "b"))
    for f in (fq1, fq2):
        with open(f + ".gz", "wb") as o:
            subprocess.check_call([tool, "-%d" % level, "-c", f] if tool != "bgzip" else [tool, "-c", f], stdout=o)
    k, t = run_both(mf, ol, bait, ks, fq1 + ".gz", fq2 + ".gz", tmp_path)
    assert t == 150000 and 0 < k < t


@pytest.mark.parametrize("n_dev", [2, 4])
def test_slabs_dealt_to_several_devices(ol, bait_text, tmp_path, n_dev):
    """`fastfilter bait --devices N` on .gz input takes the device path: the slabs of either stream are dealt to the N devices
    round robin, the link state travels through the host, every piece is cut, packed, filtered and gathered on the device that
    holds it.  MF_FAKE_DEVICES maps N logical devices (own contexts, streams, rings, bait tables, read sets) onto the one GPU of
    the box; a child process, because the variable is read when the library is loaded."""
    import subprocess
    s1 = make_reads(bait_text, 9000, seed=15)
    s2 = make_reads(bait_text, 9100, seed=16)
    fq1, fq2 = str(tmp_path / "a_1.fq.gz"), str(tmp_path / "a_2.fq.gz")
    open(fq1, "wb").write(gz_bytes(fastq_text(s1, "a", tail=b"@partial\nACGT\n"), 6))
    open(fq2, "wb").write(gz_bytes(fastq_text(s2, "b", crlf=True, last_newline=False), 1, members=3))
    bait = str(tmp_path / "bait.fa")
    open(bait, "w").write(bait_text)
    o1, o2 = str(tmp_path / "o1.fq"), str(tmp_path / "o2.fq")
    ok, ot = ol.filter_fastq_files(bait, 31, 1, 0, fq1, fq2, o1, o2, threads=2)
    g1, g2 = str(tmp_path / "g1.fq"), str(tmp_path / "g2.fq")
    cli = os.path.join(ROOT, "mitoflex_amd", "assemble", "fastfilter")
    env = dict(os.environ, MITOFILTER_LIB=HOOKS_LIB, MF_FAKE_DEVICES=str(n_dev), MF_PIPE_TIMING="1", MF_GZDEV_CHUNK_BYTES="8192", MF_GZDEV_SLAB_CHUNKS="5",
               MF_GZDEV_TEXT_PIECE="200000")
    env.pop("MF_INGEST", None)
    for args in (["--devices", str(n_dev)], ["--device-list", ",".join(str(d) for d in reversed(range(1, n_dev)))]):
        p = subprocess.run([cli, "bait", "--bait", bait, "-k", "31", "--fq1", fq1, "--fq2", fq2, "--out1", g1, "--out2", g2] + args,
                           capture_output=True, env=env, timeout=300)
        err = p.stderr.decode()
        assert p.returncode == 0, err[:3000]
        n_used = n_dev if args[0] == "--devices" else n_dev - 1
        assert "[mf device ingest] wall" in err and ("%d device(s)" % n_used) in err and "declined" not in err, err[:3000]
        assert int(p.stdout.decode().split()[0]) == ok and ot == 9000
        assert open(g1, "rb").read() == open(o1, "rb").read()
        assert open(g2, "rb").read() == open(o2, "rb").read()


@pytest.mark.parametrize("knobs", [dict(MF_GZDEV_LARGE_MB="0", MF_GZDEV_RESERVED_CUS="8", MF_UPLOAD_THREADS="1", MF_UPLOAD_STAGED="1"), dict(MF_GZDEV_LARGE_MB="0", MF_GZDEV_RESERVED_CUS="64", MF_UPLOAD_THREADS="16"),
                                   dict(MF_GZDEV_LARGE_MB="0", MF_GZDEV_NO_CUMASK="1"), dict(MF_GZDEV_LARGE_MB="0"), dict(MF_UPLOAD_STAGED="1"),
                                   dict(MF_GZDEV_LARGE_MB="0", MF_GZDEV_RESOLVE_STREAM="1", MF_GZDEV_DEC_STREAMS="2", MF_GZDEV_UPLOAD_BUFS="4", MF_GZDEV_UPLOAD_PIECE_MB="1"),
                                   dict(MF_GZDEV_RESOLVE_STREAM="1", MF_GZDEV_DEC_STREAMS="1", MF_UPLOAD_REGISTER_MAX_MB="0")],
                         ids=["masked-reserve8-upload1", "masked-reserve64-upload16", "masked-set-without-masks", "masked-set", "uploads-through-staging-buffers",
                              "masked-two-post-streams-2-decode-streams-4-upload-buffers", "plain-two-post-streams-1-decode-stream-nothing-registered"])
def test_stream_and_upload_knobs(knobs):
    """The CU masks of the decoder's streams and the uploader's thread count are read when a process makes its first stream set, so the
    variants run in child processes: gzip levels x seams, several members, flush points -- same bytes as the oracle whatever the knobs say.
    (Round 5: files below a gigabyte use the plain stream set by default -- the rest of this file runs on it --; MF_GZDEV_LARGE_MB=0 sends
    every input through the CU-masked set, the one large inputs get.)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_devingest.py"), "-m", "gpu", "-q", "-x",
                        "-k", "test_gz_levels_and_seams or test_several_members or test_flush_points or test_plain_files"],
                       capture_output=True, env=dict(os.environ, **knobs), cwd=root, timeout=900)
    assert p.returncode == 0, p.stdout.decode()[-3000:]


@pytest.mark.parametrize("level", [1, 6])
def test_configs4_at_full_size(mf, ol, bait_text, tmp_path_factory, monkeypatch, capfd, level):
    """BASELINE.json configs[4] at its stated size: 33 333 334 single-end reads of 150 bases in ONE gzip member, filtered file to
    file on one GPU.  The file is generated on the box (tools/make_fastq.py, 2 M-read blocks) and compressed by tools/pgzip.py
    (one member, 8 MiB slices).  Checked: the totals; the survivors of the first million reads byte for byte against the oracle
    run on that window of the text; the whole output byte for byte against the host pipeline (MF_INGEST=host) on the plain text."""
    import hashlib
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = 33_333_334
    d = tmp_path_factory.getbasetemp() / "configs4"          # (both levels share the text and the host pipeline's output)
    d.mkdir(exist_ok=True)
    fq, bait = str(d / "s_1.fq"), str(d / "s.bait.fa")
    if not os.path.exists(str(d / "host.md5")):
        subprocess.check_call([sys.executable, os.path.join(root, "tools", "make_fastq.py"), str(d / "s"), "--pairs", str(n), "--mates", "1", "--block", "2000000"],
                              stdout=subprocess.DEVNULL)
        os.environ["MF_INGEST"] = "host"
        try:
            ks = mf.KmerSet.from_fasta(bait, 31)
            res = mf.filter_fastq_files(ks, fq, None, str(d / "host.fq"), None)
        finally:
            os.environ["MF_INGEST"] = "device"
        assert res[1] == n
        open(str(d / "host.md5"), "w").write(hashlib.md5(open(str(d / "host.fq"), "rb").read()).hexdigest() + " %d %d" % res)
        # the oracle on a window: the first million reads
        with open(fq, "rb") as f, open(str(d / "win.fq"), "wb") as w:
            w.write(f.read(1_000_000 * 321))
        ok, ot = ol.filter_fastq_files(bait, 31, 1, 0, str(d / "win.fq"), None, str(d / "win.out"), None, threads=os.cpu_count() or 1)
        assert ot == 1_000_000 and ok > 1000
    want_md5, want_kept, want_total = open(str(d / "host.md5")).read().split()
    gz = str(d / ("s.l%d.fq.gz" % level))
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "pgzip.py"), fq, gz, "--level", str(level)])
    ks = mf.KmerSet.from_fasta(bait, 31)
    out = str(d / "dev.fq")
    best = 1e9
    monkeypatch.setenv("MF_PIPE_TIMING", "1")
    capfd.readouterr()
    for _ in range(2):
        t0 = time.perf_counter()
        kept, total = mf.filter_fastq_files(ks, gz, None, out, None)
        best = min(best, time.perf_counter() - t0)
    assert (kept, total) == (int(want_kept), int(want_total)) and total == n
    # the path streams: a 5 GB file (10.7 GB of text) goes through with a bounded amount of device memory
    import re
    err = capfd.readouterr().err
    used = [float(x) for x in re.findall(r"device memory in use at most ([0-9.]+) GB", err)]
    assert len(used) == 2 and max(used) < 40.0 and "declined" not in err, err[-2000:]
    print("configs[4] level %d: device memory in use at most %.1f GB" % (level, max(used)))
    got = open(out, "rb").read()
    assert hashlib.md5(got).hexdigest() == want_md5
    win = open(str(d / "win.out"), "rb").read()
    assert got[:len(win)] == win and got[len(win):len(win) + 5] == b"@syn."          # the window's survivors open the output
    print("configs[4] level %d: %.3f s, %.1f M reads/s" % (level, best, n / best / 1e6))
    os.unlink(gz)
    if level == 6:
        for f in os.listdir(str(d)):
            os.unlink(os.path.join(str(d), f))


@pytest.mark.parametrize("fail_at", [1, 4, 12, 30])
def test_a_failed_allocation_hands_the_call_to_the_host_pipeline(ol, bait_text, tmp_path, fail_at):
    """Out of device memory before a survivor has been written (here: the n-th new allocation of the process is made to fail,
    MF_DEVPOOL_FAIL_AT) is not an error of the call: the host pipeline takes the input, and the output is the oracle's.  A child
    process: the hook counts the allocations of a process."""
    import subprocess
    s1 = make_reads(bait_text, 6000, seed=25)
    s2 = make_reads(bait_text, 6000, seed=26)
    fq1, fq2 = str(tmp_path / "a_1.fq.gz"), str(tmp_path / "a_2.fq.gz")
    open(fq1, "wb").write(gz_bytes(fastq_text(s1, "a"), 6))
    open(fq2, "wb").write(gz_bytes(fastq_text(s2, "b"), 6))
    bait = str(tmp_path / "bait.fa")
    open(bait, "w").write(bait_text)
    o1, o2 = str(tmp_path / "o1.fq"), str(tmp_path / "o2.fq")
    ok, ot = ol.filter_fastq_files(bait, 31, 1, 0, fq1, fq2, o1, o2, threads=2)
    g1, g2 = str(tmp_path / "g1.fq"), str(tmp_path / "g2.fq")
    cli = os.path.join(ROOT, "mitoflex_amd", "assemble", "fastfilter")
    env = dict(os.environ, MITOFILTER_LIB=HOOKS_LIB, MF_PIPE_TIMING="1", MF_DEVPOOL_FAIL_AT=str(fail_at), MF_GZDEV_CHUNK_BYTES="16384", MF_GZDEV_SLAB_CHUNKS="8")
    env.pop("MF_INGEST", None)
    p = subprocess.run([cli, "bait", "--bait", bait, "-k", "31", "--fq1", fq1, "--fq2", fq2, "--out1", g1, "--out2", g2], capture_output=True, env=env, timeout=300)
    err = p.stderr.decode()
    assert p.returncode == 0, err[:3000]
    assert "declined" in err, err[:3000]
    assert int(p.stdout.decode().split()[0]) == ok and ot == 6000
    assert open(g1, "rb").read() == open(o1, "rb").read()
    assert open(g2, "rb").read() == open(o2, "rb").read()


# ---- round 5: the device memory of the path follows the input; what a process keeps between calls can be given back; big plain files take the path

def test_device_memory_follows_the_input(mf, ol, bait_text, tmp_path, monkeypatch):
    """A .gz pair of a few hundred megabytes must not hold what a 5 GB file does (round 4: 28 GB for a 0.6 GB pair): at most max(3 GB, 8 x the
    compressed bytes) of device memory in use -- everything on the device, the runtime's own included -- same kept / total as the oracle."""
    import subprocess
    import sys
    monkeypatch.delenv("MF_INGEST", raising=False)
    pre = str(tmp_path / "q")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_fastq.py"), pre, "--pairs", "1200000"], stdout=subprocess.DEVNULL)
    for m in "12":
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "pgzip.py"), f"{pre}_{m}.fq", f"{pre}_{m}.fq.gz", "--level", "6"], stdout=subprocess.DEVNULL)
    gz = os.path.getsize(pre + "_1.fq.gz") + os.path.getsize(pre + "_2.fq.gz")
    ks = mf.KmerSet.from_fasta(pre + ".bait.fa", 31)
    mf.release_cached()
    k, t = mf.filter_fastq_files(ks, pre + "_1.fq.gz", pre + "_2.fq.gz", str(tmp_path / "o1.fq"), str(tmp_path / "o2.fq"))
    st = mf.last_ingest_stats()
    assert st["path"] == 1 and t == 1200000
    assert st["device_bytes_peak"] <= max(3 << 30, 8 * gz), (st["device_bytes_peak"] / 1e9, gz / 1e9)
    ok, ot = ol.filter_fastq_files(pre + ".bait.fa", 31, 1, 0, pre + "_1.fq", pre + "_2.fq", str(tmp_path / "r1.fq"), str(tmp_path / "r2.fq"), threads=8)
    assert (k, t) == (ok, ot)
    assert open(tmp_path / "o1.fq", "rb").read() == open(tmp_path / "r1.fq", "rb").read()
    # what the process keeps for its next call goes back to the runtime on request (ABI 4), and the next call works all the same
    released = mf.release_cached()
    assert released > 0
    assert mf.release_cached() == 0
    assert mf.filter_fastq_files(ks, pre + "_1.fq.gz", pre + "_2.fq.gz", str(tmp_path / "o1.fq"), str(tmp_path / "o2.fq")) == (k, t)


def test_plain_files_take_the_device_path_by_default(mf, ol, bait_text, tmp_path, monkeypatch):
    """Plain regular files take the device path at every size (measured ahead of the host pipeline from 0.16 GB to 10.7 GB, profiles/r05);
    MF_INGEST_PLAIN_MIN_MB keeps files below that many megabytes on the host pipeline.  Either way the oracle's bytes."""
    monkeypatch.delenv("MF_INGEST", raising=False)
    bait = str(tmp_path / "bait.fa")
    open(bait, "w").write(bait_text)
    ks = mf.KmerSet.from_fasta(bait, 31)
    fq = str(tmp_path / "p.fq")
    open(fq, "wb").write(fastq_text(make_reads(bait_text, 30000, seed=91), "p"))
    run_both(mf, ol, bait, ks, fq, None, tmp_path)
    assert mf.last_ingest_stats()["path"] == 1
    monkeypatch.setenv("MF_INGEST_PLAIN_MIN_MB", "100000")
    run_both(mf, ol, bait, ks, fq, None, tmp_path)
    assert mf.last_ingest_stats()["path"] == 0


def test_the_clis_run_cold(mf, ol, bait_text, tmp_path, monkeypatch):
    """The path as the reference calls it: a process per call (utility/helper.py:78-86).  `fastfilter bait` on a .gz, twice, as fresh processes:
    the oracle's count and bytes, and the cold-start timeline (MF_COLD_TRACE) shows the set-up the CLI asks for (expect_files)."""
    import subprocess
    monkeypatch.delenv("MF_INGEST", raising=False)
    bait = str(tmp_path / "bait.fa")
    open(bait, "w").write(bait_text)
    s = make_reads(bait_text, 40000, seed=92)
    fq = str(tmp_path / "c.fq.gz")
    open(fq, "wb").write(gz_bytes(fastq_text(s, "c"), 6))
    ok, ot = ol.filter_fastq_files(bait, 31, 1, 0, fq, None, str(tmp_path / "r.fq"), None, threads=2)
    exe = os.path.join(ROOT, "mitoflex_amd", "assemble", "fastfilter")
    for _ in range(2):
        p = subprocess.run([exe, "bait", "--bait", bait, "--fq1", fq, "--out1", str(tmp_path / "o.fq")], capture_output=True, env=dict(os.environ, MF_COLD_TRACE="1"), timeout=120)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        assert int(p.stdout.decode()) == ok
        assert open(tmp_path / "o.fq", "rb").read() == open(tmp_path / "r.fq", "rb").read()
        err = p.stderr.decode()
        assert "prefetch: code objects of the decoder and the line kernels loaded" in err and "device ingest: first piece of text handed over" in err


@pytest.mark.parametrize("seed", range(int(os.environ.get("MF_INGEST_FUZZ_SEEDS", "10"))))
def test_random_files_under_random_knobs(mf, ol, bait_text, tmp_path, monkeypatch, seed):
    """Random inputs (single-end or paired, 1-3 gzip members or plain text, level 0-9, CRLF or not, mates of different length, an unterminated
    last line) under random sizes of everything that puts a seam somewhere -- chunk, slab, ring, margin, text piece, text buffers, consumers,
    the uploader's buffers and pieces, the bodies and CRC on their own stream or not, one or several decode streams: the oracle's bytes and
    counts every time (the seeds are fixed, the message carries the configuration)."""
    rng = random.Random(9000 + seed)
    knobs = {
        "MF_GZDEV_CHUNK_BYTES": str(rng.choice([1024, 4096, 9000, 32768])), "MF_GZDEV_SLAB_CHUNKS": str(rng.choice([1, 3, 7, 64])),
        "MF_GZDEV_RING_BYTES": str(rng.choice([65536, 262144, 1 << 22])), "MF_GZDEV_MARGIN": str(rng.choice([1024, 8192, 1 << 20])),
        "MF_GZDEV_TEXT_PIECE": str(rng.choice([20000, 300000, 1 << 30])), "MF_INGEST_TEXT_BUFS": str(rng.choice([2, 3, 6])),
        "MF_INGEST_CONSUMERS": str(rng.choice([1, 2, 3, 5])), "MF_INGEST_SLAB_BYTES": str(rng.choice([50001, 400000, 1 << 28])),
        "MF_GZDEV_UPLOAD_BUFS": str(rng.choice([2, 3, 4])), "MF_GZDEV_UPLOAD_PIECE_MB": str(rng.choice([1, 32])),
        "MF_GZDEV_RESOLVE_STREAM": rng.choice(["0", "1"]), "MF_GZDEV_DEC_STREAMS": str(rng.choice([1, 2, 4])),
        "MF_GZDEV_SLABS_IN_FLIGHT": str(rng.choice([1, 2, 5])), "MF_INGEST_CARRY_ROOM": rng.choice(["0", "1048576"]),
    }
    if rng.random() < 0.3:
        knobs["MF_UPLOAD_STAGED"] = "1"
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    bait = str(tmp_path / "bait.fa")
    open(bait, "w").write(bait_text)
    ks = mf.KmerSet.from_fasta(bait, 31)
    paired, plain = rng.random() < 0.5, rng.random() < 0.25
    level, members, crlf = rng.choice([0, 1, 6, 9]), rng.choice([1, 1, 2, 3]), rng.random() < 0.2
    n1 = rng.randint(1, 6000)
    n2 = n1 + rng.choice([0, 0, 0, 5, -1 if n1 > 1 else 0])
    names = []
    for m, n in ((1, n1), (2, n2)) if paired else ((1, n1),):
        t = fastq_text(make_reads(bait_text, n, seed=1000 * seed + m), "f%d" % m, crlf=crlf, last_newline=rng.random() < 0.8, seed=seed)
        p = str(tmp_path / ("r_%d.fq" % m)) + ("" if plain else ".gz")
        open(p, "wb").write(t if plain else gz_bytes(t, level, members))
        names.append(p)
    cfg = dict(seed=seed, paired=paired, plain=plain, level=level, members=members, crlf=crlf, n=(n1, n2), **knobs)
    try:
        run_both(mf, ol, bait, ks, names[0], names[1] if paired else None, tmp_path, thr=rng.choice([1, 1, 2]), pair_mode=rng.choice([0, 1]) if paired else 0)
    except AssertionError as e:
        raise AssertionError("%s under %r" % (e, cfg))
    assert mf.last_ingest_stats()["path"] == 1, cfg
