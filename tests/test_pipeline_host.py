"""The FASTQ file pipeline without a GPU: readers (mapped plain files, serial and parallel gzip, indexed stream
parsing), batch pools, the packer and the ordered writers run with a stand-in filter ("first base is A") and
are compared with the obvious Python; tiny batches and parse segments put a border inside almost every record."""
import gzip
import os
import random
import subprocess

import pytest

from tests.util_data import write_fastq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRCS = ["mf_pipeline.cpp", "mf_host.cpp", "mf_inflate.cpp", "mf_pinflate.cpp"]


def build(tmp, flags=()):
    csrc = os.path.join(ROOT, "mitoflex_amd", "csrc")
    exe = os.path.join(tmp, "pipeline_check" + "_".join(f.replace("=", "").replace(",", "") for f in flags))
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", *flags, "-I", csrc, os.path.join(ROOT, "tests", "native", "pipeline_check.cpp"),
                           *[os.path.join(csrc, s) for s in SRCS], "-lz", "-lpthread", "-o", exe])
    return exe


@pytest.fixture(scope="module")
def data(tmp_path_factory):
    d = tmp_path_factory.mktemp("pipe")
    rng = random.Random(42)
    def seqs(n):
        out = []
        for _ in range(n):
            L = rng.choice([150, 150, 151, 40, 0, 1, 300])
            out.append("".join(rng.choices("ACGTNacgt", k=L)))
        return out
    s1, s2 = seqs(5000), seqs(5100)
    for gz in (False, True):
        ext = ".fq.gz" if gz else ".fq"
        write_fastq(str(d / ("a_1" + ext)), s1, "a", trailing_partial=True, gz=gz)
        write_fastq(str(d / ("a_2" + ext)), s2, "b", crlf=True, gz=gz)
    return d, s1, s2


def expected(d, s1, s2, both):
    r1 = open(d / "a_1.fq", newline="").read().split("\n")
    r2 = open(d / "a_2.fq", newline="").read().split("\n")
    n = min(len(s1), len(s2))
    keep = []
    for i in range(n):
        a, b = s1[i][:1] in ("A", "a"), s2[i][:1] in ("A", "a")
        keep.append((a and b) if both else (a or b))
    def out(lines, strip_cr):
        o = []
        for i in range(n):
            if keep[i]:
                rec = [x.rstrip("\r") if strip_cr else x for x in lines[4 * i:4 * i + 4]]
                o.append(rec[0] + "\n" + rec[1] + "\n+\n" + rec[3] + "\n")
        return "".join(o)
    return sum(keep), n, out(r1, False), out(r2, True)


@pytest.mark.parametrize("flags", [(), ("-fsanitize=address,undefined", "-fno-omit-frame-pointer"), ("-fsanitize=thread",)],
                         ids=["plain", "asan_ubsan", "tsan"])
def test_pipeline_with_stand_in_filter(data, tmp_path, flags):
    d, s1, s2 = data
    exe = build(str(tmp_path), flags)
    for gz in (False, True):
        ext = ".fq.gz" if gz else ".fq"
        for batch, threads, seg, extra in ((2_000_000, 8, None, {}), (700, 4, "997", {}), (64, 3, "64", {}), (700, 1, None, {}),
                                          (1000, 6, "4096", {"MF_SERIAL_INFLATE": "1"}), (1000, 6, None, {"MF_ZLIB_INFLATE": "1"})):
            for both in (False, True):
                env = dict(os.environ, **extra)
                if seg:
                    env["MF_PARSE_SEG"] = seg
                o1, o2 = str(tmp_path / "o1.fq"), str(tmp_path / "o2.fq")
                p = subprocess.run([exe, str(d / ("a_1" + ext)), str(d / ("a_2" + ext)), o1, o2, str(batch), str(threads)] + (["both"] if both else []),
                                   capture_output=True, env=env)
                err = p.stderr.decode()
                assert "Sanitizer" not in err and "runtime error" not in err, err[:2000]
                kept, total, w1, w2 = expected(d, s1, s2, both)
                assert p.stdout.decode().split() == [str(kept), str(total)], (gz, batch, threads, seg, both, p.stdout, err[:500])
                assert open(o1, newline="").read() == w1 and open(o2, newline="").read() == w2, (gz, batch, threads, seg, both)
    # single end, gz output
    o = str(tmp_path / "se.fq.gz")
    p = subprocess.run([exe, str(d / "a_1.fq.gz"), "-", o, "-", "900", "5"], capture_output=True)
    kept = sum(s[:1] in ("A", "a") for s in s1)
    assert p.stdout.decode().split() == [str(kept), str(len(s1))] and "Sanitizer" not in p.stderr.decode()
    assert gzip.open(o, "rt", newline="").read().count("\n") == 4 * kept


@pytest.mark.parametrize("flags", [(), ("-fsanitize=thread",)], ids=["plain", "tsan"])
def test_pipeline_many_devices(data, tmp_path, flags):
    """N device workers with randomised delays (batches come back out of order): output bytes and counts equal the
    single-device run's, for paired and single-end input."""
    d, s1, s2 = data
    exe = build(str(tmp_path), flags)
    kept, total, w1, w2 = expected(d, s1, s2, False)
    for ext in (".fq", ".fq.gz"):
        for n_dev in (1, 2, 4, 8):
            o1, o2 = str(tmp_path / "o1.fq"), str(tmp_path / "o2.fq")
            p = subprocess.run([exe, str(d / ("a_1" + ext)), str(d / ("a_2" + ext)), o1, o2, "300", "4", "either", str(n_dev)], capture_output=True)
            err = p.stderr.decode()
            assert "Sanitizer" not in err and "runtime error" not in err, err[:2000]
            assert p.stdout.decode().split() == [str(kept), str(total)], (ext, n_dev, p.stdout, err[:500])
            assert open(o1, newline="").read() == w1 and open(o2, newline="").read() == w2, (ext, n_dev)
    o = str(tmp_path / "se.fq")
    p = subprocess.run([exe, str(d / "a_1.fq"), "-", o, "-", "250", "3", "either", "5"], capture_output=True)
    assert "Sanitizer" not in p.stderr.decode()
    assert p.stdout.decode().split() == [str(sum(s[:1] in ("A", "a") for s in s1)), str(len(s1))]


@pytest.mark.parametrize("flags", [(), ("-fsanitize=thread",)], ids=["plain", "tsan"])
def test_pipeline_mates_through_fifos(data, tmp_path, flags):
    """Paired-end input from pipes: a pipe hands over at most 64 KiB per read() and a producer may pause, so the two readers'
    batches differ in size.  Mates are paired by record count (what is left of the longer batch meets the other mate's next
    one); a mate has run out only when its reader has.  One of the writers trickles its file in bursts with pauses longer
    than the reader's idle time-out, so that batches are handed over early on that side only."""
    import threading
    import time
    d, s1, s2 = data
    exe = build(str(tmp_path), flags)
    kept, total, w1, w2 = expected(d, s1, s2, False)
    for slow in (None, 0, 1):
        f1, f2 = str(tmp_path / "p1.fifo"), str(tmp_path / "p2.fifo")
        for f in (f1, f2):
            if os.path.exists(f):
                os.unlink(f)
            os.mkfifo(f)
        def feed(src, dst, bursts):
            data_ = open(src, "rb").read()
            with open(dst, "wb", buffering=0) as w:
                if not bursts:
                    w.write(data_)
                else:
                    step = len(data_) // 5 + 1
                    for a in range(0, len(data_), step):
                        w.write(data_[a:a + step])
                        time.sleep(0.12)
        th = [threading.Thread(target=feed, args=(str(d / "a_1.fq"), f1, slow == 0)),
              threading.Thread(target=feed, args=(str(d / "a_2.fq"), f2, slow == 1))]
        for t in th:
            t.start()
        o1, o2 = str(tmp_path / "o1.fq"), str(tmp_path / "o2.fq")
        p = subprocess.run([exe, f1, f2, o1, o2, "1500", "4", "either"], capture_output=True, timeout=120)
        for t in th:
            t.join()
        err = p.stderr.decode()
        assert "Sanitizer" not in err and "runtime error" not in err, err[:2000]
        assert p.stdout.decode().split() == [str(kept), str(total)], (slow, p.stdout, err[:500])
        assert open(o1, newline="").read() == w1 and open(o2, newline="").read() == w2, slow
