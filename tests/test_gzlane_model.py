"""The lane-parallel block decode of the device DEFLATE decoder without a GPU: tools/gzlane_model.cpp runs the SAME per-lane walk the
kernel runs (mitoflex_amd/csrc/mf_gzlane.h, compiled for the host: table entry formats, the pairing of literals, the rule for where a
span ends) over 64 emulated lanes with the kernel's protocol -- guessed starts, re-walks until every lane starts where its predecessor
ended, the confirmed prefix -- expands the step's joined list the way the kernel's rounds do (a cell per output position: final symbol,
reference into the round, reference to earlier output; pointer doubling) and compares with zlib's output byte for byte.  Streams of every kind a compressor writes:
levels 1 / 6 / 9, fixed Huffman codes, Huffman only (no matches), RLE (distance 1 only), binned qualities, long reads."""
import os
import random
import subprocess
import zlib

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("gzlane") / "gzlane_model")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tools", "gzlane_model.cpp"), "-lz", "-o", exe])
    return exe


def fastq(n, L, quals, seed):
    rng = random.Random(seed)
    out = []
    for i in range(n):
        ln = L if isinstance(L, int) else rng.randint(*L)
        out.append("@r.%d lane:%d\n%s\n+\n%s\n" % (i, i % 8, "".join(rng.choice("ACGT") for _ in range(ln)), "".join(rng.choice(quals) for _ in range(ln))))
    return "".join(out).encode()


STREAMS = {"level1": (1, zlib.Z_DEFAULT_STRATEGY), "level6": (6, zlib.Z_DEFAULT_STRATEGY), "level9": (9, zlib.Z_DEFAULT_STRATEGY),
           "fixed": (6, zlib.Z_FIXED), "huffman_only": (6, zlib.Z_HUFFMAN_ONLY), "rle": (6, zlib.Z_RLE), "filtered": (6, zlib.Z_FILTERED)}


@pytest.mark.parametrize("kind", sorted(STREAMS))
@pytest.mark.parametrize("text", ["short_reads", "binned_long_reads", "runs_of_identical_reads"])
def test_lane_walk_equals_zlib(model, tmp_path, kind, text):
    if text == "short_reads":
        t = fastq(6000, 100, "#,:FI58<AEJ", 3)
    elif text == "binned_long_reads":
        t = fastq(300, (2000, 9000), "F:,#", 4)
    else:          # matches longer than their distance, period after period: symbol t of a match is position first + t, no modulo
        t = b"".join(fastq(1, 150, "F", 50 + i) * 400 + b"A" * 3000 + b"\n" for i in range(12))
    level, strategy = STREAMS[kind]
    c = zlib.compressobj(level, zlib.DEFLATED, 31, 9, strategy)
    path = str(tmp_path / "t.gz")
    open(path, "wb").write(c.compress(t) + c.flush())
    for span_bits, cells in (("768", "1"), ("768", "0"), ("1024", "1"), ("2048", "1"), ("256", "1")):          # (768: the kernel's span since round 5)          # cells = 1: the kernel's expansion rounds restated, 0: a plain LZ77 copy
        p = subprocess.run([model, path, span_bits, "6", cells], capture_output=True, timeout=300)
        assert p.returncode == 0 and p.stdout.decode().strip().endswith("PASS"), (kind, text, span_bits, cells, p.stdout[-600:], p.stderr[-600:])


def test_link_walk_against_a_model(tmp_path):
    """gz_link_walk (mf_gzdev.h) -- which chunks of a slab are accepted, where their text goes, where the walk stops -- is host code since round 5
    (the device does the windows).  tests/native/linkwalk_check.cpp holds it to an obviously right model on 20 000 random descriptor sequences
    (honest chunks, chunks that found nothing, false starts inside accepted data, gaps, members' ends), walked a slab at a time; under ASan + UBSan."""
    exe = str(tmp_path / "linkwalk_check")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", os.path.join(ROOT, "tests", "native", "hipstub"),
                           "-I", os.path.join(ROOT, "mitoflex_amd", "csrc"), os.path.join(ROOT, "tests", "native", "linkwalk_check.cpp"), "-o", exe])
    r = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "equal to the model" in r.stdout, r.stderr[-2000:]
