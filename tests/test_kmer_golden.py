"""Group B against committed golden vectors (tests/golden/kmer_bait_golden.json, generated from the STRING-LEVEL
specification oracle/kmer_bait_ref.py by tests/golden/make_kmer_bait_golden.py).

PARITY UNPINNED BY THE REFERENCE: MitoFlex holds no k-mer read filter, so the vectors pin this build's own Spec B.
What they add: the GPU box compares the HIP path with committed data, not only with a C oracle compiled on that box.
  * CPU (`-m "not gpu"`): the C oracle equals the vectors; the seeded inputs still have their recorded md5s.
  * GPU (`-m gpu`): the HIP path -- through the C ABI -- equals the vectors directly: per-read hit counts, pass bitmaps
    for T in {1, 3}, pair-keep counts through the file pipeline, byte-identical bait tables."""
import base64
import hashlib
import json
import os

import numpy as np
import pytest

from tests.util_data import make_bait, make_reads, write_fastq

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "kmer_bait_golden.json")))


def u16(b64):
    return np.frombuffer(base64.b64decode(b64), dtype="<u2").astype(np.uint32)


def bits(hexstr, n):
    return np.unpackbits(np.frombuffer(bytes.fromhex(hexstr), dtype=np.uint8), bitorder="little")[:n].astype(bool)


def md5_of(seqs):
    h = hashlib.md5()
    for s in seqs:
        h.update(s.encode()); h.update(b"\n")
    return h.hexdigest()


@pytest.fixture(scope="module")
def inputs():
    bait = make_bait()
    assert hashlib.md5(bait.encode()).hexdigest() == GOLD["bait_md5"], "tests/util_data.make_bait drifted: regenerate the fixture"
    pe = GOLD["pe10k"]
    m1, m2 = make_reads(bait, pe["pairs"], seed=pe["seed1"], uniform=True), make_reads(bait, pe["pairs"], seed=pe["seed2"], uniform=True)
    rg = make_reads(bait, GOLD["ragged"]["n"], seed=GOLD["ragged"]["seed"])
    assert (md5_of(m1), md5_of(m2), md5_of(rg)) == (pe["mate1_md5"], pe["mate2_md5"], GOLD["ragged"]["md5"]), "make_reads drifted"
    return bait, m1, m2, rg


# ------------------------------------------------------------------ CPU: the C oracle against the vectors
@pytest.mark.parametrize("k", [21, 31, 41])
def test_c_oracle_matches_golden(inputs, k):
    from oracle import oracle_lib as ol
    bait, m1, m2, rg = inputs
    t = ol.OracleTable(bait, k)
    for seqs, want in ((m1, u16(GOLD["pe10k"]["k"][str(k)]["hits1_u16"])), (m2, u16(GOLD["pe10k"]["k"][str(k)]["hits2_u16"])),
                       (rg, u16(GOLD["ragged"]["k"][str(k)]))):
        R = ol.OracleReads.from_seqs(seqs)
        for T in (1, 3):
            b, h = ol.filter_reads(t, R, T)
            assert np.array_equal(h, want)
            assert np.array_equal(np.unpackbits(b.view(np.uint8), bitorder="little")[:len(seqs)].astype(bool), want >= T)
    for T in (1, 3):
        g = GOLD["pe10k"]["k"][str(k)]["T"][str(T)]
        p1, p2 = bits(g["pass1_bits"], len(m1)), bits(g["pass2_bits"], len(m2))
        assert np.array_equal(p1, u16(GOLD["pe10k"]["k"][str(k)]["hits1_u16"]) >= T)
        assert int((p1 | p2).sum()) == g["kept_either"] and int((p1 & p2).sum()) == g["kept_both"]


@pytest.mark.parametrize("k", [11, 15, 21, 31, 32, 33, 41, 63])
def test_c_oracle_tables_match_golden(inputs, k):
    from oracle import oracle_lib as ol
    t = ol.OracleTable(inputs[0], k)
    g = GOLD["tables"][str(k)]
    assert (t.slots, t.n_keys) == (g["slots"], g["n_keys"])
    assert hashlib.md5(np.ascontiguousarray(t.keys).astype("<u8").tobytes()).hexdigest() == g["md5"]


@pytest.mark.parametrize("k", [11, 31])
def test_c_oracle_edge_cases_match_golden(k):
    from oracle import oracle_lib as ol
    t = ol.OracleTable(GOLD["edge"]["bait"], k)
    _, h = ol.filter_reads(t, ol.OracleReads.from_seqs(GOLD["edge"]["reads"]), 1)
    assert h.tolist() == GOLD["edge"]["k"][str(k)]


def test_numpy_packer_equals_spec_packer():
    from oracle import kmer_bait_ref as ref
    seqs = GOLD["edge"]["reads"] + make_reads(make_bait(), 40, seed=3)
    w, o, n = ref.pack_reads(seqs)
    words, offsets, npos = _pack(seqs)
    assert words.tolist() == w and offsets.tolist() == o and npos.tolist() == n


# ------------------------------------------------------------------ GPU: the HIP path against the vectors
@pytest.fixture(scope="module")
def mf(built_lib):
    from mitoflex_amd import mitofilter
    if mitofilter.device_count() < 1:
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    return mitofilter


def _pack(seqs):
    """Dense 2-bit little-endian stream straight from the strings (Spec B1; numpy, nothing of the C oracle on this path):
    base i in words[i >> 4] bits [2 (i & 15), +1], A/a=0 C/c=1 G/g=2 T/t=3, anything else stored as 0 and listed."""
    code = np.full(256, 255, dtype=np.uint8)
    for ch, v in zip(b"ACGTacgt", (0, 1, 2, 3, 0, 1, 2, 3)):
        code[ch] = v
    flat = np.frombuffer("".join(seqs).encode("latin-1"), dtype=np.uint8)
    c = code[flat]
    npos = np.nonzero(c == 255)[0].astype(np.uint64)
    c = np.where(c == 255, 0, c).astype(np.uint64)
    n = len(c)
    pad = (-n) % 16
    c = np.concatenate([c, np.zeros(pad, dtype=np.uint64)]).reshape(-1, 16)
    words = (c << (2 * np.arange(16, dtype=np.uint64))).sum(axis=1).astype(np.uint32)
    offsets = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
    return words, offsets, npos


def _gpu_reads(mf, seqs):
    words, offsets, npos = _pack(seqs)
    return mf.Reads.from_packed(words, offsets, npos, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [21, 31, 41])
def test_gpu_matches_golden(mf, inputs, k):
    bait, m1, m2, rg = inputs
    ks = mf.KmerSet.from_text(bait, k, 0)
    for seqs, want in ((m1, u16(GOLD["pe10k"]["k"][str(k)]["hits1_u16"])), (m2, u16(GOLD["pe10k"]["k"][str(k)]["hits2_u16"])),
                       (rg, u16(GOLD["ragged"]["k"][str(k)]))):
        reads = _gpu_reads(mf, seqs)
        n = len(seqs)
        for mode in (mf.MODE_SCREENED, mf.MODE_EXHAUSTIVE):
            b, h, _ = mf.filter_reads(ks, reads, 1, mode, want_hits=True)
            assert np.array_equal(h, want), (k, mode)
            for T in (1, 3):                                     # T = 1 without hit counts is the screen + finish pass
                b, _, st = mf.filter_reads(ks, reads, T, mode)
                assert np.array_equal(mf.unpack_bits(b, n), want >= T), (k, mode, T)
                assert st.n_pass == int((want >= T).sum())


@pytest.mark.gpu
@pytest.mark.parametrize("k", [21, 31, 41])
def test_gpu_file_pipeline_matches_golden(mf, inputs, tmp_path, k):
    bait, m1, m2, _ = inputs
    fq1, fq2 = str(tmp_path / "a_1.fq"), str(tmp_path / "a_2.fq")
    write_fastq(fq1, m1, "a")
    write_fastq(fq2, m2, "b")
    ks = mf.KmerSet.from_text(bait, k, 0)
    for T in (1, 3):
        g = GOLD["pe10k"]["k"][str(k)]["T"][str(T)]
        for mode, key in ((mf.PAIR_EITHER, "kept_either"), (mf.PAIR_BOTH, "kept_both")):
            kept, total = mf.filter_fastq_files(ks, fq1, fq2, str(tmp_path / "o1.fq"), str(tmp_path / "o2.fq"), T, mode)
            assert (kept, total) == (g[key], len(m1))
            assert open(tmp_path / "o1.fq").read().count("\n") == 4 * g[key]


@pytest.mark.gpu
@pytest.mark.parametrize("k", [11, 15, 21, 31, 32, 33, 41, 63])
def test_gpu_tables_match_golden(mf, inputs, k):
    ks = mf.KmerSet.from_text(inputs[0], k, 0)
    g = GOLD["tables"][str(k)]
    info = ks.info
    assert (info.slots, info.n_keys) == (g["slots"], g["n_keys"])
    assert hashlib.md5(np.ascontiguousarray(ks.export_table()).astype("<u8").tobytes()).hexdigest() == g["md5"]


@pytest.mark.gpu
@pytest.mark.parametrize("k", [11, 31])
def test_gpu_edge_cases_match_golden(mf, k):
    ks = mf.KmerSet.from_text(GOLD["edge"]["bait"], k, 0)
    reads = _gpu_reads(mf, GOLD["edge"]["reads"])
    for mode in (mf.MODE_SCREENED, mf.MODE_EXHAUSTIVE):
        _, h, _ = mf.filter_reads(ks, reads, 1, mode, want_hits=True)
        assert h.tolist() == GOLD["edge"]["k"][str(k)]
        b, _, _ = mf.filter_reads(ks, reads, 1, mode)
        assert mf.unpack_bits(b, len(GOLD["edge"]["reads"])).tolist() == [x >= 1 for x in GOLD["edge"]["k"][str(k)]]
