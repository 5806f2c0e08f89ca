"""GPU parity of the large-bait screens (screen2_kernel: front_mode 1 = LDS table + bait-sized front2 in L2 asked turn by turn, front_mode 2 =
front2 (+ front3) only, front_mode 4 = a one-bit LDS table in front of mode 2's look-ups; screen3_kernel: front_mode 3 = LDS table, lone positives queued and asked sixty-four at a time): every form is forced onto small inputs through mf_set_option and held bit-exact to the CPU oracle, through the threshold-1
pass (screen + finish), the candidate-bitmap pass (mark + exact, hit counts) and at every screen geometry; then baits that really
are large (100 kbp .. 2 Mbp), where the library picks the form itself.

PARITY UNPINNED BY THE REFERENCE (SURVEY.md 8a group B): the oracle is this repository's own.  Bar: bit-exact.
"""
import os

import numpy as np
import pytest

from tests.util_data import bits_to_bool, make_reads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mf(built_lib):
    from mitoflex_amd import mitofilter
    if mitofilter.device_count() < 1:
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    return mitofilter


@pytest.fixture(scope="module")
def ol():
    from oracle import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture()
def front(mf):
    """set the front options for the sets built inside a test; back to automatic afterwards"""
    def set_(mode, f2=0, f3=-1, canon=-1):
        mf.set_option("front", mode)
        mf.set_option("front2_log2b", f2)
        mf.set_option("front3_log2b", f3)
        mf.set_option("canon", canon)
    yield set_
    set_(-1, 0, -1, -1)


# (mode, front2 log2 blocks, front3 log2 blocks): roomy tables, overloaded front2 (nearly everything passes: what a round cannot
# verify is passed on), a front3 behind an overloaded front2, the smallest tables there are
FORMS = [(1, 0, -1), (1, 6, -1), (2, 0, -1), (2, 6, 0), (2, 6, 12), (2, 8, 6), (3, 0, -1), (3, 6, -1), (4, 0, -1), (4, 6, 8)]
# ... and with the screen's tables built canonical (one key per bait s-mer, the samples made canonical in the screen: six instructions for the
# 16-base samples of k >= 31, eight for the shorter ones below), the plain LDS-table screen included
FORMS += [(0, 0, -1, 1), (1, 0, -1, 1), (1, 6, -1, 1), (2, 0, -1, 1), (2, 6, 12, 1), (3, 0, -1, 1), (3, 6, -1, 1), (4, 0, -1, 1), (4, 6, 8, 1)]


@pytest.mark.parametrize("k", [19, 21, 25, 28, 31, 32, 33, 41, 63])
@pytest.mark.parametrize("form", FORMS)
def test_forced_forms_match_oracle(mf, ol, bait_text, front, k, form):
    front(*form)
    ks = mf.KmerSet.from_text(bait_text, k)
    assert ks.info.canonical_screen == (1 if len(form) > 3 and form[3] == 1 else 0)          # (sixteen-base samples for k >= 31, shorter ones below)
    t = ol.OracleTable(bait_text, k)
    for uniform in (True, False):
        seqs = make_reads(bait_text, 5000, seed=300 + k, uniform=uniform)
        R = ol.OracleReads.from_seqs(seqs)
        reads = mf.Reads.from_packed(R.words, R.offsets, R.npos)
        for thr in (1, 2):
            obits, ohits = ol.filter_reads(t, R, thr, threads=4)
            bits, hits, _ = mf.filter_reads(ks, reads, thr, mf.MODE_SCREENED, want_hits=True)
            assert np.array_equal(hits, ohits), (k, form, uniform, thr)
            assert np.array_equal(bits, obits), (k, form, uniform, thr)
            bits2, _, st = mf.filter_reads(ks, reads, thr, mf.MODE_SCREENED)
            assert np.array_equal(bits2, obits), (k, form, uniform, thr)
            assert st.n_pass == int(bits_to_bool(obits, len(seqs)).sum())
        reads.close()
    ks.close()


@pytest.mark.parametrize("k", [17, 19, 21, 25, 27])
@pytest.mark.parametrize("form", [(0, 0, -1), (1, 6, -1), (3, 0, -1), (4, 6, 8), (2, 6, 12), (0, 0, -1, 1), (1, 6, -1, 1), (4, 6, 8, 1), (2, 6, 12, 1)])
def test_stride8_sets_through_the_finish_kernels(mf, ol, bait_text, front, k, form):
    """k < 28 (a sample every 8 bases): the threshold-1 pass through screen + finish -- what sets of baits beyond ~20 kbp take -- forced on small inputs,
    every screen form, uniform and ragged reads, single and pipelined passes: bits equal the oracle's"""
    front(*form)
    mf.set_option("s8_finish", 1)
    try:
        ks = mf.KmerSet.from_text(bait_text, k)
        t = ol.OracleTable(bait_text, k)
        for uniform in (True, False):
            seqs = make_reads(bait_text, 6000, seed=500 + k, uniform=uniform)
            R = ol.OracleReads.from_seqs(seqs)
            reads = mf.Reads.from_packed(R.words, R.offsets, R.npos)
            obits, _ = ol.filter_reads(t, R, 1, threads=4)
            want = int(bits_to_bool(obits, len(seqs)).sum())
            bits, _, st = mf.filter_reads(ks, reads, 1, mf.MODE_SCREENED)
            assert np.array_equal(bits, obits), (k, form, uniform)
            assert st.n_pass == want
            per, _ = mf.filter_resident_passes(ks, reads, 1, mf.MODE_SCREENED, 5)
            assert [int(x) for x in per] == [want] * 5, (k, form, uniform)
            reads.close()
        ks.close()
    finally:
        mf.set_option("s8_finish", -1)


@pytest.mark.parametrize("form", FORMS)
def test_forced_forms_pipelined_passes(mf, ol, bait_text, front, form):
    """several pipelined passes of a 300 k-read synthetic set: every pass's tally equals the oracle's count (buffer sets, streams)"""
    front(*form)
    n, L = 300_000, 150
    ks = mf.KmerSet.from_text(bait_text, 31)
    reads = mf.Reads.synth(n, L, seed=9, bait_text=bait_text, keep_host=True)
    off = np.arange(n + 1, dtype=np.uint64) * L
    R = ol.OracleReads.from_arrays(reads.host_words, off, reads.host_npos)
    obits, _ = ol.filter_reads(ol.OracleTable(bait_text, 31), R, 1, threads=os.cpu_count() or 1)
    want = int(bits_to_bool(obits, n).sum())
    per, _ = mf.filter_resident_passes(ks, reads, 1, mf.MODE_SCREENED, 6)
    assert [int(x) for x in per] == [want] * 6
    bits, _, _ = mf.filter_reads(ks, reads, 1, mf.MODE_SCREENED)
    assert np.array_equal(bits, obits)


@pytest.mark.parametrize("size,k", [(30_000, 31), (30_000, 21), (50_000, 31), (70_000, 31), (100_000, 31), (100_000, 21), (100_000, 29), (200_000, 31), (350_000, 21), (350_000, 31), (350_000, 41), (1_000_000, 31), (2_000_000, 31),
                                    (4_500_000, 31)])
def test_large_baits_pick_their_screen(mf, ol, size, k):
    """baits that are large for real: the library picks the form -- for 16-base samples (k >= 31) canonical keys from ~40 kbp, with them the LDS table and a
    queue of lone positives up to ~120 kbp, LDS table + front2 turn by turn up to ~210 kbp, a one-bit LDS table + front2 up to ~2 Mbp, front2 + front3
    alone beyond; bits and hit counts equal the oracle's on 200 k reads (0.5 % bait reads, N), and the screened pass equals the exhaustive one"""
    from mitoflex_amd.utility.synth_bait import random_bait
    bait = random_bait(size, seed=size + k)
    ks = mf.KmerSet.from_text(bait, k)
    want_mode = {30_000: 3 if k >= 28 else 0, 50_000: 3 if k >= 28 else 1, 70_000: 3, 100_000: 3 if k >= 28 else 1, 200_000: 1, 350_000: 4, 1_000_000: 4, 2_000_000: 4, 4_500_000: 2}[size]
    assert ks.info.front_mode == want_mode, (size, k, ks.info.front_mode)
    assert ks.info.canonical_screen == (1 if size >= (70_000 if k >= 28 else 100_000) else 0)
    assert (ks.info.front3_log2_blocks > 0) == (size >= 2_000_000)
    n, L = 200_000, 150
    reads = mf.Reads.synth(n, L, seed=size, bait_text=bait, keep_host=True)
    off = np.arange(n + 1, dtype=np.uint64) * L
    R = ol.OracleReads.from_arrays(reads.host_words, off, reads.host_npos)
    t = ol.OracleTable(bait, k)
    for thr in (1, 3):
        obits, ohits = ol.filter_reads(t, R, thr, threads=os.cpu_count() or 1)
        bits, hits, _ = mf.filter_reads(ks, reads, thr, mf.MODE_SCREENED, want_hits=True)
        assert np.array_equal(hits, ohits), (size, k, thr)
        assert np.array_equal(bits, obits), (size, k, thr)
        b2, _, _ = mf.filter_reads(ks, reads, thr, mf.MODE_SCREENED)
        assert np.array_equal(b2, obits), (size, k, thr)
        b3, _, _ = mf.filter_reads(ks, reads, thr, mf.MODE_EXHAUSTIVE)
        assert np.array_equal(b3, obits), (size, k, thr)
    assert 0.003 * n < int(bits_to_bool(obits, n).sum())


def test_bait_rich_input_through_the_fronts(mf, ol, bait_text, front):
    """every read a bait read: all of a wave's samples are positives, far more than its queue holds -- the overflow is passed on unverified"""
    for form in [(1, 0, -1), (2, 0, -1), (3, 0, -1), (4, 0, -1), (0, 0, -1, 1), (1, 0, -1, 1), (3, 0, -1, 1), (4, 0, -1, 1)]:
        front(*form)
        ks = mf.KmerSet.from_text(bait_text, 31)
        n, L = 100_000, 150
        reads = mf.Reads.synth(n, L, seed=3, bait_text=bait_text, mito_ppm=1_000_000, keep_host=True)
        off = np.arange(n + 1, dtype=np.uint64) * L
        R = ol.OracleReads.from_arrays(reads.host_words, off, reads.host_npos)
        obits, _ = ol.filter_reads(ol.OracleTable(bait_text, 31), R, 1, threads=os.cpu_count() or 1)
        for _ in range(3):                      # (the pass kind adapts to what the last call saw)
            bits, _, _ = mf.filter_reads(ks, reads, 1, mf.MODE_SCREENED)
            assert np.array_equal(bits, obits), form


@pytest.mark.parametrize("ppm", [30_000, 200_000, 600_000])
def test_canonical_queue_keeps_within_the_record_lists(mf, ol, bait_text, front, ppm):
    """canonical keys, the queued form: a slice's positives are queued one by one while the wave is well under one record per lane and
    chunk, and recorded as a slice when it is not -- inputs between sparse and bait-rich, small LDS table and front2 (many false positives
    of both), bits equal to the oracle on every pass"""
    front(3, 6, -1, 1)
    mf.set_option("adapt", 0)
    try:
        ks = mf.KmerSet.from_text(bait_text, 31)
        assert ks.info.canonical_screen == 1 and ks.info.front_mode == 3
        n, L = 400_000, 150
        reads = mf.Reads.synth(n, L, seed=ppm, bait_text=bait_text, mito_ppm=ppm, sub_ppm=30_000, keep_host=True)
        off = np.arange(n + 1, dtype=np.uint64) * L
        R = ol.OracleReads.from_arrays(reads.host_words, off, reads.host_npos)
        obits, _ = ol.filter_reads(ol.OracleTable(bait_text, 31), R, 1, threads=os.cpu_count() or 1)
        want = int(bits_to_bool(obits, n).sum())
        bits, _, _ = mf.filter_reads(ks, reads, 1, mf.MODE_SCREENED)
        assert np.array_equal(bits, obits)
        per, _ = mf.filter_resident_passes(ks, reads, 1, mf.MODE_SCREENED, 4)
        assert [int(x) for x in per] == [want] * 4
    finally:
        mf.set_option("adapt", 1)


@pytest.mark.parametrize("size,k", [(100_000, 31), (350_000, 31), (100_000, 21), (2_500_000, 31)])
def test_large_baits_through_the_file_level_call(mf, ol, tmp_path, monkeypatch, size, k):
    """the drop-in boundary with a large bait: mf_filter_fastq_files on a .gz pair (device ingest path: one pass a batch) and on the host pipeline,
    kept / total and the output bytes equal to the oracle's"""
    from mitoflex_amd.utility.synth_bait import random_bait
    from tests.util_data import write_fastq
    bait_text = random_bait(size, seed=size + k)
    bait = str(tmp_path / "bait.fa")
    open(bait, "w").write(bait_text)
    s1 = make_reads(bait_text[:400_000], 30_000, seed=41)
    s2 = make_reads(bait_text[:400_000], 30_000, seed=42)
    fq1, fq2 = str(tmp_path / "a_1.fq.gz"), str(tmp_path / "a_2.fq.gz")
    write_fastq(fq1, s1, "a", gz=True)
    write_fastq(fq2, s2, "b", gz=True)
    o1, o2 = str(tmp_path / "o1.fq"), str(tmp_path / "o2.fq")
    ok, ot = ol.filter_fastq_files(bait, k, 1, mf.PAIR_EITHER, fq1, fq2, o1, o2, threads=os.cpu_count() or 1)
    assert ot == 30_000 and 0 < ok < ot
    ks = mf.KmerSet.from_fasta(bait, k)
    assert ks.info.front_mode != 0
    for path in ("device", "host"):
        monkeypatch.setenv("MF_INGEST", path)
        g1, g2 = str(tmp_path / f"g1_{path}.fq"), str(tmp_path / f"g2_{path}.fq")
        gk, gt = mf.filter_fastq_files(ks, fq1, fq2, g1, g2, 1, mf.PAIR_EITHER)
        assert (gk, gt) == (ok, ot), (size, k, path)
        assert open(g1, "rb").read() == open(o1, "rb").read() and open(g2, "rb").read() == open(o2, "rb").read(), (size, k, path)


def test_full_size_set_through_a_saturated_lds_table(mf, ol):
    """configs[1]'s 33.3 M reads against a 350 kbp bait (the 128 KiB LDS table would pass three quarters of the samples: the library leaves it
    out and looks every sample up in front2): the screened pass equals the exhaustive one on every one of the 33.3 M bits, both equal the
    oracle on a 1.5 M-read window, and consecutive pipelined passes tally the same."""
    from mitoflex_amd.utility.synth_bait import random_bait
    bait = random_bait(350_000, seed=350_000)
    ks = mf.KmerSet.from_text(bait, 31)
    assert ks.info.front_mode == 4          # (a one-bit LDS table in front of the look-ups; 2 Mbp and beyond take mode 2: test_large_baits_pick_their_screen)
    n, L = 33_333_334, 150
    reads = mf.Reads.synth(n, L, seed=20261003, bait_text=bait, keep_host=True)
    b1, _, s1 = mf.filter_reads(ks, reads, 1, mf.MODE_SCREENED)
    b2, _, s2 = mf.filter_reads(ks, reads, 1, mf.MODE_EXHAUSTIVE)
    assert np.array_equal(b1, b2) and s1.n_pass == s2.n_pass
    per, _ = mf.filter_resident_passes(ks, reads, 1, mf.MODE_SCREENED, 5)
    assert [int(x) for x in per] == [int(s1.n_pass)] * 5
    n_win = 1_500_000 // 32 * 32
    off = np.arange(n_win + 1, dtype=np.uint64) * L
    lim = n_win * L
    R = ol.OracleReads.from_arrays(reads.host_words[:lim // 16 + 8], off, reads.host_npos[reads.host_npos < lim])
    obits, _ = ol.filter_reads(ol.OracleTable(bait, 31), R, 1, threads=os.cpu_count() or 1)
    assert np.array_equal(b1[:n_win // 32], obits[:n_win // 32])
    assert 0.003 * n < int(s1.n_pass) < 0.008 * n
