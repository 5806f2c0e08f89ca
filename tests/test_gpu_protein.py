"""GPU parity for protein-space baiting: the six-frame HIP kernel, through
the C ABI, against oracle/kmer_bait_oracle.c's Spec-P functions (pinned to oracle/prot_bait_ref.py).
PARITY UNPINNED BY THE REFERENCE (it only hands profile/MT_database to tblastn).  Bar: bit-exact."""
import os
import random
import subprocess

import numpy as np
import pytest

from tests.util_data import bits_to_bool, make_protein_bait, make_reads, write_fastq

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


@pytest.fixture(scope="module")
def db():
    return make_protein_bait()            # (protein FASTA, DNA of the genes), genetic code 5


@pytest.mark.parametrize("kp", [4, 5, 7, 9, 12])
def test_peptide_table_byte_identical(mf, ol, db, kp):
    ks = mf.KmerSet.protein_from_text(db[0], kp, 5)
    t = ol.OracleTable(db[0], kp, protein=True)
    info = ks.info
    assert (info.k, info.key_words, info.slots, info.n_keys, info.kind, info.genetic_code) == (kp, 1, t.slots, t.n_keys, mf.KIND_PROTEIN, 5)
    assert info.screen_s == 0
    assert np.array_equal(ks.export_table(), t.keys)


@pytest.mark.parametrize("kp,code", [(4, 1), (7, 5), (9, 2), (12, 9), (6, 14), (8, 4), (7, 13)])
@pytest.mark.parametrize("uniform", [False, True])
def test_six_frame_filter_matches_oracle(mf, ol, kp, code, uniform):
    prot_fa, gene_fa = make_protein_bait(code=code)
    seqs = make_reads(gene_fa, 5000, seed=kp * 31 + code, uniform=uniform, mito_frac=0.4)
    R = ol.OracleReads.from_seqs(seqs)
    t = ol.OracleTable(prot_fa, kp, protein=True)
    ks = mf.KmerSet.protein_from_text(prot_fa, kp, code)
    reads = mf.Reads.from_packed(R.words, R.offsets, R.npos)
    for thr in (1, 2, 9):
        obits, ohits = ol.pfilter_reads(t, R, code, thr, threads=4)
        bits, hits, st = mf.filter_reads(ks, reads, thr, want_hits=True)
        assert np.array_equal(hits, ohits), (kp, code, thr)
        assert np.array_equal(bits, obits), (kp, code, thr)
        bits2, _, st2 = mf.filter_reads(ks, reads, thr)                  # early-exit variant
        assert np.array_equal(bits2, obits), (kp, code, thr)
        assert st2.n_pass == int(bits_to_bool(obits, len(seqs)).sum())
    assert int(ohits.max()) > 20                                         # planted reads light up a whole frame


def test_edge_cases(mf, ol, db):
    ks = mf.KmerSet.protein_from_text(db[0], 7, 5)
    t = ol.OracleTable(db[0], 7, protein=True)
    seqs = ["", "A", "AC", "ACG", "ACGTACGTACGTACGTACGT", "N" * 30, "ACGTN" * 12, "acgtacgtacgtacgtacgtacgtacgt",
            "ACGTAC" * 400, "T" * 1000]
    R = ol.OracleReads.from_seqs(seqs)
    reads = mf.Reads.from_packed(R.words, R.offsets, R.npos)
    _, hits, _ = mf.filter_reads(ks, reads, 1, want_hits=True)
    assert np.array_equal(hits, ol.pfilter_reads(t, R, 5)[1])
    # an empty set, an empty read set, bad arguments
    empty = mf.KmerSet.protein_from_text(">tiny\nMLS\n", 7, 5)
    assert empty.info.n_keys == 0
    assert mf.filter_reads(empty, reads, 1)[0].sum() == 0
    none = mf.Reads.from_packed(np.zeros(0, np.uint32), np.zeros(1, np.uint64), np.zeros(0, np.uint64))
    assert mf.filter_reads(ks, none, 1)[2].n_pass == 0
    for kp, code in ((3, 5), (13, 5), (7, 7), (7, 0)):
        with pytest.raises(mf.MitoFilterError):
            mf.KmerSet.protein_from_text(db[0], kp, code)


def test_large_database_uses_global_bit_table(mf, ol):
    """A database the size of an MT_database clade: the k-mer bit table no longer fits LDS and the
    kernel variant that keeps it in L2 runs."""
    rng = random.Random(99)
    aas = "LSFIVGATMPYNWKEDHQRC"
    prots = ["".join(rng.choices(aas, k=400)) for _ in range(1200)]          # 480 k residues
    prot_fa = "".join(f">p{i}\n{p}\n" for i, p in enumerate(prots))
    from oracle import prot_bait_ref as pr
    gene_fa = "".join(f">g{i}\n{pr.back_translate(p, 5, rng)}\n" for i, p in enumerate(prots[:40]))
    ks = mf.KmerSet.protein_from_text(prot_fa, 8, 5)
    t = ol.OracleTable(prot_fa, 8, protein=True)
    assert ks.info.n_windows > (1 << 17) and ks.info.n_keys == t.n_keys
    assert np.array_equal(ks.export_table(), t.keys)
    seqs = make_reads(gene_fa, 40000, seed=17, uniform=True, mito_frac=0.2)
    R = ol.OracleReads.from_seqs(seqs)
    reads = mf.Reads.from_packed(R.words, R.offsets, R.npos)
    obits, ohits = ol.pfilter_reads(t, R, 5, 2, threads=os.cpu_count() or 1)
    bits, hits, _ = mf.filter_reads(ks, reads, 2, want_hits=True)
    assert np.array_equal(hits, ohits) and np.array_equal(bits, obits)
    assert np.array_equal(mf.filter_reads(ks, reads, 2)[0], obits)


def test_synthetic_pe150_matches_oracle(mf, ol, db):
    """The generator's uniform 150-base layout (what the timing runs use), reads planted from the genes' DNA."""
    n, L = 300_000, 150
    ks = mf.KmerSet.protein_from_text(db[0], 9, 5)
    reads = mf.Reads.synth(n, L, seed=11, bait_text=db[1], keep_host=True)
    off = np.arange(n + 1, dtype=np.uint64) * L
    R = ol.OracleReads.from_arrays(reads.host_words, off, reads.host_npos)
    t = ol.OracleTable(db[0], 9, protein=True)
    obits, ohits = ol.pfilter_reads(t, R, 5, 1, threads=os.cpu_count() or 1)
    bits, hits, st = mf.filter_reads(ks, reads, 1, want_hits=True)
    assert np.array_equal(hits, ohits) and np.array_equal(bits, obits)
    assert np.array_equal(mf.filter_reads(ks, reads, 1)[0], obits)
    n_pass = int(bits_to_bool(obits, n).sum())
    assert 0.002 * n < n_pass < 0.008 * n            # ~0.5 % planted, a few lost to substitutions at the ends


def test_fastq_files_and_cli(mf, ol, db, tmp_path):
    prot_fa, gene_fa = db
    fq1, fq2 = str(tmp_path / "a_1.fq"), str(tmp_path / "a_2.fq.gz")
    write_fastq(fq1, make_reads(gene_fa, 4000, seed=21), "a", trailing_partial=True)
    write_fastq(fq2, make_reads(gene_fa, 4000, seed=22), "b", crlf=True, gz=True)
    bait = str(tmp_path / "db.fa")
    open(bait, "w").write(prot_fa)
    ks = mf.KmerSet.protein_from_fasta(bait, 8, 5)
    for pair_mode in (mf.PAIR_EITHER, mf.PAIR_BOTH):
        o1, o2, g1, g2 = (str(tmp_path / n) for n in ("o1.fq", "o2.fq", "g1.fq", "g2.fq"))
        ok, ot = ol.pfilter_fastq_files(bait, 8, 5, 1, pair_mode, fq1, fq2, o1, o2, threads=2)
        gk, gt = mf.filter_fastq_files(ks, fq1, fq2, g1, g2, 1, pair_mode)
        assert (gk, gt) == (ok, ot) and ot == 4000 and ok > 0
        assert open(g1, "rb").read() == open(o1, "rb").read() and open(g2, "rb").read() == open(o2, "rb").read()
    # the CLI personality: same stdout contract (one integer)
    from mitoflex_amd.assemble import assemble_wrapper as w
    exe = w.MEGAHIT().FAST_FILTER
    c1 = str(tmp_path / "c1.fq")
    out = subprocess.check_output([exe, "bait", "--protein", "--code", "5", "-k", "8", "--bait", bait, "--fq1", fq1, "--out1", c1])
    ok, _ = ol.pfilter_fastq_files(bait, 8, 5, 1, 0, fq1, None, str(tmp_path / "o_se.fq"), None)
    assert int(out) == ok
    assert open(c1, "rb").read() == open(tmp_path / "o_se.fq", "rb").read()


def test_full_size_properties(mf, ol, db):
    """5 Gbp (BASELINE.json configs[1] shape) in residue space: determinism, the planted fraction, and a
    1 M-read window from the middle against the oracle."""
    n, L = 33_333_334, 150
    ks = mf.KmerSet.protein_from_text(db[0], 9, 5)
    reads = mf.Reads.synth(n, L, seed=20261003, bait_text=db[1], keep_host=True)
    b1, _, st1 = mf.filter_reads(ks, reads, 1)
    b2, _, st2 = mf.filter_reads(ks, reads, 1)
    assert np.array_equal(b1, b2) and st1.n_pass == st2.n_pass
    assert 0.002 * n < st1.n_pass < 0.008 * n
    first, cnt = 16_000_000, 1_000_000                      # multiple of 32: the window's bits are whole words
    off = np.arange(n + 1, dtype=np.uint64) * L
    R = ol.OracleReads.from_arrays(reads.host_words, off, reads.host_npos)
    t = ol.OracleTable(db[0], 9, protein=True)
    obits, _ = ol.pfilter_reads(t, R, 5, 1, first=first, count=cnt, threads=os.cpu_count() or 1)
    assert np.array_equal(b1[first // 32:(first + cnt) // 32], obits)
    reads.close()
