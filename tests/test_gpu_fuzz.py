"""Randomised differential test: the HIP path against the C oracle on many small random
configurations (k, bait shape, read-length mix, invalid bases, threshold).  Seeds are fixed, so a
failure reproduces; the assertion message carries the configuration."""
import os
import random

import numpy as np
import pytest

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


def _rand_dna(rng, n, junk=0.0):
    alphabet = "ACGT" if junk == 0 else "ACGT" * 12 + "NRYacgtn-"
    return "".join(rng.choices(alphabet, k=n))


def _reads_from(rng, recs, n, max_len, junk):
    comp = str.maketrans("ACGTacgt", "TGCAtgca")
    out = []
    for _ in range(n):
        mode = rng.random()
        L = rng.choice([0, 1, rng.randint(2, 40), rng.randint(20, max_len), 150, 150, 151, 100])
        if mode < 0.35 and recs:
            r = rng.choice(recs)
            if len(r) > 5:
                p = rng.randrange(0, len(r))
                s = r[p:p + L]
                if rng.random() < 0.5:
                    s = s[::-1].translate(comp)
                s = list(s)
                for j in range(len(s)):
                    if rng.random() < 0.01:
                        s[j] = rng.choice("ACGT")
                s = "".join(s)
            else:
                s = _rand_dna(rng, L)
        else:
            s = _rand_dna(rng, L, junk if rng.random() < 0.3 else 0.0)
        out.append(s)
    return out


@pytest.mark.parametrize("seed", range(int(os.environ.get("MF_FUZZ_SEEDS", "40"))))
def test_fuzz_nucleotide(mf, ol, seed):
    _fuzz_nucleotide(mf, ol, seed)


@pytest.mark.parametrize("seed", range(int(os.environ.get("MF_FUZZ_SEEDS", "40"))))
def test_fuzz_nucleotide_forced_screens(mf, ol, seed):
    """the same configurations with a form of the large-bait screen forced on them (LDS table + front2 turn by turn / queued, front2 alone, a
    one-bit LDS table in front of it; small and overloaded tables; canonical keys on and off)"""
    r2 = random.Random(11000 + seed)
    opts = {"front": r2.choice([0, 1, 2, 3, 4]), "canon": r2.choice([0, 1, 1]), "front2_log2b": r2.choice([0, 6, 7, 10]), "front3_log2b": r2.choice([-1, 0, 6, 9])}
    for n, v in opts.items():
        mf.set_option(n, v)
    try:
        _fuzz_nucleotide(mf, ol, seed, opts)
    finally:
        for n, v in (("front", -1), ("canon", -1), ("front2_log2b", 0), ("front3_log2b", -1)):
            mf.set_option(n, v)


def _fuzz_nucleotide(mf, ol, seed, opts=None):
    rng = random.Random(7000 + seed)
    k = rng.choice([11, 12, 15, 16, 17, 19, 20, 21, 22, 23, 24, 26, 27, 28, 29, 30, 31, 32, 33, 34, 40, 47, 48, 55, 62, 63])
    recs = [_rand_dna(rng, rng.choice([0, 5, k - 1, k, k + 1, 200, 1500, 6000]), junk=rng.choice([0.0, 0.0, 0.1]))
            for _ in range(rng.randint(1, 5))]
    if rng.random() < 0.3:
        recs.append("T" * rng.randint(k, 90))                 # homopolymer: the all-ones s-mer
    if rng.random() < 0.3:
        recs.append("AC" * rng.randint(k, k + 30))                # low complexity: many equal windows
    bait = "".join(f">r{i} x\n" + "\n".join(r[j:j + 70] for j in range(0, len(r), 70)) + "\n" for i, r in enumerate(recs))
    seqs = _reads_from(rng, [r for r in recs if len(r) >= k], rng.choice([1, 33, 500, 3000]), rng.choice([60, 400, 1200]), 0.1)
    thr = rng.choice([1, 1, 2, 3, 8])
    R = ol.OracleReads.from_seqs(seqs)
    t = ol.OracleTable(bait, k)
    obits, ohits = ol.filter_reads(t, R, thr, threads=4)
    ks = mf.KmerSet.from_text(bait, k)
    cfg = dict(seed=seed, k=k, n_reads=len(seqs), thr=thr, recs=[len(r) for r in recs], opts=opts)
    assert np.array_equal(ks.export_table(), t.keys), cfg
    reads = mf.Reads.from_packed(R.words, R.offsets, R.npos)
    for mode in (mf.MODE_SCREENED, mf.MODE_EXHAUSTIVE):
        bits, hits, _ = mf.filter_reads(ks, reads, thr, mode, want_hits=True)
        assert np.array_equal(hits, ohits), (cfg, mode)
        assert np.array_equal(bits, obits), (cfg, mode)
        assert np.array_equal(mf.filter_reads(ks, reads, thr, mode)[0], obits), (cfg, mode)


@pytest.mark.parametrize("seed", range(int(os.environ.get("MF_FUZZ_SEEDS", "40")) // 2))
def test_fuzz_protein(mf, ol, seed):
    from oracle import prot_bait_ref as pr
    rng = random.Random(9000 + seed)
    kp = rng.randint(4, 12)
    code = rng.choice([1, 2, 3, 4, 5, 9, 11, 13, 14, 21])
    prots = ["".join(rng.choices(pr.AA + ("X*b" if rng.random() < 0.3 else ""), k=rng.choice([0, 3, kp - 1, kp, 60, 400])))
             for _ in range(rng.randint(1, 6))]
    db = "".join(f">p{i}\n{p}\n" for i, p in enumerate(prots))
    genes = [pr.back_translate(p, code, rng) for p in prots]
    seqs = _reads_from(rng, [g for g in genes if len(g) >= 3 * kp], rng.choice([1, 40, 800]), rng.choice([60, 400, 1000]), 0.1)
    thr = rng.choice([1, 1, 2, 5])
    R = ol.OracleReads.from_seqs(seqs)
    t = ol.OracleTable(db, kp, protein=True)
    obits, ohits = ol.pfilter_reads(t, R, code, thr, threads=4)
    ks = mf.KmerSet.protein_from_text(db, kp, code)
    cfg = dict(seed=seed, kp=kp, code=code, n_reads=len(seqs), thr=thr, prots=[len(p) for p in prots])
    assert np.array_equal(ks.export_table(), t.keys), cfg
    reads = mf.Reads.from_packed(R.words, R.offsets, R.npos)
    bits, hits, _ = mf.filter_reads(ks, reads, thr, want_hits=True)
    assert np.array_equal(hits, ohits), cfg
    assert np.array_equal(bits, obits), cfg
    assert np.array_equal(mf.filter_reads(ks, reads, thr)[0], obits), cfg
