"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

PARITY UNPINNED BY THE REFERENCE for these rows (SURVEY.md 8a group B): the
oracle is this repo's own (oracle/kmer_bait_oracle.c pinned to
oracle/kmer_bait_ref.py).  Bar: bit-exact (integer work).
"""
import os

import numpy as np
import pytest

from tests.util_data import bits_to_bool, make_reads, write_fastq

pytestmark = pytest.mark.gpu
# the library with the test hooks compiled in (MF_FAKE_DEVICES, MF_DEVPOOL_FAIL_AT): the shipped one carries neither
HOOKS_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mitoflex_amd", "libmitofilter_hip_hooks.so")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


@pytest.mark.parametrize("k", [11, 15, 19, 21, 23, 27, 31, 32, 33, 41, 63])
def test_table_byte_identical(mf, ol, bait_text, k):
    """Device-built table == oracle table, byte for byte (history-independent layout)."""
    ks = mf.KmerSet.from_text(bait_text, k)
    t = ol.OracleTable(bait_text, k)
    info = ks.info
    assert (info.k, info.key_words, info.slots, info.n_keys) == (k, t.kw, t.slots, t.n_keys)
    assert np.array_equal(ks.export_table(), t.keys)


@pytest.mark.parametrize("k", [15, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 41, 63])
@pytest.mark.parametrize("uniform", [False, True])
def test_filter_matches_oracle(mf, ol, bait_text, k, uniform):
    seqs = make_reads(bait_text, 6000, seed=100 + k, uniform=uniform)
    R = ol.OracleReads.from_seqs(seqs)
    t = ol.OracleTable(bait_text, k)
    ks = mf.KmerSet.from_text(bait_text, k)
    reads = mf.Reads.from_packed(R.words, R.offsets, R.npos)
    assert reads.info.uniform_len == (150 if uniform else 0)
    for thr in (1, 2, 7):
        obits, ohits = ol.filter_reads(t, R, thr, threads=4)
        for mode in (mf.MODE_SCREENED, mf.MODE_EXHAUSTIVE):
            bits, hits, st = mf.filter_reads(ks, reads, thr, mode, want_hits=True)
            assert np.array_equal(hits, ohits), (k, thr, mode)
            assert np.array_equal(bits, obits), (k, thr, mode)
            bits2, _, st2 = mf.filter_reads(ks, reads, thr, mode)          # early-exit variant
            assert np.array_equal(bits2, obits), (k, thr, mode)
            assert st2.n_pass == int(bits_to_bool(obits, len(seqs)).sum())
    # one-shot host-buffer entry point
    assert np.array_equal(mf.filter_packed(ks, R.words, R.offsets, R.npos, 1), ol.filter_reads(t, R, 1)[0])


def test_edge_cases(mf, ol, bait_text):
    k = 31
    t = ol.OracleTable(bait_text, k)
    ks = mf.KmerSet.from_text(bait_text, k)
    from tests.util_data import bait_records
    g = bait_records(bait_text)[0]
    cases = {
        "empty_set": [],
        "all_empty_reads": ["", "", ""],
        "one_exact_kmer": [g[100:131]],
        "one_short": [g[100:130]],
        "n_in_middle": [g[100:131][:15] + "N" + g[100:131][16:]],
        "n_breaks_only_some": [g[200:300][:50] + "N" + g[200:300][51:]],
        "polyT": ["T" * 150, "A" * 150, "T" * 31, "A" * 16],
        "long_read": [g[1000:6000]],
        "33_reads": [g[i * 10:i * 10 + 150] for i in range(33)],
        "straddle": ["ACGT" * 3 + g[7000:7017], g[7017:7040] + "ACGT" * 30],
    }
    for name, seqs in cases.items():
        R = ol.OracleReads.from_seqs(seqs)
        reads = mf.Reads.from_packed(R.words, R.offsets, R.npos)
        obits, ohits = ol.filter_reads(t, R, 1)
        for mode in (mf.MODE_SCREENED, mf.MODE_EXHAUSTIVE):
            bits, hits, _ = mf.filter_reads(ks, reads, 1, mode, want_hits=True)
            assert np.array_equal(bits, obits), (name, mode)
            if seqs:
                assert np.array_equal(hits, ohits), (name, mode)


def test_synth_uniform_matches_oracle(mf, ol, bait_text):
    """Library's own synthetic generator (bench input shape) checked against the oracle."""
    n, L = 300_000, 150
    ks = mf.KmerSet.from_text(bait_text, 31)
    reads = mf.Reads.synth(n, L, seed=7, bait_text=bait_text, keep_host=True)
    off = np.arange(n + 1, dtype=np.uint64) * L
    R = ol.OracleReads.from_arrays(reads.host_words, off, reads.host_npos)
    t = ol.OracleTable(bait_text, 31)
    obits, ohits = ol.filter_reads(t, R, 1, threads=os.cpu_count() or 1)
    for mode in (mf.MODE_SCREENED, mf.MODE_EXHAUSTIVE):
        bits, hits, st = mf.filter_reads(ks, reads, 1, mode, want_hits=True)
        assert np.array_equal(hits, ohits)
        assert np.array_equal(bits, obits)
    n_pass = int(bits_to_bool(obits, n).sum())
    assert 0.003 * n < n_pass < 0.008 * n            # ~0.5 % bait reads


@pytest.mark.parametrize("batch", [None, 700, 1])
@pytest.mark.parametrize("gz", [False, True])
@pytest.mark.parametrize("ingest", ["host", "default"])
def test_fastq_files_match_oracle(mf, ol, bait_text, tmp_path, gz, batch, ingest, monkeypatch):
    """File level, through the chunked host pipeline (MF_INGEST=host: regular files take the device ingest path by default since round 5,
    tests/test_gpu_devingest.py) and through whatever the library picks; small batch sizes force many batches, carried partial
    records and out-of-step mate files."""
    if ingest == "host":
        monkeypatch.setenv("MF_INGEST", "host")
    elif batch is not None:
        pytest.skip("the batch knobs are the host pipeline's")
    if batch is not None:
        if gz and batch == 1:
            pytest.skip("one-read batches are exercised on the plain files")
        monkeypatch.setenv("MF_BATCH_READS", str(batch))
        # plain files are mapped and parsed in parallel segments: tiny segments put a border inside
        # almost every record (headers, sequences, CRLF pairs, the partial tail)
        monkeypatch.setenv("MF_PARSE_SEG", "997" if batch == 700 else "64")
    ext = ".fq.gz" if gz else ".fq"
    n1 = 3000 if batch != 1 else 150
    s1 = make_reads(bait_text, n1, seed=1)
    s2 = make_reads(bait_text, n1 + 100, seed=2)      # longer mate file: zipped to the shorter
    fq1, fq2 = str(tmp_path / ("a_1" + ext)), str(tmp_path / ("a_2" + ext))
    write_fastq(fq1, s1, "a", crlf=False, trailing_partial=True, gz=gz)
    write_fastq(fq2, s2, "b", crlf=True, gz=gz)
    bait = str(tmp_path / "bait.fa")
    open(bait, "w").write(bait_text)
    ks = mf.KmerSet.from_fasta(bait, 31)
    for pair_mode in (mf.PAIR_EITHER, mf.PAIR_BOTH):
        o1, o2 = str(tmp_path / "o1.fq"), str(tmp_path / "o2.fq")
        g1, g2 = str(tmp_path / ("g1" + ext)), str(tmp_path / ("g2" + ext))
        ok, ot = ol.filter_fastq_files(bait, 31, 1, pair_mode, fq1, fq2, o1, o2, threads=2)
        gk, gt = mf.filter_fastq_files(ks, fq1, fq2, g1, g2, 1, pair_mode)
        assert (gk, gt) == (ok, ot) and ot == n1
        import gzip
        rd = (lambda p: gzip.open(p, "rb").read()) if gz else (lambda p: open(p, "rb").read())
        assert rd(g1) == open(o1, "rb").read()
        assert rd(g2) == open(o2, "rb").read()
    # single end
    o1, g1 = str(tmp_path / "se_o.fq"), str(tmp_path / "se_g.fq")
    ok, ot = ol.filter_fastq_files(bait, 31, 2, 0, fq1, None, o1, None)
    gk, gt = mf.filter_fastq_files(ks, fq1, None, g1, None, 2)
    assert (gk, gt) == (ok, ot)
    assert open(g1, "rb").read() == open(o1, "rb").read()
    # reads straight from a FASTQ file
    reads = mf.Reads.from_fastq(fq1)
    R = ol.OracleReads.from_fastq(fq1)
    assert reads.info.n_reads == R.n_reads == n1
    t = ol.OracleTable(bait_text, 31)
    assert np.array_equal(mf.filter_reads(ks, reads, 1)[0], ol.filter_reads(t, R, 1)[0])


def test_fifo_handoff_to_buildlib(mf, ol, bait_text, tmp_path, monkeypatch):
    """MEGAHIT.build_lib() with bait_fifo streams the survivors of the real HIP filter
    to `megahit_core buildlib` through named pipes; what the (stand-in) reader receives is byte for byte
    what the oracle writes to files."""
    from mitoflex_amd.assemble import assemble_wrapper as w
    from tests.test_callsite import FAKE_CORE
    core = tmp_path / "megahit_core"
    core.write_text(FAKE_CORE)
    core.chmod(0o755)
    fq1, fq2 = str(tmp_path / "a_1.fq"), str(tmp_path / "a_2.fq")
    write_fastq(fq1, make_reads(bait_text, 20000, seed=11), "a")
    write_fastq(fq2, make_reads(bait_text, 20000, seed=12), "b")
    bait = str(tmp_path / "bait.fa")
    open(bait, "w").write(bait_text)
    monkeypatch.setattr(w.MEGAHIT, "MEGAHIT_CORE", str(core))
    monkeypatch.setattr(w.a_conf, "bait_fasta", bait)
    monkeypatch.setattr(w.a_conf, "bait_fifo", True)
    for pe in (True, False):
        t = tmp_path / ("pe" if pe else "se")
        t.mkdir()
        m = w.MEGAHIT(fq1=fq1, fq2=fq2 if pe else None, temp_dir=str(t), read_lib=str(t / "reads.lib"))
        m.build_lib()
        o1, o2 = str(t / "o1.fq"), str(t / "o2.fq")
        ok, _ = ol.filter_fastq_files(bait, 31, 1, 0, fq1, fq2 if pe else None, o1, o2 if pe else None, threads=2)
        assert m.bait_kept == ok and ok > 0
        assert open(str(t / "reads.lib.m1"), "rb").read() == open(o1, "rb").read()
        if pe:
            assert open(str(t / "reads.lib.m2"), "rb").read() == open(o2, "rb").read()


# --------------------------------------------------------------------------- full-size properties
FULL = 33_333_334          # BASELINE.json configs[1]: 5 Gbp of 150-base reads


@pytest.fixture(scope="module")
def full_set(mf, bait_text):
    reads = mf.Reads.synth(FULL, 150, seed=20261003, bait_text=bait_text, keep_host=True)
    yield reads
    reads.close()


@pytest.mark.parametrize("k", [31, 21, 41])
def test_full_size_screened_equals_exhaustive(mf, ol, bait_text, full_set, k):
    """At the benchmark's full size the oracle is too slow to run on everything, so: (1) the
    screened pipeline and the brute-force exhaustive kernel must agree on every bit, (2) two runs
    agree (determinism), (3) a 1.5 M-read window from the middle equals the oracle, (4) the number of
    passing reads is what the generator planted (about 0.5 %)."""
    ks = mf.KmerSet.from_text(bait_text, k)
    b_s, _, st_s = mf.filter_reads(ks, full_set, 1, mf.MODE_SCREENED)
    b_e, _, st_e = mf.filter_reads(ks, full_set, 1, mf.MODE_EXHAUSTIVE)
    assert np.array_equal(b_s, b_e)
    assert st_s.n_pass == st_e.n_pass == int(np.unpackbits(b_s.view(np.uint8)).sum())
    b_s2, _, _ = mf.filter_reads(ks, full_set, 1, mf.MODE_SCREENED)
    assert np.array_equal(b_s, b_s2)
    assert 0.003 * FULL < st_s.n_pass < 0.007 * FULL
    # (fused pass, threshold 1: n_candidates counts stage-1 positives that went through exact counting, not reads)
    assert st_s.n_candidates >= st_s.n_pass and st_s.n_candidates < 0.1 * FULL
    # oracle on a window of whole bitmap words in the middle of the set
    first, count = 16_000_000, 1_500_000
    assert first % 32 == 0
    L = 150
    # the window as its own packed set: re-base the stream at a word boundary (first*L*2 bits is a multiple of 32)
    w0 = first * L // 16
    words = full_set.host_words[w0: w0 + (count * L + 15) // 16 + 8]
    npos = full_set.host_npos
    npos = npos[(npos >= first * L) & (npos < (first + count) * L)] - np.uint64(first * L)
    off = np.arange(count + 1, dtype=np.uint64) * L
    R = ol.OracleReads.from_arrays(words, off, npos)
    obits, _ = ol.filter_reads(ol.OracleTable(bait_text, k), R, 1, threads=os.cpu_count() or 1)
    assert np.array_equal(b_s[first // 32: (first + count) // 32], obits[: count // 32])


def test_full_size_thresholds_are_monotone(mf, bait_text, full_set):
    """pass(T+1) is a subset of pass(T); exhaustive and screened agree for T = 3 too."""
    ks = mf.KmerSet.from_text(bait_text, 31)
    prev = None
    for thr in (1, 3, 20, 121):
        b, _, st = mf.filter_reads(ks, full_set, thr, mf.MODE_SCREENED)
        if thr == 3:
            assert np.array_equal(b, mf.filter_reads(ks, full_set, thr, mf.MODE_EXHAUSTIVE)[0])
        if prev is not None:
            assert not np.any(b & ~prev)
        prev = b
    assert st.n_pass == 0                      # a 150-base read has only 120 windows


def test_shard_of_50gbp_config(mf, bait_text):
    """BASELINE.json configs[3]: 50 Gbp over 8 GPUs = 41 666 667 reads per GPU; one such shard."""
    n = 41_666_667
    ks = mf.KmerSet.from_text(bait_text, 31)
    reads = mf.Reads.synth(n, 150, seed=5, bait_text=bait_text)
    b_s, _, st_s = mf.filter_reads(ks, reads, 1, mf.MODE_SCREENED)
    b_e, _, st_e = mf.filter_reads(ks, reads, 1, mf.MODE_EXHAUSTIVE)
    assert np.array_equal(b_s, b_e) and st_s.n_pass == st_e.n_pass and 0.003 * n < st_s.n_pass < 0.007 * n


def test_ragged_two_million(mf, ol, bait_text):
    """Variable-length reads (offsets path, binary search on positives) at a size that matters."""
    rng = np.random.default_rng(3)
    n = 2_000_000
    lens = rng.integers(0, 260, size=n).astype(np.uint64)
    lens[rng.integers(0, n, size=1000)] = 0
    off = np.zeros(n + 1, dtype=np.uint64); off[1:] = np.cumsum(lens)
    total = int(off[-1])
    words = rng.integers(0, 2**32, size=(total + 15) // 16 + 8, dtype=np.uint32)
    words[(total + 15) // 16:] = 0
    # plant bait-derived reads: copy bait sequence over ~1 % of reads
    from tests.util_data import bait_records
    g = bait_records(bait_text)[0]
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    gb = np.array([code[c] for c in g], dtype=np.uint64)
    big = words.view(np.uint32)
    for r in rng.integers(0, n, size=20000):
        L = int(lens[r])
        if L < 40:
            continue
        p = int(rng.integers(0, len(g) - L))
        b0 = int(off[r])
        for j in range(L):
            gi = b0 + j
            sh = 2 * (gi & 15)
            big[gi >> 4] = (int(big[gi >> 4]) & ~(3 << sh) | (int(gb[p + j]) << sh)) & 0xFFFFFFFF
    npos = np.unique(rng.integers(0, total, size=3000)).astype(np.uint64)
    for gi in npos:                                   # invalid bases are stored as 0
        gi = int(gi); sh = 2 * (gi & 15)
        big[gi >> 4] = int(big[gi >> 4]) & ~(3 << sh) & 0xFFFFFFFF
    R = ol.OracleReads.from_arrays(words, off, npos)
    obits, ohits = ol.filter_reads(ol.OracleTable(bait_text, 31), R, 1, threads=os.cpu_count() or 1)
    ks = mf.KmerSet.from_text(bait_text, 31)
    reads = mf.Reads.from_packed(words, off, npos)
    assert reads.info.uniform_len == 0
    for mode in (mf.MODE_SCREENED, mf.MODE_EXHAUSTIVE):
        bits, hits, _ = mf.filter_reads(ks, reads, 1, mode, want_hits=True)
        assert np.array_equal(hits, ohits)
        assert np.array_equal(bits, obits)
    assert int(np.unpackbits(obits.view(np.uint8)).sum()) > 10000


def test_large_bait_uses_exact_smer_stage(mf, ol):
    """A 300 kbp bait overfills the LDS tables, which switches stage 3 (exact s-mer table in global
    memory) on; results must still be bit-exact."""
    import random
    rng = random.Random(77)
    bait = ">big\n" + "".join(rng.choices("ACGT", k=300_000)) + "\n>second\n" + "".join(rng.choices("ACGT", k=50_000)) + "\n"
    for k in (31, 25, 41):
        ks = mf.KmerSet.from_text(bait, k)
        info = ks.info
        assert info.n_keys > 300_000 and info.n_smers > 600_000
        t = ol.OracleTable(bait, k)
        assert np.array_equal(ks.export_table(), t.keys)
        seqs = make_reads(bait, 20000, seed=k, uniform=(k != 25))
        R = ol.OracleReads.from_seqs(seqs)
        reads = mf.Reads.from_packed(R.words, R.offsets, R.npos)
        obits, ohits = ol.filter_reads(t, R, 1, threads=os.cpu_count() or 1)
        for mode in (mf.MODE_SCREENED, mf.MODE_EXHAUSTIVE):
            bits, hits, _ = mf.filter_reads(ks, reads, 1, mode, want_hits=True)
            assert np.array_equal(hits, ohits), (k, mode)
            assert np.array_equal(bits, obits), (k, mode)


def test_bim_bait_sensitivity(mf, bait_text, tmp_path):
    """`bim` slot (bim/bim.py:43-58): parity with bwa is unpinned, so report what can be measured:
    pairs drawn from the bait (1 % substitutions) are kept, background pairs are not."""
    import random
    from mitoflex_amd.bim.bim import kmer_bait_map
    from tests.util_data import bait_records, revcomp
    rng = random.Random(4)
    g = bait_records(bait_text)[0]
    m1, m2, truth = [], [], []
    for i in range(4000):
        if i % 4 == 0:
            p = rng.randrange(0, len(g) - 400)
            frag = g[p:p + 400]
            a, b = frag[:150], revcomp(frag[-150:])
            mut = lambda s: "".join(c if rng.random() > 0.01 else rng.choice("ACGT") for c in s)
            m1.append(mut(a)); m2.append(mut(b)); truth.append(True)
        else:
            m1.append("".join(rng.choices("ACGT", k=150))); m2.append("".join(rng.choices("ACGT", k=150))); truth.append(False)
    fq1, fq2 = str(tmp_path / "r_1.fq"), str(tmp_path / "r_2.fq")
    write_fastq(fq1, m1, "p"); write_fastq(fq2, m2, "p")
    bait = str(tmp_path / "seed.fa"); open(bait, "w").write(bait_text)
    stats, o1, o2 = kmer_bait_map(8, bait, str(tmp_path), "gen0", fq1, fq2)
    assert stats.endswith("gen0.bait.stats") and o1.endswith("gen0.1.fq") and o2.endswith("gen0.2.fq")
    from mitoflex_amd.bim.bim import cal_insert
    assert 398 < cal_insert(stats, str(tmp_path), "gen0") < 402          # every fragment above is 400 bases long
    kept = {ln[1:].split()[0] for ln in open(o1).read().split("\n")[0::4] if ln}
    want = {f"p{i}" for i, t in enumerate(truth) if t}
    assert kept <= want                                  # no background pair survives
    assert len(kept) >= 0.99 * len(want)                 # sensitivity on bait-derived pairs
    assert open(o2).read().count("\n") == 4 * len(kept)


@pytest.mark.parametrize("n_dev", [2, 4])
def test_many_logical_devices_match_oracle(ol, bait_text, tmp_path, n_dev):
    """The multi-device file path (one worker, context, table copy and read set per device; batches dealt round robin and
    written back in order) on a single-GPU box: MF_FAKE_DEVICES maps N logical devices onto the physical one.  Runs in a
    child process (the variable is read when the library is loaded) through the `fastfilter bait` CLI."""
    import subprocess
    s1 = make_reads(bait_text, 4000, seed=5)
    s2 = make_reads(bait_text, 4000, seed=6)
    fq1, fq2 = str(tmp_path / "a_1.fq"), str(tmp_path / "a_2.fq")
    write_fastq(fq1, s1, "a")
    write_fastq(fq2, s2, "b")
    bait = str(tmp_path / "bait.fa")
    open(bait, "w").write(bait_text)
    o1, o2 = str(tmp_path / "o1.fq"), str(tmp_path / "o2.fq")
    ok, ot = ol.filter_fastq_files(bait, 31, 1, 0, fq1, fq2, o1, o2, threads=2)
    g1, g2 = str(tmp_path / "g1.fq"), str(tmp_path / "g2.fq")
    cli = os.path.join(ROOT, "mitoflex_amd", "assemble", "fastfilter")
    env = dict(os.environ, MITOFILTER_LIB=HOOKS_LIB, MF_FAKE_DEVICES=str(n_dev), MF_BATCH_READS="300", MF_INGEST="host")          # (the host pipeline over N devices; the device path over N devices: test_gpu_devingest.py)
    p = subprocess.run([cli, "bait", "--bait", bait, "-k", "31", "--fq1", fq1, "--fq2", fq2, "--out1", g1, "--out2", g2,
                        "--devices", str(n_dev)], capture_output=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[:2000]
    assert int(p.stdout.decode().split()[0]) == ok and ot == 4000
    assert open(g1, "rb").read() == open(o1, "rb").read()
    assert open(g2, "rb").read() == open(o2, "rb").read()


@pytest.mark.parametrize("world", [2, 8])
def test_bench_ranks_under_torchrun(tmp_path, world):
    """bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, one process per rank), with the ranks
    sharing the one visible GPU: the launcher's environment, the /dev/shm rendezvous, per-rank shards and rank 0's JSON line
    (eight ranks: the shape of the 8-GPU run, with a small shard instead of configs[3]'s 41.67 M reads per rank)."""
    import json
    import subprocess
    import sys
    port = 29000 + (os.getpid() + world) % 2000
    reads = 2000000 if world == 2 else 600000
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1",
                        "--prewarm-ms", "5", "--reads", str(reads), "--no-exhaustive", "--cpu-sample", "0", "--e2e-pairs", "0", "--rank-file-reads", "60000"],
                       capture_output=True, timeout=600, cwd=str(tmp_path), env=dict(os.environ, MF_BENCH_SHARE_GPU="1"))
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == 3 and d["scaling"] == "weak" and d["value"] > 0
    assert d["extra"]["reads_per_gpu"] == reads and 0.003 < d["extra"]["passed"] / reads < 0.008
    assert abs(d["value"] - world * reads * 3 / (d["ms_per_step"] * 3 / 1e3)) < 1e-6 * d["value"]        # whole-job rate over all ranks
    # configs[3]: per-GPU figures beside the aggregate (the aggregate is priced on the slowest rank)
    per, ms = d["extra"]["per_gpu_reads_per_s"], d["extra"]["per_rank_ms_per_step"]
    assert len(per) == len(ms) == world and all(x > 0 for x in per)
    assert abs(max(ms) - d["ms_per_step"]) < 1e-3 and sum(per) >= d["value"] * (1 - 1e-9)
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["whole_pass_frac"] > 0
    # every rank filtered a .gz shard of its own, file to file, through the device ingest path on "its" GPU
    rates, fl = d["extra"]["per_gpu_file_reads_per_s"], d["extra"]["file_level_one_process_per_gpu"]
    assert len(rates) == world and all(x > 0 for x in rates) and fl["device_ingest_on_every_rank"] and fl["reads_per_rank"] == 60000
    assert abs(fl["aggregate_reads_per_s"] - sum(rates)) < 1e-6 * sum(rates)


def test_bench_refuses_more_ranks_than_gpus(tmp_path):
    """outside tests, ranks are never wrapped onto one device silently"""
    import subprocess
    import sys
    port = 27000 + os.getpid() % 2000
    env = {k: v for k, v in os.environ.items() if k != "MF_BENCH_SHARE_GPU"}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--reads", "100000", "--no-exhaustive", "--cpu-sample", "0", "--e2e-pairs", "0"],
                       capture_output=True, timeout=300, cwd=str(tmp_path), env=env)
    assert p.returncode != 0 and b"one rank per GPU" in p.stderr


@pytest.mark.parametrize("k", [21, 31, 41])
def test_multi_pass_calls_match_oracle(mf, ol, bait_text, k):
    """A call with several passes pipelines them: the later kernels of pass i run beside the screen of pass i + 1 on a second
    stream, buffer sets rotate, and the exact kernel of all passes but the last takes its co-resident form.  What comes back
    (the tally of the last pass, and the result bitmap a following single pass leaves) must not depend on any of that."""
    n = 200_000
    ks = mf.KmerSet.from_text(bait_text, k)
    reads = mf.Reads.synth(n, 150, seed=11, bait_text=bait_text, keep_host=True)
    off = np.arange(n + 1, dtype=np.uint64) * 150
    R = ol.OracleReads.from_arrays(reads.host_words, off, reads.host_npos)
    t = ol.OracleTable(bait_text, k)
    for thr in (1, 2):
        obits, _ = ol.filter_reads(t, R, thr, threads=os.cpu_count() or 1)
        want = int(bits_to_bool(obits, n).sum())
        for steps in (2, 3, 5):
            st = mf.filter_resident(ks, reads, thr, mf.MODE_SCREENED, steps)
            assert st.n_pass == want, (k, thr, steps)
            bits, _, st1 = mf.filter_reads(ks, reads, thr, mf.MODE_SCREENED)
            assert np.array_equal(bits, obits) and st1.n_pass == want, (k, thr, steps)
        # every pass of a pipelined call, not only the last: each tallies into a block of its own
        per, st = mf.filter_resident_passes(ks, reads, thr, mf.MODE_SCREENED, 7)
        assert per.tolist() == [want] * 7 and st.n_pass == want, (k, thr, per.tolist())


@pytest.mark.parametrize("kind", ["split", "serial", "split-co", "split-one-stream", "one-screen-stream", "two-finish-streams"])
def test_other_pass_kinds_match_oracle(kind):
    """Environment switches select how a screened pass is run (read once per process, hence the child process).  MF_PASS=split:
    screen + mark + exact for every threshold; MF_PASS=serial: screen + finish without the cross-pass overlap -- and with the
    finish kernel for the stride-8 geometries (k < 28), which the default leaves to the candidate-bitmap pass.  `split-co`
    puts the co-resident form of the exact kernel (half the threads, folded bit table: what runs beside the next pass's screen
    in a multi-pass call) behind every screen; `split-one-stream` is the three-kernel pass without the second stream;
    `one-screen-stream` keeps all screens on one stream; `two-finish-streams` puts the finish kernels of consecutive passes on
    two streams for every input (the library does that by itself for bait-rich read sets only)."""
    import subprocess
    import sys
    extra = {"split": dict(MF_PASS="split"), "serial": dict(MF_PASS="serial"), "split-co": dict(MF_PASS="split", MF_EXACT_CO="1"),
             "split-one-stream": dict(MF_PASS="split", MF_SPLIT_PIPE="0"), "one-screen-stream": dict(MF_SCREEN_STREAMS="1"),
             "two-finish-streams": dict(MF_FINISH_STREAMS="2")}[kind]
    env = dict(os.environ, MF_ENV_KNOBS="1", **extra)          # (without MF_ENV_KNOBS=1 the library ignores these variables: test_options_are_not_read_from_a_stray_environment)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_kmer_golden.py"),
                        "-m", "gpu", "-q", "-x", "-k", "test_filter_matches_oracle or test_edge_cases or test_gpu_matches_golden or test_synth_uniform or test_multi_pass_calls"],
                       capture_output=True, env=env, cwd=ROOT, timeout=900)
    assert p.returncode == 0, p.stdout.decode()[-3000:]


def test_options_are_not_read_from_a_stray_environment(mf, ol, bait_text):
    """MF_PASS & co select kernels; a production process must not pick them up from a stray variable: without MF_ENV_KNOBS=1 they are
    ignored (the three-kernel pass leaves mark-kernel time in the stats, the default pass does not), mf_set_option is the way in, and
    either way the bits are the oracle's."""
    import subprocess
    import sys
    code = (
        "import os, sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from mitoflex_amd import mitofilter as mf\n"
        "from tests.util_data import make_bait\n"
        "bait = make_bait(); ks = mf.KmerSet.from_text(bait, 31, 0)\n"
        "r = mf.Reads.synth(300000, 150, seed=5, bait_text=bait, mito_ppm=5000, sub_ppm=10000, n_read_ppm=10000, n_base_ppm=1000, device=0)\n"
        "os.environ['MF_EVENT_STRIDE'] = '1'\n"
        "a = mf.filter_resident(ks, r, 1, mf.MODE_SCREENED, 2)\n"
        "mf.set_option('pass', 'split'); b = mf.filter_resident(ks, r, 1, mf.MODE_SCREENED, 2); mf.set_option('pass', 'default')\n"
        "c = mf.filter_resident(ks, r, 1, mf.MODE_SCREENED, 2)\n"
        "print(a.ms_mark > 0, b.ms_mark > 0, c.ms_mark > 0, a.n_pass == b.n_pass == c.n_pass)\n"
        "try:\n    mf.set_option('pass', 'nonsense'); print('accepted')\nexcept mf.MitoFilterError:\n    print('rejected')\n"
    ) % ROOT
    for knobs, want in ((None, "False True False True"), ("1", "True True False True")):
        env = dict(os.environ, MF_PASS="split")
        env.pop("MF_ENV_KNOBS", None)
        if knobs:
            env["MF_ENV_KNOBS"] = knobs
        p = subprocess.run([sys.executable, "-c", code], capture_output=True, env=env, timeout=300)
        assert p.returncode == 0, p.stderr.decode()[-2000:]
        out = p.stdout.decode().split("\n")
        assert out[0] == want and out[1] == "rejected", (knobs, out)
