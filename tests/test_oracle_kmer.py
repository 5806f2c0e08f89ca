"""Oracle B (oracle/kmer_bait_oracle.c) pinned against the string-level specification
(oracle/kmer_bait_ref.py).  PARITY UNPINNED BY THE REFERENCE: MitoFlex has no k-mer read filter,
so these two independent restatements of rows B1-B5 are what the GPU path is held to."""
import gzip
import random

import numpy as np
import pytest

from oracle import kmer_bait_ref as ref
from oracle import oracle_lib as ol
from tests.util_data import bait_records, make_reads, write_fastq

KS = [11, 15, 21, 31, 32, 33, 41, 63]


def _table_as_ints(t):
    keys, out = t.keys, []
    for s in range(t.slots):
        if t.kw == 1:
            v = int(keys[s]); out.append(-1 if v == 2**64 - 1 else v)
        else:
            lo, hi = int(keys[2 * s]), int(keys[2 * s + 1])
            out.append(-1 if (lo == 2**64 - 1 and hi == 2**64 - 1) else lo | (hi << 64))
    return out


@pytest.fixture(scope="module")
def small_bait():
    rng = random.Random(9)
    return (">a first\n" + "".join(rng.choices("ACGT", k=700)) + "\nNNNN" + "".join(rng.choices("acgt", k=300)) + "\r\n"
            ">b\n" + "".join(rng.choices("ACGTRY", k=400)) + "\n>short\nACGTACGT\n>c\n" + "T" * 80 + "A" * 70 + "\n")


@pytest.mark.parametrize("k", KS)
def test_table_matches_spec(small_bait, k):
    t = ol.OracleTable(small_bait, k)
    bs = ref.bait_set(small_bait, k)
    assert t.n_keys == len(bs) and t.slots == ref.table_slots(small_bait, k)
    assert _table_as_ints(t) == ref.table_layout(small_bait, k)
    for code in list(bs)[:50]:
        assert t.contains(code)
    assert not t.contains((1 << (2 * k)) - 2) or ((1 << (2 * k)) - 2) in bs


@pytest.mark.parametrize("k", [21, 31, 41])
@pytest.mark.parametrize("load", ["normal", "crowded"])
def test_history_independent_layout(small_bait, k, load):
    """The min-swap insertion the device builder uses reproduces the ascending-order layout for any
    insertion order, also when the table is crowded and probes wrap around."""
    keys = sorted(ref.bait_set(small_bait, k))
    slots = ref.table_slots(small_bait, k)
    if load == "crowded":
        keys = keys[:200]
        slots = 256
    expect = [-1] * slots
    for key in keys:
        s = ref.hash_key(key, k) & (slots - 1)
        while expect[s] != -1:
            s = (s + 1) & (slots - 1)
        expect[s] = key
    rng = random.Random(k)
    for _ in range(5):
        order = keys + keys[:20]
        rng.shuffle(order)
        assert ref.ordered_insert_any_order(order, k, slots) == expect


@pytest.mark.parametrize("k", KS)
def test_filter_matches_spec(small_bait, k):
    rng = random.Random(k)
    recs = [r for r in ref.read_fasta_records(small_bait) if len(r) > 200]
    seqs = []
    for i in range(150):
        r = rng.choice(recs); p = rng.randrange(0, len(r) - 160)
        s = r[p:p + rng.randint(0, 160)]
        if rng.random() < .5:
            s = ref.revcomp(ref._norm(s).replace("N", "A"))
        if rng.random() < .3:
            s = "".join(rng.choices("ACGTN", k=rng.randint(0, 170)))
        s = list(s)
        for j in range(len(s)):
            if rng.random() < .01:
                s[j] = rng.choice("ACGTNacgt")
        seqs.append("".join(s))
    bs = ref.bait_set(small_bait, k)
    t = ol.OracleTable(small_bait, k)
    R = ol.OracleReads.from_seqs(seqs)
    w, o, npos = ref.pack_reads(seqs)
    assert list(R.offsets) == o and list(R.npos) == npos and list(R.words[:len(w)]) == w
    exp = [ref.read_hits(s, k, bs) for s in seqs]
    for thr in (1, 3):
        bits, hits = ol.filter_reads(t, R, thr, threads=3)
        assert list(hits) == exp
        assert [(int(bits[i >> 5]) >> (i & 31)) & 1 for i in range(len(seqs))] == [int(h >= thr) for h in exp]
    # sub-range entry point used by the sharding tests
    bits, hits = ol.filter_reads(t, R, 1, first=40, count=70)
    assert list(hits) == exp[40:110]


@pytest.mark.parametrize("gz", [False, True])
def test_fastq_conventions(bait_text, tmp_path, gz):
    """4-line records, CR stripped, partial tail dropped, .gz by extension, PE zipped to the shorter
    file, survivors written header/seq/+/qual (reference conventions: filter_bin main.rs:214,261-321)."""
    ext = ".fq.gz" if gz else ".fq"
    s1, s2 = make_reads(bait_text, 400, seed=3), make_reads(bait_text, 420, seed=4)
    fq1, fq2 = str(tmp_path / ("r1" + ext)), str(tmp_path / ("r2" + ext))
    write_fastq(fq1, s1, "a", crlf=True, trailing_partial=True, gz=gz)
    write_fastq(fq2, s2, "b", gz=gz)
    assert [s for _, s, _ in ref.read_fastq(fq1)] == s1
    R = ol.OracleReads.from_fastq(fq1)
    assert R.n_reads == 400
    bait = str(tmp_path / "bait.fa"); open(bait, "w").write(bait_text)
    bs = ref.bait_set(bait_text, 31)
    p1, p2 = ref.filter_reads(s1, 31, bs, 1), ref.filter_reads(s2[:400], 31, bs, 1)
    for mode, name in ((0, "either"), (1, "both")):
        o1, o2 = str(tmp_path / "o1.fq"), str(tmp_path / "o2.fq")
        kept, total = ol.filter_fastq_files(bait, 31, 1, mode, fq1, fq2, o1, o2, threads=2)
        keep = ref.pair_keep(p1, p2, name)
        assert (kept, total) == (sum(keep), 400)
        r1, r2 = ref.read_fastq(fq1), ref.read_fastq(fq2)
        assert open(o1).read() == ref.format_fastq(r for r, k_ in zip(r1, keep) if k_)
        assert open(o2).read() == ref.format_fastq(r for r, k_ in zip(r2, keep) if k_)
