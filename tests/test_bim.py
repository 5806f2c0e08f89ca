"""The `bim` row: `kmer_bait_map` / `cal_insert` stand where the reference's `bwa_map` / `cal_insert`
(bim/bim.py:43-78) stand in the loop of MitoFlex.py:346-375.  Parity with bwa / samtools is UNPINNED (un-vendored
tools); what is tested is what can be: the insert-size estimate against simulated truth, the reference's parsing of
the stats text, and -- on the GPU -- a multi-generation loop with a stand-in assembler whose kept sets equal the oracle's."""
import os
import random

import pytest

from tests.util_data import revcomp, write_fastq


def _genome(n=16000, seed=77):
    rng = random.Random(seed)
    return "".join(rng.choices("ACGT", k=n))


def _pairs(genome, n, seed, L=150, mean=350, sd=30, sub=0.01):
    """n inward pairs (mate 1 forward / mate 2 reversed or the other way round) with known fragment starts and sizes."""
    rng = random.Random(seed)
    m1, m2, frags = [], [], []
    for _ in range(n):
        size = max(L + 10, int(rng.gauss(mean, sd)))
        a = rng.randrange(0, len(genome) - size)
        frag = genome[a:a + size]
        x, y = frag[:L], revcomp(frag[-L:])
        if rng.random() < 0.5:
            x, y = y, x
        mut = lambda s: "".join(c if rng.random() >= sub else rng.choice("ACGT") for c in s)
        m1.append(mut(x)); m2.append(mut(y)); frags.append((a, size))
    return m1, m2, frags


def test_insert_size_estimate_and_cal_insert(tmp_path):
    from mitoflex_amd.bim import bim
    g = _genome()
    fa = tmp_path / "bait.fa"
    fa.write_text(">g\n" + "\n".join(g[i:i + 60] for i in range(0, len(g), 60)) + "\n>other\n" + _genome(900, 5) + "\n")
    m1, m2, frags = _pairs(g, 3000, 1)
    write_fastq(str(tmp_path / "k.1.fq"), m1, "a")
    write_fastq(str(tmp_path / "k.2.fq"), m2, "b")
    hist = bim.estimate_insert_sizes(str(fa), str(tmp_path / "k.1.fq"), str(tmp_path / "k.2.fq"), 31)
    assert sum(hist.values()) > 0.97 * len(m1)                      # nearly every pair has an anchor in both mates
    truth = sum(s for _, s in frags) / len(frags)
    est = sum(a * b for a, b in hist.items()) / sum(hist.values())
    assert abs(est - truth) < 1.5
    # the stats text is parsed the way bim/bim.py:65-78 parses `samtools stats | grep ^IS | cut -f 2-`
    st = tmp_path / "w.bait.stats"
    st.write_text("# comment\nSN\tfoo:\t1\n" + "".join(f"IS\t{a}\t{b}\t{b}\t0\t0\n" for a, b in sorted(hist.items())))
    assert bim.cal_insert(str(st), str(tmp_path), "w") == pytest.approx(est)
    assert (tmp_path / "w.stats").read_text() == st.read_text()    # teed like the reference's stat_file
    st.write_text("# nothing\n")
    with pytest.raises(ZeroDivisionError):                          # the reference divides by the sum of an empty column too
        bim.cal_insert(str(st), str(tmp_path), "w")


@pytest.mark.gpu
def test_bim_generations_with_stand_in_assembler(built_lib, tmp_path):
    """MitoFlex.py:346-375 with `kmer_bait_map` for `bwa_map` and a stand-in for `assemble()`: generation 0 baits with a 2 kbp
    seed; each generation's "assembly" is the stretch of the genome its kept reads cover (the stand-in knows where the
    reads came from), and that is the next bait -- rebuilt as a k-mer set on the device every generation.  The kept set of
    every generation equals the CPU oracle's for the same bait, grows, and the insert size stays at the simulated value."""
    from mitoflex_amd import mitofilter as mf
    from mitoflex_amd.bim import bim
    from oracle import oracle_lib as ol
    if mf.device_count() < 1:
        pytest.fail("no GPU visible")
    g = _genome(12000, 3)
    m1, m2, frags = _pairs(g, 4000, 9, sub=0.005)
    rng = random.Random(4)
    bg1 = ["".join(rng.choices("ACGT", k=150)) for _ in range(4000)]
    bg2 = ["".join(rng.choices("ACGT", k=150)) for _ in range(4000)]
    order = list(range(8000)); rng.shuffle(order)
    all1 = [(m1 + bg1)[i] for i in order]; all2 = [(m2 + bg2)[i] for i in order]
    where = [frags[i] if i < 4000 else None for i in order]
    fq1, fq2 = str(tmp_path / "r_1.fq"), str(tmp_path / "r_2.fq")
    write_fastq(fq1, all1, "r"); write_fastq(fq2, all2, "r")
    bait = str(tmp_path / "work.bait.fa")
    open(bait, "w").write(">seed\n" + g[5000:7000] + "\n")
    asm = tmp_path / "asm"; asm.mkdir()
    kept_prev, spans = set(), []
    for gen in range(3):
        stats, k1, k2 = bim.kmer_bait_map(8, bait, str(asm), "work", fq1, fq2)
        ins = bim.cal_insert(stats, str(asm), "work")
        assert 340 < ins < 360
        o1, o2 = str(tmp_path / "o1.fq"), str(tmp_path / "o2.fq")
        ok, ot = ol.filter_fastq_files(bait, 31, 1, 0, fq1, fq2, o1, o2, threads=2)
        assert open(k1, "rb").read() == open(o1, "rb").read() and open(k2, "rb").read() == open(o2, "rb").read()
        kept = {int(l[2:].split("/")[0].split()[0]) for i, l in enumerate(open(k1)) if i % 4 == 0}
        assert kept >= kept_prev and all(where[i] is not None for i in kept)       # grows; no background pair ever
        kept_prev = kept
        # stand-in assembler: the genome interval covered by the kept fragments
        lo = min(where[i][0] for i in kept); hi = max(where[i][0] + where[i][1] for i in kept)
        spans.append(hi - lo)
        open(bait, "w").write(f">gen{gen + 1}\n" + g[lo:hi] + "\n")
    assert spans[0] > 2000 and spans[1] > spans[0] and spans[2] >= spans[1]
