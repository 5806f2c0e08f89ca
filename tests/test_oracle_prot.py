"""Spec P (protein-space baiting): the C oracle against the string-level
definition in oracle/prot_bait_ref.py.  PARITY UNPINNED BY THE REFERENCE -- see that file."""
import random

import numpy as np
import pytest

from oracle import oracle_lib as ol
from oracle import prot_bait_ref as pr
from tests.util_data import bits_to_bool, make_protein_bait, make_reads, write_fastq


def test_genetic_code_tables_known_differences():
    std = pr.GENETIC_CODES[1]
    for code, tab in pr.GENETIC_CODES.items():
        assert len(tab) == 64 and set(tab) <= set(pr.AA + "*")
    t = pr.translate_codon
    assert [t(c, 1) for c in ("ATG", "TGA", "TAA", "TAG", "AGA", "ATA", "AAA", "CTG")] == list("M***RIKL")
    # the mitochondrial deviations MitoFlex's clades use (profile/codes.json: 2, 4, 5, 9) and their relatives
    assert (t("AGA", 2), t("AGG", 2), t("ATA", 2), t("TGA", 2)) == ("*", "*", "M", "W")
    assert (t("TGA", 4), t("ATA", 4), t("AGA", 4)) == ("W", "I", "R")
    assert (t("AGA", 5), t("AGG", 5), t("ATA", 5), t("TGA", 5), t("AAA", 5)) == ("S", "S", "M", "W", "K")
    assert (t("AAA", 9), t("AGA", 9), t("AGG", 9), t("TGA", 9), t("ATA", 9)) == ("N", "S", "S", "W", "I")
    assert (t("AGA", 13), t("AGG", 13), t("ATA", 13), t("TGA", 13)) == ("G", "G", "M", "W")
    assert (t("TAA", 14), t("AAA", 14), t("TGA", 14), t("TAG", 14)) == ("Y", "N", "W", "*")
    assert (t("CTT", 3), t("CTG", 3), t("ATA", 3), t("TGA", 3)) == ("T", "T", "M", "W")
    assert (t("AAA", 21), t("ATA", 21), t("AGA", 21), t("TGA", 21)) == ("N", "M", "S", "W")
    assert pr.GENETIC_CODES[11] == std
    # how many codons differ from the standard code
    diff = {c: sum(a != b for a, b in zip(tab, std)) for c, tab in pr.GENETIC_CODES.items()}
    assert diff == {1: 0, 2: 4, 3: 6, 4: 1, 5: 4, 9: 4, 11: 0, 13: 4, 14: 5, 21: 5}
    assert t("ANG", 1) == "X" and t("acg", 1) == "T"


def test_six_frames_by_hand():
    #        M  L  S  *      (code 1: ATG CTT TCA TAA)
    seq = "ATGCTTTCATAA"
    f = pr.six_frames(seq, 1)
    assert f[0] == "MLS*"
    assert f[1] == pr.translate("TGCTTTCATAA", 1) == "CFH"
    assert f[3] == pr.translate("TTATGAAAGCAT", 1) == "L*KH"
    assert len(f) == 6 and [len(x) for x in f] == [4, 3, 3, 4, 3, 3]
    assert pr.six_frames("ATGNTTTCA", 1)[0] == "MXS"


def test_pep_code_and_hits_by_hand():
    assert pr.pep_code("ACD") == 0 | (1 << 5) | (2 << 10)
    assert pr.pep_code("AXD") is None and pr.pep_code("A*D") is None and pr.pep_code("acd") == pr.pep_code("ACD")
    rng = random.Random(3)
    prot = "MLSFIVGATMPYNWKEDHQRC" * 3
    bait = pr.bait_set(">p\n" + prot + "\n", 7)
    gene = pr.back_translate(prot, 5, rng)
    assert pr.translate(gene, 5) == prot
    n_win = len(prot) - 7 + 1
    assert pr.read_hits(gene, 7, 5, bait) >= n_win                          # frame +0 carries every window
    assert pr.read_hits("GG" + gene, 7, 5, bait) >= n_win                   # frame +2
    assert pr.read_hits(pr.revcomp_any(gene) + "A", 7, 5, bait) >= n_win    # a reverse frame
    broken = gene[:30] + "N" + gene[31:]
    assert pr.read_hits(broken, 7, 5, bait) < pr.read_hits(gene, 7, 5, bait)


@pytest.mark.parametrize("kp,code", [(4, 1), (7, 5), (9, 2), (12, 9), (5, 14)])
def test_c_oracle_matches_string_spec(kp, code):
    prot_fa, gene_fa = make_protein_bait(code=code)
    seqs = make_reads(gene_fa, 400, seed=kp * 100 + code, mito_frac=0.4)
    bait = pr.bait_set(prot_fa, kp)
    want = [pr.read_hits(s, kp, code, bait) for s in seqs]
    T = ol.OracleTable(prot_fa, kp, protein=True)
    R = ol.OracleReads.from_seqs(seqs)
    for thr in (1, 3):
        bits, hits = ol.pfilter_reads(T, R, code, thr, threads=3)
        assert hits.tolist() == want
        assert bits_to_bool(bits, len(seqs)).tolist() == [h >= thr for h in want]
    assert sum(h > 0 for h in want) > 40            # the planted reads are found
    # table: same layout as the ascending-insertion definition
    lay = pr.table_layout(prot_fa, kp)
    assert T.slots == len(lay) == pr.table_slots(prot_fa, kp)
    assert T.n_keys == len(bait)
    keys = T.keys
    assert [int(k) if k != np.uint64(0xFFFFFFFFFFFFFFFF) else -1 for k in keys] == lay


def test_c_oracle_edge_cases():
    prot_fa, gene_fa = make_protein_bait()
    T = ol.OracleTable(prot_fa, 7, protein=True)
    bait = pr.bait_set(prot_fa, 7)
    seqs = ["", "A", "AC", "ACG", "ACGTACGTACGTACGTACGT", "N" * 30, "ACGTN" * 12, "acgtacgtacgtacgtacgtacgtacgt"]
    R = ol.OracleReads.from_seqs(seqs)
    _, hits = ol.pfilter_reads(T, R, 5)
    assert hits.tolist() == [pr.read_hits(s, 7, 5, bait) for s in seqs]
    with pytest.raises(RuntimeError):
        ol.OracleTable(prot_fa, 3, protein=True)
    with pytest.raises(RuntimeError):
        ol.OracleTable(prot_fa, 13, protein=True)
    with pytest.raises(RuntimeError):
        ol.pfilter_reads(T, R, 7)                    # genetic code 7 does not exist
    empty = ol.OracleTable(">tiny\nMLS\n", 7, protein=True)
    assert empty.n_keys == 0 and empty.slots == 1024
    assert ol.pfilter_reads(empty, R, 5)[1].sum() == 0


def test_c_oracle_fastq_files(tmp_path):
    prot_fa, gene_fa = make_protein_bait()
    s1 = make_reads(gene_fa, 300, seed=5)
    s2 = make_reads(gene_fa, 300, seed=6)
    fq1, fq2 = str(tmp_path / "a_1.fq"), str(tmp_path / "a_2.fq")
    write_fastq(fq1, s1, "a"); write_fastq(fq2, s2, "b", crlf=True)
    bait_path = str(tmp_path / "db.fa")
    open(bait_path, "w").write(prot_fa)
    bait = pr.bait_set(prot_fa, 8)
    p1 = pr.filter_reads(s1, 8, 5, bait); p2 = pr.filter_reads(s2, 8, 5, bait)
    for mode, rule in ((0, lambda a, b: a or b), (1, lambda a, b: a and b)):
        kept, total = ol.pfilter_fastq_files(bait_path, 8, 5, 1, mode, fq1, fq2, str(tmp_path / "o1.fq"), str(tmp_path / "o2.fq"))
        keep = [rule(a, b) for a, b in zip(p1, p2)]
        assert (kept, total) == (sum(keep), 300)
        got = [ln for ln in open(tmp_path / "o1.fq").read().split("\n")[1::4]]
        assert got == [s for s, k in zip(s1, keep) if k]
