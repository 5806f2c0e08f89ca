"""Seeded synthetic inputs shared by the tests (SURVEY.md 8d shapes, small)."""
from __future__ import annotations

import random
from typing import List

import numpy as np

COMP = str.maketrans("ACGTacgt", "TGCAtgca")


def revcomp(s: str) -> str:
    return s.translate(COMP)[::-1]


from mitoflex_amd.utility.synth_bait import bait_records, make_bait  # noqa: E402,F401  (the generator lives in the package: bench.py uses it too)


def make_reads(bait_text: str, n: int, seed: int, read_len: int = 150, uniform: bool = False,
               mito_frac: float = 0.3, n_frac: float = 0.1, sub_rate: float = 0.01) -> List[str]:
    """Mixed bag: bait-derived reads (both strands, substitutions), random reads, reads with N,
    short / empty reads, lower case.  uniform=True keeps every read at read_len."""
    rng = random.Random(seed)
    recs = [r for r in bait_records(bait_text) if len(r) > read_len + 10]
    out = []
    for i in range(n):
        L = read_len if uniform else rng.choice([read_len, read_len, rng.randint(0, 40), rng.randint(20, 400), 31, 30, 0])
        if rng.random() < mito_frac and L > 0:
            r = rng.choice(recs)
            p = rng.randrange(0, len(r) - L) if len(r) > L else 0
            s = r[p:p + L].upper()
            s = "".join(c if c in "ACGT" else "N" for c in s)
            if rng.random() < 0.5:
                s = revcomp(s)
            s = list(s)
            for j in range(len(s)):
                if rng.random() < sub_rate:
                    s[j] = rng.choice("ACGT")
            s = "".join(s)
        else:
            s = "".join(rng.choices("ACGT", k=L))
        if rng.random() < n_frac and L > 0:
            s = list(s)
            for _ in range(rng.randint(1, 3)):
                s[rng.randrange(L)] = rng.choice("NnRK.")
            s = "".join(s)
        if rng.random() < 0.05:
            s = s.lower()
        out.append(s)
    return out


def write_fastq(path: str, seqs: List[str], prefix: str = "r", crlf: bool = False, trailing_partial: bool = False,
                gz: bool = False):
    import gzip
    rng = random.Random(len(seqs))
    eol = "\r\n" if crlf else "\n"
    parts = []
    for i, s in enumerate(seqs):
        q = "".join(rng.choices("FGHIJ:;<=>?@", k=len(s)))
        parts.append(f"@{prefix}{i} desc{eol}{s}{eol}+{'x' if i % 7 == 0 else ''}{eol}{q}{eol}")
    if trailing_partial:
        parts.append(f"@partial{eol}ACGT{eol}")
    data = "".join(parts).encode()
    if gz:
        with gzip.open(path, "wb") as f:
            f.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)


def bits_to_bool(bits: np.ndarray, n: int) -> np.ndarray:
    return np.unpackbits(bits.view(np.uint8), bitorder="little")[:n].astype(bool)


def make_protein_bait(seed: int = 20261003, n_records: int = 12, code: int = 5):
    """A small synthetic protein database in the style of profile/MT_database (one record per gene,
    residues on one or several lines) plus the DNA of genes that translate to it under genetic code
    `code` -- the DNA is what reads are sampled from.  Returns (protein_fasta, gene_dna_fasta).
    Oddities on purpose: an 'X', a 'B', lower case, a trailing '*', a CRLF line, a record shorter
    than any k-mer, the same peptide in two records."""
    from oracle import prot_bait_ref as pr
    rng = random.Random(seed)
    # residue frequencies roughly like mitochondrial proteins (L, S, F, I heavy)
    weights = {"L": 16, "S": 10, "F": 9, "I": 8, "V": 7, "G": 7, "A": 6, "T": 6, "M": 5, "P": 4, "Y": 4, "N": 4,
               "W": 3, "K": 2, "E": 2, "D": 2, "H": 2, "Q": 2, "R": 2, "C": 1}
    aas, w = list(weights), list(weights.values())
    prots = ["".join(rng.choices(aas, weights=w, k=rng.randint(120, 520))) for _ in range(n_records)]
    prots[3] = prots[3][:200] + prots[1][50:110] + prots[3][200:]          # shared peptide stretch
    genes = [pr.back_translate(p, code, rng) for p in prots]
    shown = list(prots)
    shown[0] = shown[0][:40] + "X" + shown[0][41:90] + "B" + shown[0][91:]
    shown[2] = shown[2][:100] + shown[2][100:160].lower() + shown[2][160:] + "*"
    lines_p, lines_g = [], []
    for i, (p, g) in enumerate(zip(shown, genes)):
        lines_p.append(f">gi_SYN{i:03d}_GENE{i}_Synthetica_sp._{len(p)}_aa")
        if i % 3 == 0:
            lines_p.append(p)
        else:
            lines_p += [p[j:j + 60] for j in range(0, len(p), 60)]
        lines_g.append(f">gene{i}")
        lines_g.append(g)
    lines_p[2] = lines_p[2] + "\r"
    lines_p += [">tiny", "MLS"]
    return "\n".join(lines_p) + "\n", "\n".join(lines_g) + "\n"
