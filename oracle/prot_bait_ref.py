"""TEST INFRASTRUCTURE ONLY -- string-level specification of protein-space baiting ("Spec P").

PARITY UNPINNED BY THE REFERENCE.  SURVEY.md 8f "next" #4: `profile/MT_database/*.fa` is a
PROTEIN database (findmitoscaf/findmitoscaf.py:57 hands it to tblastn through
annotation/annotation_tookit.py:55-97 with `-db_gencode <code>`), so the only way to bait reads
with it is in amino-acid space: translate every read in six frames and look its peptide k-mers
up in the set of peptide k-mers of the database.  The reference does this only on assembled
contigs and only through tblastn; nothing under /root/reference computes what is written here,
so this file is the definition.  It is written with Python strings so it can be audited by eye.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Spec P
  P1  amino-acid alphabet: the 20 standard one-letter codes, case-insensitive, coded by their
      rank in "ACDEFGHIKLMNPQRSTVWY" (A=0 .. Y=19).  Every other character of a protein record
      (X, B, Z, J, U, O, '*', '-') is invalid and breaks windows.  Records follow the same FASTA
      rules as the nucleotide bait (Spec B5): '>' at line start opens a record, whitespace inside
      sequence lines is ignored, windows never span records.
  P2  peptide k-mer code of a window w of kp residues (4 <= kp <= 12):
      sum_i code(w[i]) << 5i  (first residue least significant) -- an unsigned 5*kp-bit integer.
  P3  translation uses one NCBI genetic code (transl_table id; the reference picks it per clade
      from profile/codes.json: 2, 4, 5, 9).  A codon holding an invalid base is invalid; a stop
      codon is invalid; both break windows.  Start codons are translated as ordinary codons.
  P4  six frames of a read of length L: forward offsets 0,1,2 and offsets 0,1,2 of the reverse
      complement; a frame is the codons at offset, offset+3, ... that fit entirely.
  P5  hits(read) = number of (frame, window) pairs whose kp codons are all valid and whose
      peptide k-mer code is in the bait set (all positions, not distinct k-mers);
      pass = hits >= T, T >= 1.  Pair rule, FASTQ conventions and the table layout (slots =
      pow2 >= max(1024, 2 * n_windows), distinct keys ascending, linear probing from
      fold32(key) & (slots-1), empty = all ones) are the nucleotide filter's.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Set

from oracle import kmer_bait_ref as kb

AA = "ACDEFGHIKLMNPQRSTVWY"
_AA_CODE = {c: i for i, c in enumerate(AA)}

# NCBI transl_table strings, codons in the order TTT TTC TTA TTG TCT ... GGG (bases T, C, A, G)
GENETIC_CODES: Dict[int, str] = {
    1:  "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG",   # standard
    2:  "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSS**VVVVAAAADDEEGGGG",   # vertebrate mitochondrial
    3:  "FFLLSSSSYY**CCWWTTTTPPPPHHQQRRRRIIMMTTTTNNKKSSRRVVVVAAAADDEEGGGG",   # yeast mitochondrial
    4:  "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG",   # mold / protozoan / coelenterate mito
    5:  "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSSSSVVVVAAAADDEEGGGG",   # invertebrate mitochondrial
    9:  "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNNKSSSSVVVVAAAADDEEGGGG",   # echinoderm / flatworm mito
    11: "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG",   # bacterial (= 1 for residues)
    13: "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNKKSSGGVVVVAAAADDEEGGGG",   # ascidian mitochondrial
    14: "FFLLSSSSYYY*CCWWLLLLPPPPHHQQRRRRIIIMTTTTNNNKSSSSVVVVAAAADDEEGGGG",   # alternative flatworm mito
    21: "FFLLSSSSYY**CCWWLLLLPPPPHHQQRRRRIIMMTTTTNNNKSSSSVVVVAAAADDEEGGGG",   # trematode mitochondrial
}
_TCAG = "TCAG"


def translate_codon(codon: str, code: int) -> str:
    """One residue, '*' for a stop, 'X' when the codon holds anything but ACGT."""
    c = codon.upper()
    if len(c) != 3 or any(b not in "ACGT" for b in c):
        return "X"
    return GENETIC_CODES[code][16 * _TCAG.index(c[0]) + 4 * _TCAG.index(c[1]) + _TCAG.index(c[2])]


def translate(seq: str, code: int) -> str:
    return "".join(translate_codon(seq[i:i + 3], code) for i in range(0, len(seq) - 2, 3))


def revcomp_any(seq: str) -> str:
    """Reverse complement that keeps invalid letters invalid (they become 'N')."""
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    return "".join(comp.get(b, "N") for b in reversed(seq.upper()))


def six_frames(seq: str, code: int) -> List[str]:
    rc = revcomp_any(seq)
    return [translate(seq[o:], code) for o in range(3)] + [translate(rc[o:], code) for o in range(3)]


def pep_code(window: str) -> Optional[int]:
    v = 0
    for i, ch in enumerate(window.upper()):
        a = _AA_CODE.get(ch)
        if a is None:
            return None
        v |= a << (5 * i)
    return v


def protein_records(fasta_text: str) -> List[str]:
    return kb.read_fasta_records(fasta_text)


def bait_set(fasta_text: str, kp: int) -> Set[int]:
    if not 4 <= kp <= 12:
        raise ValueError("kp out of range [4,12]")
    out: Set[int] = set()
    for rec in protein_records(fasta_text):
        for i in range(len(rec) - kp + 1):
            v = pep_code(rec[i:i + kp])
            if v is not None:
                out.add(v)
    return out


def read_hits(seq: str, kp: int, code: int, bait: Set[int]) -> int:
    hits = 0
    for pep in six_frames(seq, code):
        for i in range(len(pep) - kp + 1):
            v = pep_code(pep[i:i + kp])
            if v is not None and v in bait:
                hits += 1
    return hits


def filter_reads(seqs: Sequence[str], kp: int, code: int, bait: Set[int], threshold: int = 1) -> List[bool]:
    if threshold < 1:
        raise ValueError("threshold must be >= 1")
    return [read_hits(s, kp, code, bait) >= threshold for s in seqs]


def n_windows(fasta_text: str, kp: int) -> int:
    return sum(max(0, len(r) - kp + 1) for r in protein_records(fasta_text))


def table_slots(fasta_text: str, kp: int) -> int:
    s, n = 1024, n_windows(fasta_text, kp)
    while s < 2 * n:
        s <<= 1
    return s


def table_layout(fasta_text: str, kp: int) -> List[int]:
    """Distinct keys ascending, plain linear probing from fold32(key); empty = -1."""
    slots = table_slots(fasta_text, kp)
    tab = [-1] * slots
    for key in sorted(bait_set(fasta_text, kp)):
        s = kb.fold32(key) & (slots - 1)
        while tab[s] != -1:
            s = (s + 1) & (slots - 1)
        tab[s] = key
    return tab


def back_translate(protein: str, code: int, rng) -> str:
    """A DNA sequence that translates to `protein` (test-data helper): a random synonymous codon per residue."""
    table = GENETIC_CODES[code]
    by_aa: Dict[str, List[str]] = {}
    for i, aa in enumerate(table):
        by_aa.setdefault(aa, []).append(_TCAG[i // 16] + _TCAG[(i // 4) % 4] + _TCAG[i % 4])
    return "".join(rng.choice(by_aa[a]) for a in protein.upper() if a in by_aa)
