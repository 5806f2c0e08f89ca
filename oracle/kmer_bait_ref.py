"""TEST INFRASTRUCTURE ONLY -- string-level specification of the k-mer bait filter.

PARITY UNPINNED BY THE REFERENCE: MitoFlex contains no k-mer read filter
(the reference `assemble/fastfilter` is a contig FASTA
length/depth filter, `assemble/fastfilter_src/src/main.rs:9-134`).  This file
is therefore the *definition* of the Group-B semantics (SURVEY.md 8a rows
B1-B5) written in the most obvious way possible -- Python strings, a set of
strings, `str.translate` for the reverse complement -- so that the C oracle
(`oracle/kmer_bait_oracle.c`) and the HIP path can both be checked against
something that is easy to audit by eye.  Only small inputs: it is O(L*k) per read.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product (`mitoflex_amd/`) never does.

Conventions borrowed from the reference where it has any:
  * FASTQ is read as strict 4-line records, line 3 ignored, trailing partial
    record dropped, CR stripped  (filter/filter_bin/src/main.rs:287-321:
    `lines().tuples()` over 4 lines; `lines()` strips "\n" and "\r\n").
  * gzip is selected by the ".gz" extension (filter/filter_bin/src/helper.rs:22).
  * survivors are written header / seq / "+" / qual (main.rs:261-268).

Semantics fixed here (no reference counterpart):
  B1  alphabet  A/a=0 C/c=1 G/g=2 T/t=3, every other byte is "invalid".
  B2  a window of k consecutive bases is valid iff it holds no invalid base;
      canonical(window) = min(code(window), code(revcomp(window))) where
      code(w) = sum_j base(w[j]) << 2j   (first base least significant),
      compared as an unsigned 2k-bit integer.
  B3  membership is exact on the full canonical k-mer.
  B4  hits(read) = number of valid windows (all positions, not distinct
      k-mers) whose canonical k-mer is in the bait set; pass = hits >= T, T>=1.
      Paired-end: a pair is kept if either mate passes (mode "either") or if
      both pass (mode "both").
  B5  bait set = canonical k-mers of every valid window of every record of a
      nucleotide FASTA (multi-line records; whitespace inside sequence lines
      ignored; IUPAC / other letters are invalid and break windows; windows
      never span records).
"""
from __future__ import annotations

import gzip
from typing import Iterable, List, Sequence, Set, Tuple

_CODE = {"A": 0, "C": 1, "G": 2, "T": 3}
_COMP = str.maketrans("ACGT", "TGCA")


def _norm(seq: str) -> str:
    """Upper-case ACGT kept, every other character becomes 'N'."""
    return "".join(c if c in "ACGT" else "N" for c in seq.upper())


def kmer_code(window: str) -> int:
    """code(w) = sum_j base(w[j]) << 2j  -- first base least significant."""
    v = 0
    for j, c in enumerate(window):
        v |= _CODE[c] << (2 * j)
    return v


def revcomp(window: str) -> str:
    return window.translate(_COMP)[::-1]


def canonical_code(window: str) -> int:
    return min(kmer_code(window), kmer_code(revcomp(window)))


def read_fasta_records(text: str) -> List[str]:
    """B5: '>' starts a record; other lines are sequence (whitespace dropped)."""
    recs: List[List[str]] = []
    cur: List[str] | None = None
    for line in text.split("\n"):
        if line.startswith(">"):
            cur = []
            recs.append(cur)
            continue
        body = "".join(line.split())
        if not body:
            continue
        if cur is None:  # sequence before any header: anonymous record
            cur = []
            recs.append(cur)
        cur.append(body)
    return ["".join(r) for r in recs]


def bait_set(fasta_text: str, k: int) -> Set[int]:
    out: Set[int] = set()
    for rec in read_fasta_records(fasta_text):
        s = _norm(rec)
        for p in range(0, len(s) - k + 1):
            w = s[p:p + k]
            if "N" in w:
                continue
            out.add(canonical_code(w))
    return out


def read_hits(seq: str, k: int, bait: Set[int]) -> int:
    s = _norm(seq)
    hits = 0
    for p in range(0, len(s) - k + 1):
        w = s[p:p + k]
        if "N" in w:
            continue
        if canonical_code(w) in bait:
            hits += 1
    return hits


def filter_reads(seqs: Sequence[str], k: int, bait: Set[int], threshold: int = 1) -> List[bool]:
    assert threshold >= 1
    return [read_hits(s, k, bait) >= threshold for s in seqs]


def pair_keep(pass1: Sequence[bool], pass2: Sequence[bool], mode: str = "either") -> List[bool]:
    if mode == "either":
        return [a or b for a, b in zip(pass1, pass2)]
    if mode == "both":
        return [a and b for a, b in zip(pass1, pass2)]
    raise ValueError(mode)


def _open_text(path: str):
    if path.endswith(".gz"):
        return gzip.open(path, "rt", newline="")
    return open(path, "rt", newline="")


def read_fastq(path: str) -> List[Tuple[str, str, str]]:
    """Strict 4-line records -> (header, seq, qual); partial tail dropped."""
    with _open_text(path) as fh:
        data = fh.read()
    lines = data.split("\n")
    if lines and lines[-1] == "":
        lines.pop()
    lines = [ln[:-1] if ln.endswith("\r") else ln for ln in lines]
    out = []
    for i in range(0, len(lines) - 3, 4):
        out.append((lines[i], lines[i + 1], lines[i + 3]))
    return out


def format_fastq(records: Iterable[Tuple[str, str, str]]) -> str:
    return "".join(f"{h}\n{s}\n+\n{q}\n" for h, s, q in records)


# ---- packed layout (B1), stated once more at string level ------------------

def pack_reads(seqs: Sequence[str]):
    """Dense 2-bit little-endian stream, no per-read padding.

    Returns (words: list[int] u32, offsets: list[int] base offsets (n+1),
    npos: sorted list of global base indices that are invalid).
    Base i of the stream sits in words[i >> 4] at bits [2*(i&15), 2*(i&15)+1];
    invalid bases are stored as 0 and listed in npos.
    """
    offsets = [0]
    npos: List[int] = []
    total = 0
    big = 0
    for s in seqs:
        for j, c in enumerate(_norm(s)):
            if c == "N":
                npos.append(total + j)
            else:
                big |= _CODE[c] << (2 * (total + j))
        total += len(s)
        offsets.append(total)
    n_words = (total + 15) // 16
    words = [(big >> (32 * i)) & 0xFFFFFFFF for i in range(n_words)]
    return words, offsets, npos


# ---- table layout (B3), string-level restatement ---------------------------

_M64 = (1 << 64) - 1


_M32 = (1 << 32) - 1


def fold32(x: int) -> int:
    """Two 32-bit multiplies per 64-bit word, folded, xor-shift finish."""
    a = ((x & _M32) * 0x9E3779B1) & _M32
    b = ((x >> 32) * 0x85EBCA77) & _M32
    h = a ^ (((b << 15) | (b >> 17)) & _M32)
    return h ^ (h >> 15)


def hash_key(code: int, k: int) -> int:
    lo, hi = code & _M64, code >> 64
    if k <= 32:
        return fold32(lo)
    h = fold32(lo) ^ ((fold32(hi) * 0xC2B2AE3D) & _M32)
    return h ^ (h >> 16)


def n_windows(fasta_text: str, k: int) -> int:
    return sum(max(0, len(r) - k + 1) for r in read_fasta_records(fasta_text))


def table_slots(fasta_text: str, k: int) -> int:
    slots = 1024
    while slots < 2 * n_windows(fasta_text, k):
        slots <<= 1
    return slots


def table_layout(fasta_text: str, k: int) -> List[int]:
    """Distinct canonical keys inserted ascending, plain linear probing.
    Returns a list of `slots` ints; empty slot = -1."""
    slots = table_slots(fasta_text, k)
    tab = [-1] * slots
    for key in sorted(bait_set(fasta_text, k)):
        s = hash_key(key, k) & (slots - 1)
        while tab[s] != -1:
            s = (s + 1) & (slots - 1)
        tab[s] = key
    return tab


def ordered_insert_any_order(keys: Sequence[int], k: int, slots: int) -> List[int]:
    """The history-independent insertion the device builder uses (min-swap
    linear probing, Shun & Blelloch, SPAA'14 'Phase-concurrent hash tables
    for determinism'): at each slot keep the smaller key and carry the larger
    one onward.  For any insertion order it must reproduce table_layout()."""
    EMPTY = 1 << 130
    tab = [EMPTY] * slots
    for v in keys:
        s = hash_key(v, k) & (slots - 1)
        while v != EMPTY:
            old = tab[s]
            if old == v:
                break
            if old > v:
                tab[s] = v
                v = old
            s = (s + 1) & (slots - 1)
    return [-1 if x == EMPTY else x for x in tab]
