"""Read baiting for the `bim` loop (SURVEY.md 8f "next" #1).

The reference baits reads by alignment: `bwa_map(threads, fasta_file, basedir, prefix, fastq1,
fastq2, quality=30) -> (bam, fq1, fq2)` runs `bwa index; bwa mem | samtools view -q 30 |
samtools fastq` (bim/bim.py:43-58), and the loop in MitoFlex.py:346-375 hands the BAM to
`cal_insert(bam, basedir, prefix)` (bim/bim.py:65-78: `samtools stats | grep ^IS`, count-weighted mean
of the insert sizes) when `--insert-size-auto` is on, feeds the survivors to `assemble()` and uses the new
contigs as the next bait.

`kmer_bait_map` and `cal_insert` here have the same signatures and return shapes, so the loop runs
unchanged with `from mitoflex_amd.bim.bim import kmer_bait_map as bwa_map, cal_insert`:

* `kmer_bait_map` baits with the GPU k-mer filter -- the bait FASTA of every generation is turned into a
  canonical-k-mer set on the device and the reads are screened against it -- and puts, where the BAM path
  was, the path of a samtools-stats-shaped text file `<prefix>.bait.stats` whose `IS` lines hold an
  insert-size histogram of the kept pairs;
* `cal_insert` reads that file exactly the way the reference reads `samtools stats` (`^IS`, first two
  columns, count-weighted mean) and tees it to `<prefix>.stats` like the reference does.

The histogram comes from k-mer anchors, not alignments: for a kept pair whose mates each hold a bait
k-mer on opposite strands of the same bait record, the first such k-mer of each mate fixes where the
mate starts on the bait, and the outer distance is the insert size (what samtools calls TLEN for an
inward pair).  A sample of the first kept pairs is enough for a mean.

Parity with bwa/samtools is UNPINNED (un-vendored external tools, no reference test pins them): the two
select different read sets by design and the insert size is an estimate from exact k-mer anchors.
What can be stated is measured by tests/test_gpu_parity.py::test_bim_bait_sensitivity (reads drawn from
the bait with 1 % substitutions: > 99 % of pairs kept, no background pair) and by tests/test_bim.py
(insert-size estimate within a base or two of the truth on simulated pairs; a three-generation loop
with a stand-in assembler whose kept sets equal the oracle's and grow).
"""
from __future__ import annotations

from os import path
from typing import Dict, Iterator, List, Optional, Tuple

INSERT_SAMPLE_PAIRS = 20000        # kept pairs looked at for the insert-size histogram
_CODE = {"A": 0, "C": 1, "G": 2, "T": 3}
_COMP = str.maketrans("ACGTacgt", "TGCAtgca")


def _bait_records(fasta_file: str) -> List[str]:
    recs: List[List[str]] = []
    with open(fasta_file) as f:
        for line in f:
            if line.startswith(">"):
                recs.append([])
            elif recs:
                recs[-1].append("".join(line.split()).upper())
    return ["".join(r) for r in recs]


def _anchor_index(records: List[str], k: int) -> Dict[str, Tuple[int, int]]:
    """forward k-mer text -> (record, position); k-mers that occur twice are left out (ambiguous anchors)."""
    seen: Dict[str, Optional[Tuple[int, int]]] = {}
    for ri, rec in enumerate(records):
        for p in range(len(rec) - k + 1):
            w = rec[p:p + k]
            if w.strip("ACGT"):
                continue
            seen[w] = None if w in seen else (ri, p)
    return {w: v for w, v in seen.items() if v is not None}


def _first_anchor(seq: str, k: int, idx: Dict[str, Tuple[int, int]]):
    """(record, start of the read on the bait's forward strand, strand) from the first k-mer of the read that is a
    unique bait k-mer on either strand; None if there is none."""
    s = seq.upper()
    rc = s.translate(_COMP)[::-1]
    L = len(s)
    for o in range(L - k + 1):
        hit = idx.get(s[o:o + k])
        if hit is not None:                          # read lies forward: its first base sits at p - o
            return hit[0], hit[1] - o, +1
        hit = idx.get(rc[L - k - o:L - o])           # the same window on the other strand
        if hit is not None:                          # read lies reversed: the window's bait start is hit[1], the read's last base
            return hit[0], hit[1] - (L - k - o), -1  # maps to hit[1] - (bases of the read behind the window)
    return None


def _fastq_seqs(fq: str) -> Iterator[str]:
    with open(fq) as f:
        for i, line in enumerate(f):
            if i % 4 == 1:
                yield line.rstrip("\r\n")


def estimate_insert_sizes(fasta_file: str, fq1: str, fq2: str, kmer: int = 31, max_pairs: int = INSERT_SAMPLE_PAIRS) -> Dict[int, int]:
    """Insert-size histogram {size: pairs} of up to `max_pairs` pairs of (fq1, fq2) from exact k-mer anchors on the bait."""
    idx = _anchor_index(_bait_records(fasta_file), kmer)
    hist: Dict[int, int] = {}
    for n, (s1, s2) in enumerate(zip(_fastq_seqs(fq1), _fastq_seqs(fq2))):
        if n >= max_pairs:
            break
        a1, a2 = _first_anchor(s1, kmer, idx), _first_anchor(s2, kmer, idx)
        if a1 is None or a2 is None or a1[0] != a2[0] or a1[2] == a2[2]:
            continue
        fwd, rev, rev_len = (a1, a2, len(s2)) if a1[2] > 0 else (a2, a1, len(s1))
        size = rev[1] + rev_len - fwd[1]             # forward mate's first base .. reversed mate's last base on the bait
        if 0 < size <= 100000:
            hist[size] = hist.get(size, 0) + 1
    return hist


def kmer_bait_map(threads: int, fasta_file: str, basedir: str, prefix: str,
                  fastq1: str, fastq2: Optional[str], quality: int = 30,
                  kmer: int = 31, threshold: int = 1, devices: int = 1) -> Tuple[str, str, Optional[str]]:
    """Drop-in for `bwa_map`: returns (stats, fq1, fq2).  `stats` stands where the BAM path was and is what `cal_insert`
    takes; `threads` and `quality` are accepted for signature compatibility and ignored."""
    from mitoflex_amd import mitofilter as mf
    fq1 = path.join(basedir, prefix + ".1.fq")
    fq2 = path.join(basedir, prefix + ".2.fq") if fastq2 is not None else None
    ks = mf.KmerSet.from_fasta(fasta_file, kmer, 0)          # the device-side set builder, once per generation
    try:
        mf.filter_fastq_files(ks, fastq1, fastq2, fq1, fq2, threshold, mf.PAIR_EITHER, devices)
    finally:
        ks.close()
    stats = path.join(basedir, prefix + ".bait.stats")
    hist = estimate_insert_sizes(fasta_file, fq1, fq2, kmer) if fq2 is not None else {}
    with open(stats, "w") as f:
        f.write("# insert sizes of kept pairs from k-mer anchors on the bait (mitoflex_amd.bim); samtools-stats IS layout\n")
        for size in sorted(hist):
            f.write(f"IS\t{size}\t{hist[size]}\t{hist[size]}\t0\t0\n")
    return stats, fq1, fq2


def cal_insert(bam: str, basedir: str, prefix: str) -> float:
    """Mirror of bim/bim.py:65-78 over the stats file `kmer_bait_map` returned in the BAM's place: `^IS` lines, first two
    columns, count-weighted mean; the text is teed to `<prefix>.stats`.  Like the reference it fails (ZeroDivisionError)
    when there is no insert-size line at all."""
    stat_file = path.join(basedir, prefix + ".stats")
    text = open(bam).read()
    with open(stat_file, "w") as f:
        f.write(text)
    stats = [[int(y) for y in x.split("\t")[1:]][:2] for x in text.split("\n") if x.startswith("IS")]
    avg_ins = sum(a * b for a, b in stats) / sum(b for _, b in stats)
    return avg_ins
