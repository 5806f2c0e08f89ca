"""Read baiting for the `bim` loop (SURVEY.md 8f "next" #1).

The reference baits reads by alignment: `bwa_map(threads, fasta_file, basedir, prefix, fastq1,
fastq2, quality=30) -> (bam, fq1, fq2)` runs `bwa index; bwa mem | samtools view -q 30 |
samtools fastq` (bim/bim.py:43-58) and the loop in MitoFlex.py:346-375 feeds the survivors to
`assemble()` and uses the new contigs as the next bait.  `kmer_bait_map` has the same signature
and return shape but baits with the GPU k-mer filter: the bait FASTA of every generation is turned
into a canonical-k-mer set on the device and the reads are screened against it.

Parity with bwa/samtools is UNPINNED (un-vendored external tools, no reference test pins them):
the two select different read sets by design.  What can be stated is measured by
tests/test_gpu_parity.py::test_bim_bait_sensitivity: on reads drawn from the bait with 1 %
substitutions the filter keeps > 99 % of pairs, and it keeps no background pair.
"""
from __future__ import annotations

from os import path
from typing import Optional, Tuple


def kmer_bait_map(threads: int, fasta_file: str, basedir: str, prefix: str,
                  fastq1: str, fastq2: Optional[str], quality: int = 30,
                  kmer: int = 31, threshold: int = 1, devices: int = 1) -> Tuple[None, str, Optional[str]]:
    """Drop-in for `bwa_map`: returns (None, fq1, fq2) -- there is no BAM; `threads` and `quality`
    are accepted for signature compatibility and ignored."""
    from mitoflex_amd import mitofilter as mf
    fq1 = path.join(basedir, prefix + ".1.fq")
    fq2 = path.join(basedir, prefix + ".2.fq") if fastq2 is not None else None
    ks = mf.KmerSet.from_fasta(fasta_file, kmer, 0)
    try:
        mf.filter_fastq_files(ks, fastq1, fastq2, fq1, fq2, threshold, mf.PAIR_EITHER, devices)
    finally:
        ks.close()
    return None, fq1, fq2
