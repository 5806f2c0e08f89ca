#!/usr/bin/env python3
"""Fast synthetic PE FASTQ writer (fixed-width records, numpy): n pairs of L-base reads, a fraction
sampled from the bait (both strands, substitutions), a few N.  Used for end-to-end timing only."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitoflex_amd.utility.synth_bait import bait_records, make_bait  # noqa: E402


def write_mate(path, n, L, seed, mate, bait_seq, mito_frac, sub_rate, qual="uniform", first=0, append=False):
    """records first .. first + n - 1 (written in blocks: the generator is not the same stream as one big call)"""
    rng = np.random.default_rng(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    seq = acgt[rng.integers(0, 4, size=(n, L), dtype=np.uint8)]
    g = np.frombuffer(bait_seq.encode(), dtype=np.uint8)
    comp = np.zeros(256, dtype=np.uint8); comp[list(b"ACGT")] = list(b"TGCA")
    is_mito = np.random.default_rng(seed // 10).random(n) < mito_frac          # same choice for both mates
    idx = np.nonzero(is_mito)[0]
    pos = rng.integers(0, len(g) - L, size=len(idx))
    win = g[pos[:, None] + np.arange(L)[None, :]]
    rc = rng.random(len(idx)) < 0.5
    win[rc] = comp[win[rc][:, ::-1]]
    sub = rng.random(win.shape) < sub_rate
    win[sub] = acgt[rng.integers(0, 4, size=int(sub.sum()), dtype=np.uint8)]
    seq[idx] = win
    nmask = rng.random(n) < 0.01
    seq[np.nonzero(nmask)[0], rng.integers(0, L, size=int(nmask.sum()))] = ord("N")
    hdr = np.char.add(np.char.add("@syn.", np.char.zfill((first + np.arange(n)).astype(str), 9)), "/%d" % mate).astype("S")
    hl = hdr.dtype.itemsize
    rec = np.empty((n, hl + 1 + L + 3 + L + 1), dtype=np.uint8)
    rec[:, :hl] = hdr.view(np.uint8).reshape(n, hl)
    rec[:, hl] = 10
    rec[:, hl + 1: hl + 1 + L] = seq
    rec[:, hl + 1 + L: hl + 4 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
    if qual == "binned":          # four quality bins, mostly the top one (what current instruments write); the default is eighteen equiprobable values
        bins = np.frombuffer(b"F:,#", dtype=np.uint8)
        rec[:, hl + 4 + L: hl + 4 + 2 * L] = bins[np.searchsorted(np.array([0.88, 0.94, 0.98]), rng.random((n, L), dtype=np.float32))]
    else:
        rec[:, hl + 4 + L: hl + 4 + 2 * L] = rng.integers(ord("8"), ord("J"), size=(n, L), dtype=np.uint8)
    rec[:, -1] = 10
    with open(path, "ab" if append else "wb") as f:
        rec.tofile(f)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("prefix"); ap.add_argument("--pairs", type=int, default=1_000_000); ap.add_argument("--len", type=int, default=150)
    ap.add_argument("--mito", type=float, default=0.005); ap.add_argument("--seed", type=int, default=12340)
    ap.add_argument("--qual", choices=["uniform", "binned"], default="uniform")
    ap.add_argument("--mates", type=int, default=2, help="1: only mate 1 (single-end sets)")
    ap.add_argument("--block", type=int, default=0, help="write in blocks of this many records (memory bound for large sets)")
    ap.add_argument("--only-mate", type=int, default=0, help="write this mate's file only (1 or 2): very large pairs are made and compressed one file at a time")
    a = ap.parse_args()
    bait = make_bait()
    open(a.prefix + ".bait.fa", "w").write(bait)
    g = bait_records(bait)[0]
    for mate in (1, 2)[:a.mates]:
        if a.only_mate and mate != a.only_mate:
            continue
        if a.block and a.pairs > a.block:          # big files: block by block (a block's seed follows from its index)
            for b, first in enumerate(range(0, a.pairs, a.block)):
                write_mate(f"{a.prefix}_{mate}.fq", min(a.block, a.pairs - first), a.len, a.seed + mate + 1000 * (b + 1), mate, g, a.mito, 0.01, a.qual,
                           first=first, append=b > 0)
        else:
            write_mate(f"{a.prefix}_{mate}.fq", a.pairs, a.len, a.seed + mate, mate, g, a.mito, 0.01, a.qual)
