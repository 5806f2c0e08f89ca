"""The synthetic bait of the benchmark and of the tests: a seeded 16 569 bp 'mitogenome' plus a short second record
("sampled from a synthetic 16 569 bp bait genome").  Lives in the package so that bench.py and the tools do not
import from tests/."""
from __future__ import annotations

import random
from typing import List


def make_bait(seed: int = 20261003, length: int = 16569, second_record: int = 1200) -> str:
    """A synthetic 16 569 bp 'mitogenome' plus a short second record with IUPAC codes,
    lower case, a poly-T run and an N block, written as multi-line FASTA with CRLF on one line."""
    rng = random.Random(seed)
    g = "".join(rng.choices("ACGT", k=length))
    g = g[:5000] + "T" * 40 + g[5040:9000] + "A" * 35 + g[9035:]          # homopolymers (all-ones / all-zero s-mers)
    r2 = list("".join(rng.choices("ACGT", k=second_record)))
    for p in (100, 101, 102, 500, 777):
        r2[p] = "N"
    r2[300] = "R"; r2[301] = "y"
    r2 = "".join(r2)
    r2 = r2[:600] + r2[600:900].lower() + r2[900:]
    lines = [">synthetic_mito_1 len=%d" % length]
    lines += [g[i:i + 70] for i in range(0, len(g), 70)]
    lines += [">rec2 with iupac"]
    lines += [r2[i:i + 60] for i in range(0, len(r2), 60)]
    lines[3] += "\r"
    return "\n".join(lines) + "\n"


def bait_records(bait_text: str) -> List[str]:
    recs, cur = [], None
    for ln in bait_text.split("\n"):
        if ln.startswith(">"):
            cur = []; recs.append(cur)
        elif cur is not None:
            cur.append("".join(ln.split()))
    return ["".join(r) for r in recs]


def random_bait(length: int, seed: int = 1, n_records: int = 0, line: int = 70) -> str:
    """A seeded random nucleotide bait of `length` bases in all (the bait-size axis of bench.py / tools/bait_sweep.py: 33 kbp =
    two mitogenomes ... 8.5 Mbp = the reference's profile/MT_database read as nucleotides), cut into records of about
    16.5 kbp (a clade-wide bait is many mitogenomes, not one long sequence) unless n_records says otherwise."""
    import numpy as np
    rng = np.random.default_rng(seed)
    codes = rng.integers(0, 4, size=length, dtype=np.uint8)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[codes].tobytes().decode()
    if n_records <= 0:
        n_records = max(1, round(length / 16569))
    per = (length + n_records - 1) // n_records
    out = []
    for r in range(n_records):
        rec = seq[r * per:(r + 1) * per]
        if not rec:
            break
        out.append(">random_bait_%d len=%d" % (r, len(rec)))
        out.extend(rec[i:i + line] for i in range(0, len(rec), line))
    return "\n".join(out) + "\n"


def ragged_offsets(total_bases: int, lo: int = 60, hi: int = 150, seed: int = 7):
    """base offsets that cut a dense stream of `total_bases` into reads of lo..hi bases (what `filter_v2 -s/-e` and its quality
    cuts leave of 150-base reads, /root/reference filter/filter_bin/src/main.rs:239-268 in spirit): u64[n + 1], offsets[0] = 0,
    offsets[n] = total_bases"""
    import numpy as np
    rng = np.random.default_rng(seed)
    n_est = int(total_bases / ((lo + hi) / 2) * 1.02) + 16
    lens = rng.integers(lo, hi + 1, size=n_est, dtype=np.int64)
    off = np.concatenate(([0], np.cumsum(lens)))
    n = int(np.searchsorted(off, total_bases, side="left"))
    off = off[:n + 1].copy()
    off[n] = total_bases                       # (the last read takes what is left: it may be shorter than lo)
    return off.astype(np.uint64)


def realistic_bait(seed: int = 20261004, length: int = 16569) -> str:
    """A mitogenome-shaped bait for the low-complexity leg of bench.py: 68 % A+T (animal mitogenomes run 60-80 %), poly-T and poly-A runs
    and a (TA)n stretch of the kind the control region carries; one record."""
    rng = random.Random(seed)
    g = "".join(rng.choices("ACGT", weights=[34, 16, 16, 34], k=length))
    g = g[:3000] + "T" * 40 + g[3040:7000] + "A" * 35 + g[7035:15500] + "TA" * 60 + g[15620:]
    g = g[:16000] + "T" * 18 + "C" + "T" * 14 + g[16033:]
    lines = [">realistic_mito len=%d" % len(g)] + [g[i:i + 70] for i in range(0, len(g), 70)]
    return "\n".join(lines) + "\n"
