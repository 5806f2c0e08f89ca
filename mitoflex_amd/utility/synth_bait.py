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
