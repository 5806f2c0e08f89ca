#!/usr/bin/env python3
"""Golden vectors for the k-mer bait filter from the STRING-LEVEL specification
oracle/kmer_bait_ref.py (Python strings and a set -- obviously right, slow), so that the GPU box compares the HIP path
with committed data and not only with a C oracle compiled on that same box.

PARITY UNPINNED BY THE REFERENCE: MitoFlex holds no k-mer read filter, so there is no reference output to capture.
This fixture pins the build's own Spec B -- nothing more.

Inputs are seeded (tests/util_data.py); their md5s are stored so that a drift of the generators is noticed.  Contents:
  pe10k   BASELINE.json configs[0] shape: 5 000 pairs x 2 x 150 b, 30 % bait-derived, k in {21, 31, 41}: per-read hit
          counts (uint16, base64), pass bitmaps for T in {1, 3}, pair-keep counts
  ragged  3 000 reads of mixed length with N / IUPAC / lower case / empty reads, k in {21, 31, 41}: hit counts
  edge    hand-made reads against a hand-made bait, k = 11 and 31: hit counts
  tables  bait table layouts (slots, number of keys, md5 of the little-endian u64 layout) for k in {11 .. 63}

    python tests/golden/make_kmer_bait_golden.py        (about a minute)
"""
import base64
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import kmer_bait_ref as ref  # noqa: E402
from tests.util_data import make_bait, make_reads  # noqa: E402

EDGE_BAIT = ">e1\nACGTTGCAACGTAGCTAGCTAGGATCCGATCGATTACGATCGGGCTATATATCGCGCGATATCGCTAGCTAGGCTAA\n>e2\nTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA\n"
EDGE_READS = [
    "", "A", "ACGTTGCAAC", "ACGTTGCAACG", "acgttgcaacgtagctagctaggatccgatcgattacgatcg",        # empty, shorter than k, exactly k = 11, lower case
    "ACGTTGCAACGTAGCTAGCTAGGATCCGATCGATTACGATCGGGCTATATATCGCGCGATATCGCTAGCTAGGCTAA",       # the whole record
    "TTAGCCTAGCTAGCGATATCGCGCGATATATAGCCCGATCGTAATCGATCGGATCCTAGCTAGCTACGTTGCAACGT",       # its reverse complement
    "ACGTTGCAACGTAGCTNGCTAGGATCCGATCGATTACGATCGGGCTATATATCGCGCGATATCGCTAGCTAGGCTAA",       # one N
    "ACGTTGCAACGTAGCTRGCTAGGATCCGATCGATTACGATCGGGCTATATATCGCGCGATATCGCTAGCTAGGCTAA",       # IUPAC code
    "T" * 60, "A" * 60, "T" * 30 + "A" * 30, "GGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGGG",
    "NNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNN",
    "CATGCATGCATGCATGCATGCATGCATGCATGCATGCATGCATGCATG",
]


def md5_of(seqs):
    h = hashlib.md5()
    for s in seqs:
        h.update(s.encode()); h.update(b"\n")
    return h.hexdigest()


def b64_u16(v):
    return base64.b64encode(np.asarray(v, dtype="<u2").tobytes()).decode()


def bitmap_hex(flags):
    b = np.packbits(np.asarray(flags, dtype=np.uint8), bitorder="little")
    return b.tobytes().hex()


def main():
    bait = make_bait()
    out = {"spec": "oracle/kmer_bait_ref.py", "generator": "tests/golden/make_kmer_bait_golden.py",
           "note": "PARITY UNPINNED BY THE REFERENCE (no k-mer read filter exists in MitoFlex); pins the build's own Spec B",
           "bait_md5": hashlib.md5(bait.encode()).hexdigest()}
    # ---- pe10k
    m1 = make_reads(bait, 5000, seed=101, uniform=True)
    m2 = make_reads(bait, 5000, seed=102, uniform=True)
    pe = {"pairs": 5000, "read_len": 150, "seed1": 101, "seed2": 102, "mate1_md5": md5_of(m1), "mate2_md5": md5_of(m2), "k": {}}
    for k in (21, 31, 41):
        bs = ref.bait_set(bait, k)
        h1 = [ref.read_hits(s, k, bs) for s in m1]
        h2 = [ref.read_hits(s, k, bs) for s in m2]
        e = {"hits1_u16": b64_u16(h1), "hits2_u16": b64_u16(h2), "T": {}}
        for T in (1, 3):
            p1, p2 = [h >= T for h in h1], [h >= T for h in h2]
            e["T"][str(T)] = {"pass1_bits": bitmap_hex(p1), "pass2_bits": bitmap_hex(p2),
                              "kept_either": sum(ref.pair_keep(p1, p2, "either")), "kept_both": sum(ref.pair_keep(p1, p2, "both"))}
        pe["k"][str(k)] = e
        print("pe10k k=%d: %d / %d mates with a hit" % (k, sum(h > 0 for h in h1), sum(h > 0 for h in h2)), file=sys.stderr)
    out["pe10k"] = pe
    # ---- ragged
    rg = make_reads(bait, 3000, seed=7)
    sets = {k: ref.bait_set(bait, k) for k in (21, 31, 41)}
    out["ragged"] = {"n": 3000, "seed": 7, "md5": md5_of(rg),
                     "k": {str(k): b64_u16([ref.read_hits(s, k, sets[k]) for s in rg]) for k in (21, 31, 41)}}
    # ---- edge
    out["edge"] = {"bait": EDGE_BAIT, "reads": EDGE_READS,
                   "k": {str(k): (lambda bs: [ref.read_hits(s, k, bs) for s in EDGE_READS])(ref.bait_set(EDGE_BAIT, k)) for k in (11, 31)}}
    # ---- tables
    tb = {}
    for k in (11, 15, 21, 31, 32, 33, 41, 63):
        lay = ref.table_layout(bait, k)
        words = 1 if k <= 32 else 2
        raw = b"".join((((1 << (64 * words)) - 1) if v < 0 else v).to_bytes(8 * words, "little") for v in lay)
        tb[str(k)] = {"slots": len(lay), "n_keys": sum(v >= 0 for v in lay), "md5": hashlib.md5(raw).hexdigest()}
    out["tables"] = tb
    with open(os.path.join(HERE, "kmer_bait_golden.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
