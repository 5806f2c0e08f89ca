#!/usr/bin/env python3
"""Timing of protein-space baiting on the 5 Gbp PE150 shape: a small database
(k-mer bit table in LDS) and one the size of an MT_database clade (bit table in L2)."""
import json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mitoflex_amd import mitofilter as mf
mf.load()
import numpy as np
from tests.util_data import make_protein_bait
from oracle import prot_bait_ref as pr

n = int(sys.argv[1]) if len(sys.argv) > 1 else 33_333_334
small_fa, gene_fa = make_protein_bait()
rng = random.Random(5)
aas, w = "LSFIVGATMPYNWKEDHQRC", [16, 10, 9, 8, 7, 7, 6, 6, 5, 4, 4, 4, 3, 2, 2, 2, 2, 2, 2, 1]
big = ["".join(rng.choices(aas, weights=w, k=290)) for _ in range(4200)]          # ~1.2 M residues, like Mollusca.fa
big_fa = small_fa + "".join(f">p{i}\n{p}\n" for i, p in enumerate(big))
reads = mf.Reads.synth(n, 150, seed=20261003, bait_text=gene_fa, keep_host=True)
out = {}
for name, fa, kp in (("small_db_kp9", small_fa, 9), ("clade_size_db_kp9", big_fa, 9), ("small_db_kp7", small_fa, 7)):
    ks = mf.KmerSet.protein_from_text(fa, kp, 5)
    mf.filter_resident(ks, reads, 1, 0, 1)
    st = mf.filter_resident(ks, reads, 1, 0, 5)
    out[name] = {"keys": int(ks.info.n_keys), "ms_per_pass": round(st.ms_total, 3), "reads_per_s": n / (st.ms_total / 1e3),
                 "passed": int(st.n_pass), "codon_steps_per_s": n * 296 / (st.ms_total / 1e3)}
# CPU oracle on a sample, all host threads
from oracle import oracle_lib as ol
m = min(n, 2_000_000)
off = np.arange(m + 1, dtype=np.uint64) * 150
R = ol.OracleReads.from_arrays(reads.host_words, off, reads.host_npos[reads.host_npos < m * 150])
T = ol.OracleTable(small_fa, 9, protein=True)
t0 = time.perf_counter(); obits, _ = ol.pfilter_reads(T, R, 5, 1, threads=os.cpu_count()); dt = time.perf_counter() - t0
out["cpu_oracle"] = {"reads_per_s": m / dt, "threads": os.cpu_count(), "sample": m}
ks = mf.KmerSet.protein_from_text(small_fa, 9, 5)
out["sample_bits_match_oracle"] = bool(np.array_equal(mf.filter_reads(ks, reads, 1)[0][:m // 32], obits[:m // 32]))
print(json.dumps(out, indent=1))
