# The bait-size axis: the same 5 Gbp resident read set against baits of 16.5 kbp .. 8.5 Mbp (random genomes; the last is the
# size of the reference's profile/MT_database read as nucleotides, 2.83 M residues x 3).  Per bait: set build time, table sizes,
# ms per pipelined pass, the pass's tallies, and whether the screened pass's bits equal the exhaustive pass's.
#   python tools/bait_sweep.py [reads] [sizes,comma,separated] [k]
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitoflex_amd import mitofilter as mf
from mitoflex_amd.utility.synth_bait import make_bait, random_bait

n = int(sys.argv[1]) if len(sys.argv) > 1 else 33_333_334
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [16569, 33000, 100000, 350000, 1000000, 8500000]
k = int(sys.argv[3]) if len(sys.argv) > 3 else 31
check = os.environ.get("SWEEP_CHECK", "1") == "1"
STEPS = int(os.environ.get("SWEEP_STEPS", "50"))          # passes a timed call, as bench.py's default K (a call pays ~0.3 ms of ramp and tail whatever K is)
print(f"# {STEPS} pipelined passes a timed call")
for kv in [x for x in os.environ.get("SWEEP_OPTS", "").split(",") if x]:          # e.g. SWEEP_OPTS=front=2,front2_log2b=17
    name, val = kv.split("=")
    mf.set_option(name, val)
HBM = 8000.0
for size in sizes:
    bait = make_bait() if size == 16569 else random_bait(size, seed=size)
    t0 = time.perf_counter()
    ks = mf.KmerSet.from_text(bait, k)
    t_build = time.perf_counter() - t0
    inf = ks.info
    reads = mf.Reads.synth(n, 150, 20261003, bait, mito_ppm=5000, sub_ppm=10000, n_read_ppm=10000, n_base_ppm=1000)
    os.environ["MF_EVENT_STRIDE"] = "1000000000"
    for _ in range(3):                       # (the pass kind follows what the previous calls of this read set saw)
        mf.filter_resident(ks, reads, 1, mf.MODE_SCREENED, 3)
    mf.device_synchronize(0)
    t0 = time.perf_counter()
    st = mf.filter_resident(ks, reads, 1, mf.MODE_SCREENED, STEPS)
    mf.device_synchronize(0)
    dt = (time.perf_counter() - t0) / STEPS
    os.environ["MF_EVENT_STRIDE"] = "1"
    sp = mf.filter_resident(ks, reads, 1, mf.MODE_SCREENED, STEPS)
    frac = st.algorithmic_bytes / dt / 1e9 / HBM
    line = (f"bait {size:>8d} bp: build {t_build:6.3f} s, keys {inf.n_keys:>9d}, s-mers {inf.n_smers:>9d}, stage-1 words {inf.bloom_words:>7d} | "
            f"{dt * 1e3:8.4f} ms/pass = {n / dt / 1e9:7.2f} G reads/s = {frac:5.3f} of HBM | screen {sp.ms_screen * 1e3:7.1f} us, mark {sp.ms_mark * 1e3:7.1f}, last {sp.ms_exact * 1e3:7.1f} | "
            f"work items {st.n_candidates:>9d} ({st.n_candidates / n:6.3f} a read), pass {st.n_pass}")
    if check:
        b1, _, _ = mf.filter_reads(ks, reads, 1, mf.MODE_SCREENED)
        t0 = time.perf_counter()
        b2, _, se = mf.filter_reads(ks, reads, 1, mf.MODE_EXHAUSTIVE)
        line += f" | exhaustive {se.ms_total:7.2f} ms, screened == exhaustive: {bool(np.array_equal(b1, b2))}"
    print(line, flush=True)
    reads.close()
    ks.close()
