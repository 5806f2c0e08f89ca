#!/usr/bin/env python3
"""A few passes of the resident path over one synthetic read set, for profiling (tools/pmc_any.sh, rocprofv3 --kernel-trace --stats):
   python3 tools/run_passes.py [--reads N] [--bait-size B] [--k K] [--thr T] [--mode screened|exhaustive] [--hits] [--ragged] [--steps S] [--opts a=b,c=d]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitoflex_amd import mitofilter as mf
from mitoflex_amd.utility.synth_bait import make_bait, random_bait

ap = argparse.ArgumentParser()
ap.add_argument("--reads", type=int, default=33_333_334)
ap.add_argument("--bait-size", type=int, default=16569)
ap.add_argument("--k", type=int, default=31)
ap.add_argument("--thr", type=int, default=1)
ap.add_argument("--mode", default="screened")
ap.add_argument("--hits", action="store_true", help="ask for hit counts (the count-all form of the exact kernel)")
ap.add_argument("--ragged", action="store_true", help="cut the same stream into reads of 60..150 bases")
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--calls", type=int, default=2)
ap.add_argument("--opts", default="")
a = ap.parse_args()
for kv in [x for x in a.opts.split(",") if x]:
    n, v = kv.split("=")
    mf.set_option(n, v)
bait = make_bait() if a.bait_size == 16569 else random_bait(a.bait_size, seed=a.bait_size)
ks = mf.KmerSet.from_text(bait, a.k)
reads = mf.Reads.synth(a.reads, 150, 20261003, bait, keep_host=a.ragged)
if a.ragged:
    from mitoflex_amd.utility.synth_bait import ragged_offsets
    off = ragged_offsets(a.reads * 150)
    r2 = mf.Reads.from_packed(reads.host_words, off, reads.host_npos)
    reads.close()
    reads = r2
mode = mf.MODE_SCREENED if a.mode == "screened" else mf.MODE_EXHAUSTIVE
for _ in range(a.calls):
    if a.hits:
        _, _, st = mf.filter_reads(ks, reads, a.thr, mode, want_hits=True)
    else:
        st = mf.filter_resident(ks, reads, a.thr, mode, a.steps)
print(f"reads {reads.info.n_reads} bait {a.bait_size} k {a.k} T {a.thr} {a.mode}{' hits' if a.hits else ''}: {st.ms_total:.4f} ms/pass, pass {st.n_pass}, candidates {st.n_candidates}, "
      f"screen {st.ms_screen * 1e3:.1f} us mark {st.ms_mark * 1e3:.1f} last {st.ms_exact * 1e3:.1f}")
