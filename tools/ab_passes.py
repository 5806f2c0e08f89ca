#!/usr/bin/env python3
"""ms per pipelined pass of the resident path at a few (k, bait size) points, for same-box A/B of library variants:
   MITOFILTER_LIB=.../libmitofilter_hip_x.so python3 tools/ab_passes.py "31:16569,21:16569,31:100000" [reads]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mitoflex_amd import mitofilter as mf
from mitoflex_amd.utility.synth_bait import make_bait, random_bait

points = [(int(a), int(b)) for a, b in (p.split(":") for p in sys.argv[1].split(","))]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 33_333_334
os.environ["MF_EVENT_STRIDE"] = "1000000000"
reads, cur = None, None
out = []
for k, size in points:
    bait = make_bait() if size == 16569 else random_bait(size, seed=size)
    if cur != size:
        if reads is not None:
            reads.close()
        reads = mf.Reads.synth(n, 150, 20261003, bait)
        cur = size
    ks = mf.KmerSet.from_text(bait, k)
    for _ in range(3):
        mf.filter_resident(ks, reads, 1, mf.MODE_SCREENED, 10)
    best = 1e9
    for _ in range(3):
        mf.device_synchronize(0)
        t0 = time.perf_counter()
        mf.filter_resident(ks, reads, 1, mf.MODE_SCREENED, 20)
        mf.device_synchronize(0)
        best = min(best, (time.perf_counter() - t0) / 20)
    os.environ["MF_EVENT_STRIDE"] = "1"
    sp = mf.filter_resident(ks, reads, 1, mf.MODE_SCREENED, 10)
    os.environ["MF_EVENT_STRIDE"] = "1000000000"
    out.append(f"k={k} bait={size}: {best * 1e3:.4f} ms/pass (screen {sp.ms_screen * 1e3:.1f} us)")
    ks.close()
print(" | ".join(out))
