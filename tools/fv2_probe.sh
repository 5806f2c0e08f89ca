#!/bin/bash
# filter_v2 -d on 2 M pairs .gz: library calls (warm), device path, and the bait filter on the same pair
cd $GRAFT_REPO_ROOT; T=/tmp/fv2p; mkdir -p $T
python tools/make_fastq.py $T/p --pairs ${1:-2000000} --block 2000000 > /dev/null; for m in 1 2; do python tools/pgzip.py $T/p_$m.fq $T/p_$m.fq.gz --level 6; done
python - <<PY
import time, os, sys
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
T="$T"
ts = []
for _ in range(6):
    t0 = time.time(); r = mf.qualfilter_files(T+"/p_1.fq.gz", T+"/p_2.fq.gz", T+"/o1.fq", T+"/o2.fq", dedup=True); ts.append(time.time()-t0)
print("filter_v2 -d, .gz pair, library:", r, " ".join(f"{t:.3f}" for t in ts), flush=True)
ts = []
for _ in range(6):
    t0 = time.time(); r = mf.qualfilter_files(T+"/p_1.fq", T+"/p_2.fq", T+"/o1.fq", T+"/o2.fq", dedup=True); ts.append(time.time()-t0)
print("filter_v2 -d, plain pair, library:", r, " ".join(f"{t:.3f}" for t in ts), flush=True)
ks = mf.KmerSet.from_fasta(T+"/p.bait.fa", 31)
ts = []
for _ in range(6):
    t0 = time.time(); r = mf.filter_fastq_files(ks, T+"/p_1.fq.gz", T+"/p_2.fq.gz", T+"/o1.fq", T+"/o2.fq"); ts.append(time.time()-t0)
print("bait filter, .gz pair:", r, " ".join(f"{t:.3f}" for t in ts), flush=True)
PY
rm -rf $T
