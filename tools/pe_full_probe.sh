#!/bin/bash
# configs[1]'s shape through the file path: 16 666 667 pairs of 150 b, plain and .gz (pgzip -6), warm calls with MF_PIPE_TIMING
cd $GRAFT_REPO_ROOT; T=/tmp/pefull; mkdir -p $T
python tools/make_fastq.py $T/p --pairs ${1:-16666667} --block 2000000 > /dev/null
for m in 1 2; do python tools/pgzip.py $T/p_$m.fq $T/p_$m.fq.gz --level 6 & done; wait
ls -l $T | awk '{print $5, $9}'
MF_PIPE_TIMING=1 python - <<PY 2>&1 | grep "mf device ingest\] wall\|^call" | cut -c1-2600
import time, os, sys
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
T="$T"
ks = mf.KmerSet.from_fasta(T+"/p.bait.fa", 31)
for tag, a, b in (("plain", T+"/p_1.fq", T+"/p_2.fq"), ("gz", T+"/p_1.fq.gz", T+"/p_2.fq.gz")):
    for i in range(4):
        t0 = time.time(); r = mf.filter_fastq_files(ks, a, b, T+"/o1.fq", T+"/o2.fq"); print("call", tag, i, r, round(time.time()-t0, 4), mf.last_ingest_stats()["device_bytes_peak"] / 1e9, flush=True)
PY
rm -rf $T
