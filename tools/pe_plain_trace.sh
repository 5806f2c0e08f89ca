#!/bin/bash
cd $GRAFT_REPO_ROOT; T=/tmp/pept; mkdir -p $T
python tools/make_fastq.py $T/p --pairs ${1:-16666667} --block 2000000 > /dev/null
MF_PIPE_TIMING=1 MF_DEVINGEST_TRACE=1 python - <<PY 2>&1 | grep "plain producer\|mf device ingest\] wall\|^call\|consumer [0-9]: piece\|filtered: records" | tail -150 | cut -c1-400
import time, os, sys
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
T="$T"
ks = mf.KmerSet.from_fasta(T+"/p.bait.fa", 31)
for i in range(3):
    t0 = time.time(); r = mf.filter_fastq_files(ks, T+"/p_1.fq", T+"/p_2.fq", T+"/o1.fq", T+"/o2.fq"); print("call", i, r, round(time.time()-t0, 4), flush=True)
PY
rm -rf $T
