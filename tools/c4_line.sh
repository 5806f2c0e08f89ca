#!/bin/bash
# configs[4] (pgzip control): the MF_PIPE_TIMING line of warm calls
cd $GRAFT_REPO_ROOT; T=/tmp/c4l; mkdir -p $T gpurun_out/r05
python tools/make_fastq.py $T/s --pairs ${1:-33333334} --mates 1 --block 2000000 > /dev/null
python tools/pgzip.py $T/s_1.fq $T/s.fq.gz --level 6; rm $T/s_1.fq
MF_PIPE_TIMING=1 python - <<PY 2>&1 | grep "mf device ingest\] wall\|^call" | cut -c1-2000
import time, os, sys
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
T="$T"
ks = mf.KmerSet.from_fasta(T+"/s.bait.fa", 31)
for i in range(4):
    t0 = time.time(); r = mf.filter_fastq_files(ks, T+"/s.fq.gz", None, T+"/o.fq", None); print("call", i, r, round(time.time()-t0, 4), flush=True)
PY
rm -rf $T
