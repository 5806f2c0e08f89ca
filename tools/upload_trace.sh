#!/bin/bash
# where a plain 10.7 GB file's call spends its time, uploads from the registered mapping vs staged (MF_DEVINGEST_TRACE)
cd $GRAFT_REPO_ROOT; T=/tmp/uptr; mkdir -p $T
python tools/make_fastq.py $T/s --pairs ${1:-33333334} --mates 1 --block 2000000 > /dev/null
for mode in registered staged; do
python - $mode <<PY 2>&1 | grep "registered mapping\|plain producer\|^=====\|^registered\|^staged" 
import time, os, sys
sys.path.insert(0, ".")
mode = sys.argv[1]
if mode == "staged": os.environ["MF_UPLOAD_STAGED"] = "1"
os.environ["MF_DEVINGEST_TRACE"] = "1"
from mitoflex_amd import mitofilter as mf
T="$T"
ks = mf.KmerSet.from_fasta(T+"/s.bait.fa", 31)
f = T+"/s_1.fq"
mf.filter_fastq_files(ks, f, None, T+"/o.fq", None)
mf.filter_fastq_files(ks, f, None, T+"/o.fq", None)
print("=====", mode, flush=True)
t0 = time.time(); r = mf.filter_fastq_files(ks, f, None, T+"/o.fq", None); print(mode, r, time.time() - t0, mf.last_ingest_stats(), flush=True)
PY
done
rm -rf $T
