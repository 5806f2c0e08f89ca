#!/bin/bash
# mid-size .gz input (2 M pairs, 0.3 GB a mate): the decoder's chunk size against the call's time -- quality filter (-d) and bait filter, warm process
R=$GRAFT_REPO_ROOT; T=/tmp/csp; mkdir -p $T
python $R/tools/make_fastq.py $T/q --pairs 2000000 > /dev/null
for m in 1 2; do python $R/tools/pgzip.py $T/q_$m.fq $T/q_$m.fq.gz --level 6; done
cd $R; python - <<PY
import os, time
from mitoflex_amd import mitofilter as mf
T="$T"
ks = mf.KmerSet.from_fasta(T + "/q.bait.fa", k=31)
def best(f, reps=4):
    b=1e9
    for _ in range(reps):
        t0=time.perf_counter(); f(); b=min(b,time.perf_counter()-t0)
    return b
for ck in ("", "98304", "131072", "262144", ""):
    if ck: os.environ["MF_GZDEV_CHUNK_BYTES"]=ck
    else: os.environ.pop("MF_GZDEV_CHUNK_BYTES", None)
    q=best(lambda: mf.qualfilter_files(T+"/q_1.fq.gz", T+"/q_2.fq.gz", T+"/o1.fq", T+"/o2.fq", dedup=True))
    b=best(lambda: mf.filter_fastq_files(ks, T+"/q_1.fq.gz", T+"/q_2.fq.gz", T+"/b1.fq", T+"/b2.fq"))
    s=best(lambda: mf.filter_fastq_files(ks, T+"/q_1.fq.gz", None, T+"/b1.fq", None))
    print("chunk", ck or "default(64K)", "quality filter PE -d %.4f s | bait filter PE %.4f s | bait filter SE %.4f s" % (q, b, s), flush=True)
PY
rm -rf $T
