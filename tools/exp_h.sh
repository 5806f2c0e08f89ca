#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; T=/tmp/e2e; mkdir -p $T
python tools/make_fastq.py $T/s --pairs ${PAIRS:-4000000}
gzip -1 -c $T/s_1.fq > $T/s_1.fq.gz
nproc; cat /sys/fs/cgroup/cpu.max
python - <<PY
import time, os, sys
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
T="$T"
ks = mf.KmerSet.from_fasta(T+"/s.bait.fa", 31)
for rep in range(2):
    t0=time.time(); k,t = mf.filter_fastq_files(ks, T+"/s_1.fq.gz", None, T+"/og.fq", None); dt=time.time()-t0
    print("SE gz: %.2f s %.2f M reads/s" % (dt, t/dt/1e6), flush=True)
t0=time.time(); k,t = mf.filter_fastq_files(ks, T+"/s_1.fq", None, T+"/o.fq", None); dt=time.time()-t0
print("SE plain: %.2f s %.2f M reads/s" % (dt, t/dt/1e6))
PY
rm -rf $T
