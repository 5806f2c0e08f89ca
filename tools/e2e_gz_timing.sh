#!/bin/bash
cd $GRAFT_REPO_ROOT; T=/tmp/e2e; mkdir -p $T
python tools/make_fastq.py $T/s --pairs 8000000 > /dev/null
gzip -1 -c $T/s_1.fq > $T/s_1.fq.gz; ls -la $T/s_1.fq.gz $T/s_1.fq
python - 2> $GRAFT_REPO_ROOT/gpurun_out/gz_timing.err <<PY
import time, os, sys, resource
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
T="$T"
ks = mf.KmerSet.from_fasta(T+"/s.bait.fa", 31)
os.environ["MF_PIPE_TIMING"]="1"; os.environ["MF_GZ_TIMING"]="1"
mf.filter_fastq_files(ks, T+"/s_1.fq.gz", None, T+"/og_se.fq", None)
r0 = resource.getrusage(resource.RUSAGE_SELF); t0 = time.time()
kept, total = mf.filter_fastq_files(ks, T+"/s_1.fq.gz", None, T+"/og_se.fq", None)
dt = time.time() - t0; r1 = resource.getrusage(resource.RUSAGE_SELF)
print("SE gz %.2f M reads/s wall %.3f s user %.2f s sys %.2f s" % (total/dt/1e6, dt, r1.ru_utime - r0.ru_utime, r1.ru_stime - r0.ru_stime))
PY
rm -rf $T
