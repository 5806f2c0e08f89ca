#!/bin/bash
# a small .gz file (1 M reads) through the device ingest path, three calls in one process: where the time of a short call goes
cd $GRAFT_REPO_ROOT; T=/tmp/e2es; mkdir -p $T
python tools/make_fastq.py $T/s --pairs ${1:-1000000} --mates 1 > /dev/null
gzip -6 -c $T/s_1.fq > $T/s.fq.gz
python - <<PY
import time, sys, os
sys.path.insert(0, ".")
os.environ["MF_PIPE_TIMING"] = "1"
from mitoflex_amd import mitofilter as mf
ks = mf.KmerSet.from_fasta("$T/s.bait.fa", 31)
for i in range(4):
    t0 = time.time(); r = mf.filter_fastq_files(ks, "$T/s.fq.gz", None, "$T/o.fq", None); dt = time.time() - t0
    print(f"call {i}: {dt*1e3:7.1f} ms  {r[1]/dt/1e6:6.2f} M reads/s", flush=True)
os.environ["MF_INGEST"] = "host"
for i in range(2):
    t0 = time.time(); r = mf.filter_fastq_files(ks, "$T/s.fq.gz", None, "$T/o.fq", None); dt = time.time() - t0
    print(f"host pipeline call {i}: {dt*1e3:7.1f} ms  {r[1]/dt/1e6:6.2f} M reads/s", flush=True)
PY
rm -rf $T
