#!/bin/bash
# CUs kept free of decode work: 32 / 64 / 96, paired and single-end
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=/tmp/e2ep; mkdir -p $T
python tools/make_fastq.py $T/p --pairs 16666667 --block 2000000 > /dev/null
python tools/pgzip.py $T/p_1.fq $T/p_1.fq.gz --level 6 & python tools/pgzip.py $T/p_2.fq $T/p_2.fq.gz --level 6 & wait
cat $T/p_1.fq $T/p_2.fq > $T/s_1.fq; python tools/pgzip.py $T/s_1.fq $T/s.fq.gz --level 6
for v in "MF_GZDEV_RESERVED_CUS=32" "MF_GZDEV_RESERVED_CUS=64" "MF_GZDEV_RESERVED_CUS=96"; do
env $v python - <<PY 2>&1 | grep call
import time, sys, os
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
ks = mf.KmerSet.from_fasta("$T/p.bait.fa", 31)
for i in range(4):
    t0 = time.time(); r = mf.filter_fastq_files(ks, "$T/p_1.fq.gz", "$T/p_2.fq.gz", "$T/o1.fq", "$T/o2.fq"); dt = time.time() - t0
    if i: print(f"PE $v call {i}: {dt:7.3f} s  {2*r[1]/dt/1e6:6.2f} M reads/s", flush=True)
for i in range(4):
    t0 = time.time(); r = mf.filter_fastq_files(ks, "$T/s.fq.gz", None, "$T/o.fq", None); dt = time.time() - t0
    if i: print(f"SE $v call {i}: {dt:7.3f} s  {r[1]/dt/1e6:6.2f} M reads/s", flush=True)
PY
done
rm -rf $T
