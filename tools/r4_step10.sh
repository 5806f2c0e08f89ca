#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
READS=${1:-33333334}
T=/tmp/e2ef; mkdir -p $T
python tools/make_fastq.py $T/s --pairs $READS --mates 1 --block 2000000 > /dev/null
python tools/pgzip.py $T/s_1.fq $T/s.fq.gz --level 6
for v in ${VARIANTS:-"MF_GZDEV_CHUNK_BYTES=262144" "MF_GZDEV_CHUNK_BYTES=524288"}; do
env $v MF_GZDEV_CHUNK_BYTES=262144 MF_PIPE_TIMING=1 python - <<PY > gpurun_out/r4s10_e2e.log 2>&1
import time, sys, os
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
ks = mf.KmerSet.from_fasta("$T/s.bait.fa", 31)
for i in range(3):
    t0 = time.time(); r = mf.filter_fastq_files(ks, "$T/s.fq.gz", None, "$T/o.fq", None); dt = time.time() - t0
    print(f"SE $v call {i}: {dt:7.3f} s  {r[1]/dt/1e6:6.2f} M reads/s  kept {r[0]} of {r[1]}", flush=True)
PY
grep -E "call|wall" gpurun_out/r4s10_e2e.log | tail -4 | sed -e 's/buffers of this call.*1 device(s) //' | cut -c1-640
done
rm -rf $T
