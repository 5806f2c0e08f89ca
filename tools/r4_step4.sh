#!/bin/bash
# round 4, step 4: device ingest tests, then configs[4]'s shape timed (SE) and a PE pair of half that size each
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
READS=${1:-33333334}
timeout 1500 python -m pytest tests/test_gpu_devingest.py -m gpu -x -q -k "not configs4 and not knobs" > gpurun_out/r4s4_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4s4_pytest.log
tail -4 gpurun_out/r4s4_pytest.log
T=/tmp/e2ef; mkdir -p $T
python tools/make_fastq.py $T/s --pairs $READS --mates 1 --block 2000000 > /dev/null
python tools/pgzip.py $T/s_1.fq $T/s.fq.gz --level 6
python tools/make_fastq.py $T/p --pairs $((READS/2)) --block 2000000 > /dev/null
python tools/pgzip.py $T/p_1.fq $T/p_1.fq.gz --level 6 & python tools/pgzip.py $T/p_2.fq $T/p_2.fq.gz --level 6 & wait
ls -l $T | awk '{print $5, $9}'
for cons in ${CONS:-3 1 4}; do
MF_INGEST_CONSUMERS=$cons MF_PIPE_TIMING=1 python - <<PY > gpurun_out/r4s4_e2e_c$cons.log 2>&1
import time, sys, os
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
ks = mf.KmerSet.from_fasta("$T/s.bait.fa", 31)
for i in range(3):
    t0 = time.time(); r = mf.filter_fastq_files(ks, "$T/s.fq.gz", None, "$T/o.fq", None); dt = time.time() - t0
    print(f"SE consumers $cons call {i}: {dt:7.3f} s  {r[1]/dt/1e6:6.2f} M reads/s  kept {r[0]} of {r[1]}", flush=True)
for i in range(3):
    t0 = time.time(); r = mf.filter_fastq_files(ks, "$T/p_1.fq.gz", "$T/p_2.fq.gz", "$T/o1.fq", "$T/o2.fq"); dt = time.time() - t0
    print(f"PE consumers $cons call {i}: {dt:7.3f} s  {2*r[1]/dt/1e6:6.2f} M reads/s  kept {r[0]} of {r[1]} pairs", flush=True)
PY
grep -E "call|wall" gpurun_out/r4s4_e2e_c$cons.log | cut -c1-330
done
rm -rf $T
