#!/bin/bash
# round 4, step 6: tests, then configs[4]'s shape timed at two chunk sizes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
READS=${1:-33333334}
timeout 1500 python -m pytest tests/test_gpu_devingest.py -m gpu -x -q -k "not configs4 and not knobs" > gpurun_out/r4s6_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4s6_pytest.log
tail -4 gpurun_out/r4s6_pytest.log
T=/tmp/e2ef; mkdir -p $T
python tools/make_fastq.py $T/s --pairs $READS --mates 1 --block 2000000 > /dev/null
python tools/pgzip.py $T/s_1.fq $T/s.fq.gz --level 6
for ck in 262144 524288; do
MF_GZDEV_CHUNK_BYTES=$ck MF_PIPE_TIMING=1 python - <<PY > gpurun_out/r4s6_e2e_$ck.log 2>&1
import time, sys, os
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
ks = mf.KmerSet.from_fasta("$T/s.bait.fa", 31)
for i in range(4):
    t0 = time.time(); r = mf.filter_fastq_files(ks, "$T/s.fq.gz", None, "$T/o.fq", None); dt = time.time() - t0
    print(f"SE chunk $ck call {i}: {dt:7.3f} s  {r[1]/dt/1e6:6.2f} M reads/s  kept {r[0]} of {r[1]}", flush=True)
PY
grep -E "call|wall" gpurun_out/r4s6_e2e_$ck.log | cut -c1-500
done
rm -rf $T
