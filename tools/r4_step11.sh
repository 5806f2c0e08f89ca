#!/bin/bash
# tests, the small file, configs[4]'s shape
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_devingest.py -m gpu -x -q -k "not configs4 and not knobs" > gpurun_out/r4s11_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4s11_pytest.log; tail -3 gpurun_out/r4s11_pytest.log
MF_PIPE_TIMING=1 bash tools/e2e_small_gz.sh 500000 > gpurun_out/r4s11_small.log 2>&1; grep -E "^call|put away|first text" gpurun_out/r4s11_small.log | sed -e 's/.*| first text/first text/' | cut -c1-200 | head -12
T=/tmp/e2ef; mkdir -p $T
python tools/make_fastq.py $T/s --pairs 33333334 --mates 1 --block 2000000 > /dev/null
python tools/pgzip.py $T/s_1.fq $T/s.fq.gz --level 6
for v in "MF_INGEST_TEXT_BUFS=6" "MF_INGEST_TEXT_BUFS=10"; do
env $v MF_PIPE_TIMING=1 python - <<PY > gpurun_out/r4s11_e2e.log 2>&1
import time, sys, os
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
ks = mf.KmerSet.from_fasta("$T/s.bait.fa", 31)
for i in range(4):
    t0 = time.time(); r = mf.filter_fastq_files(ks, "$T/s.fq.gz", None, "$T/o.fq", None); dt = time.time() - t0
    print(f"SE $v call {i}: {dt:7.3f} s  {r[1]/dt/1e6:6.2f} M reads/s  kept {r[0]} of {r[1]}", flush=True)
PY
grep -E " call |put away|wall" gpurun_out/r4s11_e2e.log | cut -c1-120
done
rm -rf $T
