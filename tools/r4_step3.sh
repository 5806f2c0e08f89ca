#!/bin/bash
# round 4, step 3: the lane-parallel decode kernel against zlib (tools/gzdev_check), the device ingest tests on it, and configs[4]'s
# shape timed with either kernel
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
READS=${1:-33333334}
( CHUNKS="256 64" timeout 900 bash tools/gzdev_run.sh 400000 ) > gpurun_out/r4s3_gzdev_lanes.log 2>&1
grep -E "PASS|FAIL|kernel .* ms|failed" gpurun_out/r4s3_gzdev_lanes.log | tail -30
( MF_GZDEV_KERNEL=serial CHUNKS="256" timeout 900 bash tools/gzdev_run.sh 400000 ) > gpurun_out/r4s3_gzdev_serial.log 2>&1
grep -E "PASS|FAIL|kernel .* ms" gpurun_out/r4s3_gzdev_serial.log | tail -12
timeout 1500 python -m pytest tests/test_gpu_devingest.py -m gpu -x -q -k "not configs4 and not knobs" > gpurun_out/r4s3_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4s3_pytest.log
tail -4 gpurun_out/r4s3_pytest.log
T=/tmp/e2ef; mkdir -p $T
python tools/make_fastq.py $T/s --pairs $READS --mates 1 --block 2000000 > /dev/null
python tools/pgzip.py $T/s_1.fq $T/s.fq.gz --level 6
ls -l $T | awk '{print $5, $9}'
for kern in lanes serial; do
MF_GZDEV_KERNEL=$kern MF_PIPE_TIMING=1 python - <<PY > gpurun_out/r4s3_e2e_$kern.log 2>&1
import time, sys, os
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
ks = mf.KmerSet.from_fasta("$T/s.bait.fa", 31)
for i in range(3):
    t0 = time.time(); r = mf.filter_fastq_files(ks, "$T/s.fq.gz", None, "$T/o.fq", None); dt = time.time() - t0
    print(f"$kern call {i}: {dt:7.3f} s  {r[1]/dt/1e6:6.2f} M reads/s  kept {r[0]} of {r[1]}", flush=True)
PY
grep -E "call|wall" gpurun_out/r4s3_e2e_$kern.log | cut -c1-420
done
rm -rf $T
