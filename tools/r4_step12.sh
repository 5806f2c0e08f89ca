#!/bin/bash
# the driver's bench command under rocprofv3 (kernel stats, traffic), then a paired .gz input timed
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04
sed -e 's#gpurun_out/r03#gpurun_out/r04#' tools/prof_bench.sh > /tmp/prof_bench_r04.sh; sed -i 's/--no-group-a --no-live-traffic"/--no-group-a --no-live-traffic --real-gz-reads 0 --k-sweep none"/' /tmp/prof_bench_r04.sh; rm -f gpurun_out/r04/bench_traffic.txt; bash /tmp/prof_bench_r04.sh > gpurun_out/r04/prof_bench.log 2>&1; tail -3 gpurun_out/r04/prof_bench.log | cut -c1-200
T=/tmp/e2ep; mkdir -p $T
python tools/make_fastq.py $T/p --pairs 16666667 --block 2000000 > /dev/null
python tools/pgzip.py $T/p_1.fq $T/p_1.fq.gz --level 6 & python tools/pgzip.py $T/p_2.fq $T/p_2.fq.gz --level 6 & wait
for v in "MF_INGEST_CONSUMERS=3" "MF_INGEST_CONSUMERS=5 MF_INGEST_TEXT_BUFS=8"; do
env $v MF_PIPE_TIMING=1 python - <<PY > gpurun_out/r4s12_pe.log 2>&1
import time, sys, os
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
ks = mf.KmerSet.from_fasta("$T/p.bait.fa", 31)
for i in range(3):
    t0 = time.time(); r = mf.filter_fastq_files(ks, "$T/p_1.fq.gz", "$T/p_2.fq.gz", "$T/o1.fq", "$T/o2.fq"); dt = time.time() - t0
    print(f"PE $v call {i}: {dt:7.3f} s  {2*r[1]/dt/1e6:6.2f} M reads/s  kept {r[0]} of {r[1]} pairs", flush=True)
PY
grep -E " call |wall" gpurun_out/r4s12_pe.log | tail -3 | sed -e 's/ | buffers of this call.*device(s)//' | cut -c1-900
done
rm -rf $T
