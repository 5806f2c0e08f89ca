#!/bin/bash
# round 4, step 5: tests again, configs[4]'s shape timed with 1 and 3 consumers, a trace of one call, the box's copy and read rates
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
READS=${1:-33333334}
timeout 1500 python -m pytest tests/test_gpu_devingest.py -m gpu -x -q -k "not configs4 and not knobs" > gpurun_out/r4s5_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4s5_pytest.log
tail -4 gpurun_out/r4s5_pytest.log
T=/tmp/e2ef; mkdir -p $T
python tools/make_fastq.py $T/s --pairs $READS --mates 1 --block 2000000 > /dev/null
python tools/pgzip.py $T/s_1.fq $T/s.fq.gz --level 6
python tools/make_fastq.py $T/p --pairs $((READS/2)) --block 2000000 > /dev/null
python tools/pgzip.py $T/p_1.fq $T/p_1.fq.gz --level 6 & python tools/pgzip.py $T/p_2.fq $T/p_2.fq.gz --level 6 & wait
for cons in ${CONS:-1 3}; do
MF_INGEST_CONSUMERS=$cons MF_PIPE_TIMING=1 python - <<PY > gpurun_out/r4s5_e2e_c$cons.log 2>&1
import time, sys, os
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
ks = mf.KmerSet.from_fasta("$T/s.bait.fa", 31)
print("H2D GB/s (1 GiB pinned):", mf.h2d_bandwidth(0, 1 << 30, 3), " (32 MiB):", mf.h2d_bandwidth(0, 32 << 20, 5))
for i in range(3):
    t0 = time.time(); r = mf.filter_fastq_files(ks, "$T/s.fq.gz", None, "$T/o.fq", None); dt = time.time() - t0
    print(f"SE consumers $cons call {i}: {dt:7.3f} s  {r[1]/dt/1e6:6.2f} M reads/s  kept {r[0]} of {r[1]}", mf.last_ingest_stats(), flush=True)
for i in range(3):
    t0 = time.time(); r = mf.filter_fastq_files(ks, "$T/p_1.fq.gz", "$T/p_2.fq.gz", "$T/o1.fq", "$T/o2.fq"); dt = time.time() - t0
    print(f"PE consumers $cons call {i}: {dt:7.3f} s  {2*r[1]/dt/1e6:6.2f} M reads/s  kept {r[0]} of {r[1]} pairs", flush=True)
PY
grep -E "call|wall|H2D" gpurun_out/r4s5_e2e_c$cons.log | cut -c1-560
done
MF_INGEST_CONSUMERS=3 MF_DEVINGEST_TRACE=1 python - <<PY > gpurun_out/r4s5_trace.log 2>&1
import time, sys, os
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
ks = mf.KmerSet.from_fasta("$T/s.bait.fa", 31)
for i in range(2):
    print("=== call", i, flush=True)
    r = mf.filter_fastq_files(ks, "$T/s.fq.gz", None, "$T/o.fq", None)
PY
python - <<PY
import time, os
t0=time.time(); n=0
fd=os.open("$T/s.fq.gz", os.O_RDONLY)
import threading
def rd(a,b):
    off=a
    while off<b:
        d=os.pread(fd, min(32<<20,b-off), off); off+=len(d)
sz=os.path.getsize("$T/s.fq.gz"); th=[threading.Thread(target=rd,args=(sz*i//8, sz*(i+1)//8)) for i in range(8)]
[t.start() for t in th]; [t.join() for t in th]
print("pread 8 threads from the page cache: %.2f GB/s" % (sz/(time.time()-t0)/1e9))
PY
rm -rf $T
