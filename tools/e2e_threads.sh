#!/bin/bash
# file-to-file rate against the number of host threads (the GPU box runs under a CPU quota far below its thread count)
cd $GRAFT_REPO_ROOT; T=/tmp/e2e; mkdir -p $T
PAIRS=${1:-8000000}
python tools/make_fastq.py $T/s --pairs $PAIRS > /dev/null
( gzip -1 -c $T/s_1.fq > $T/s_1.fq.gz ) & ( gzip -1 -c $T/s_2.fq > $T/s_2.fq.gz ) & wait
cat /sys/fs/cgroup/cpu.max 2>/dev/null
for th in ${THREADS_LIST:-default 8 12 16 20 24 32 48}; do
if [ $th = default ]; then unset MF_PACK_THREADS; else export MF_PACK_THREADS=$th; fi
python - <<PY
import time, os, sys
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
T="$T"
ks = mf.KmerSet.from_fasta(T+"/s.bait.fa", 31)
def run(f1, f2, o1, o2, reps):
    best = 1e9
    for _ in range(reps):
        t0 = time.time(); kept, total = mf.filter_fastq_files(ks, f1, f2, o1, o2); best = min(best, time.time()-t0)
    return total * (2 if f2 else 1) / best / 1e6
print("pack threads %-8s SE gz %6.2f  PE gz %6.2f  PE plain %7.2f  SE plain %7.2f M reads/s" % ("$th", run(T+"/s_1.fq.gz", None, T+"/og_se.fq", None, 2),
      run(T+"/s_1.fq.gz", T+"/s_2.fq.gz", T+"/og_1.fq", T+"/og_2.fq", 2), run(T+"/s_1.fq", T+"/s_2.fq", T+"/o_1.fq", T+"/o_2.fq", 3), run(T+"/s_1.fq", None, T+"/o_se.fq", None, 3)), flush=True)
PY
done
rm -rf $T
