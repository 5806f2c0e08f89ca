#!/bin/bash
# small transfers of the consumers / link step: kernel reads and writes of pinned memory against copies on the engine, by direction
cd $GRAFT_REPO_ROOT; T=/tmp/scab; mkdir -p $T
python tools/make_fastq.py $T/p --pairs 2000000 --block 2000000 > /dev/null; for m in 1 2; do python tools/pgzip.py $T/p_$m.fq $T/p_$m.fq.gz --level 6; done
python tools/make_fastq.py $T/q --pairs 16666667 --block 2000000 > /dev/null
run() {
python - "$@" <<PY
import time, os, sys
sys.path.insert(0, ".")
for kv in sys.argv[1:]:
    k, v = kv.split("=", 1); os.environ[k] = v
from mitoflex_amd import mitofilter as mf
T="$T"
def best(f, n=6):
    ts = []
    for _ in range(n):
        t0 = time.time(); f(); ts.append(time.time()-t0)
    return " ".join(f"{t:.3f}" for t in ts[1:])
a = best(lambda: mf.qualfilter_files(T+"/p_1.fq.gz", T+"/p_2.fq.gz", T+"/o1.fq", T+"/o2.fq", dedup=True))
b = best(lambda: mf.qualfilter_files(T+"/p_1.fq", T+"/p_2.fq", T+"/o1.fq", T+"/o2.fq", dedup=True))
ks = mf.KmerSet.from_fasta(T+"/q.bait.fa", 31)
c = best(lambda: mf.filter_fastq_files(ks, T+"/q_1.fq", T+"/q_2.fq", T+"/o1.fq", T+"/o2.fq"))
print(f"{' '.join(sys.argv[1:]) or 'kernels both ways (default)':40s} | filter_v2 -d .gz pair {a} | plain pair {b} | bait filter, plain 2 x 5.35 GB {c}", flush=True)
PY
}
run
run MF_SMALL_D2H_ON_ENGINE=1
run MF_SMALL_H2D_ON_ENGINE=1
run MF_SMALL_D2H_ON_ENGINE=1 MF_SMALL_H2D_ON_ENGINE=1
run
rm -rf $T
