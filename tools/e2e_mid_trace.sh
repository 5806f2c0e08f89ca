#!/bin/bash
# one traced call of the device ingest path on a mid-size .gz: the slab cadence of the producer (MF_DEVINGEST_TRACE) and the stage summary
cd $GRAFT_REPO_ROOT; T=/tmp/e2et; mkdir -p $T
python tools/make_fastq.py $T/s --pairs ${1:-32000000} --mates 1 --block 2000000 > /dev/null
python tools/pgzip.py $T/s_1.fq $T/s.fq.gz --level 6
python - <<PY 2> $T/trace.txt
import time, sys, os
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
ks = mf.KmerSet.from_fasta("$T/s.bait.fa", 31)
mf.filter_fastq_files(ks, "$T/s.fq.gz", None, "$T/o.fq", None)
os.environ["MF_PIPE_TIMING"] = "1"; os.environ["MF_DEVINGEST_TRACE"] = "1"
t0 = time.time(); r = mf.filter_fastq_files(ks, "$T/s.fq.gz", None, "$T/o.fq", None); print("traced call %.3f s" % (time.time() - t0))
PY
grep -E "slab [0-9]+ decoded|mf device ingest" $T/trace.txt | awk 'NR % 4 == 1 || /mf device ingest/' | cut -c1-260 | tail -16
rm -rf $T
