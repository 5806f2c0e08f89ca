#!/bin/bash
cd $GRAFT_REPO_ROOT; T=/tmp/pgzp; mkdir -p $T
python tools/make_fastq.py $T/s --pairs 8000000 --mates 1 --block 2000000 > /dev/null
gzip -6 -c $T/s_1.fq > $T/g6.fq.gz &
python tools/pgzip.py $T/s_1.fq $T/p6.fq.gz --level 6 &
python tools/pgzip.py $T/s_1.fq $T/p6big.fq.gz --level 6 --slice-mb 256 &
wait
ls -l $T
python - <<PY
import sys, time, os
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
ks = mf.KmerSet.from_fasta("$T/s.bait.fa", 31)
for f in ("g6", "p6", "p6big", "g6", "p6"):
    t0 = time.time(); r = mf.filter_fastq_files(ks, "$T/%s.fq.gz" % f, None, "$T/o.fq", None); print(f, r, round(time.time() - t0, 3), flush=True)
PY
rm -rf $T
