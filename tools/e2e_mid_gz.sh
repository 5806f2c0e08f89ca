#!/bin/bash
# a mid-size single-end .gz (default 16 M reads, one member written by tools/pgzip.py at level 6) through the device ingest path;
# MITOFILTER_LIBS="a.so b.so" alternates libraries in separate processes (same-box A/B)
cd $GRAFT_REPO_ROOT; T=/tmp/e2em; mkdir -p $T
python tools/make_fastq.py $T/s --pairs ${1:-16000000} --mates 1 --block 2000000 > /dev/null
python tools/pgzip.py $T/s_1.fq $T/s.fq.gz --level 6
for rep in 1 2; do for lib in ${MITOFILTER_LIBS:-default}; do
[ "$lib" = default ] && unset MITOFILTER_LIB || export MITOFILTER_LIB=$lib
python - <<PY
import time, sys, os
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
ks = mf.KmerSet.from_fasta("$T/s.bait.fa", 31)
best = 1e9
for i in range(3):
    t0 = time.time(); r = mf.filter_fastq_files(ks, "$T/s.fq.gz", None, "$T/o.fq", None); best = min(best, time.time() - t0)
print("$lib: %.3f s  %.1f M reads/s  kept %d/%d" % (best, r[1] / best / 1e6, r[0], r[1]), flush=True)
PY
done; done
rm -rf $T
