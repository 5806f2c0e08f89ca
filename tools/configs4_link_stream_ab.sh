#!/bin/bash
# configs[4], consecutive calls in one process: the link stream masked to the reserved CUs against a plain one (same box)
cd $GRAFT_REPO_ROOT; T=/tmp/c4; mkdir -p $T
python tools/make_fastq.py $T/f --pairs 33333334 --mates 1 --block 2000000
python tools/pgzip.py $T/f_1.fq $T/f_1.fq.gz --level 6; rm $T/f_1.fq; ls -la $T
for lm in 1 0 1 0; do
MF_GZDEV_LINK_MASK=$lm python - <<PY
import time, os
from mitoflex_amd import mitofilter as mf
T="$T"
ks = mf.KmerSet.from_fasta(T + "/f.bait.fa", k=31)
ts=[]
for i in range(5):
    t0 = time.time(); kept, total = mf.filter_fastq_files(ks, T + "/f_1.fq.gz", None, T + "/o.fq", None); ts.append(time.time() - t0)
print("link mask", os.environ["MF_GZDEV_LINK_MASK"], " ".join("%.3f" % t for t in ts), "kept", kept, "of", total, flush=True)
PY
done
rm -rf $T
