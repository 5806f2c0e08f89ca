#!/bin/bash
# configs[4] (pgzip control) and the same text plain, uploads from the registered mapping vs through staging buffers (MF_UPLOAD_STAGED=1).
#   tools/upload_ab.sh [reads=33333334]
cd $GRAFT_REPO_ROOT; T=/tmp/upab; mkdir -p $T
READS=${1:-33333334}
python tools/make_fastq.py $T/s --pairs $READS --mates 1 --block 2000000 > /dev/null
python tools/pgzip.py $T/s_1.fq $T/s.fq.gz --level 6
ls -l $T | awk '{print $5, $9}'
for mode in registered staged registered staged; do
python - $mode <<PY
import time, os, sys
sys.path.insert(0, ".")
mode = sys.argv[1]
if mode == "staged": os.environ["MF_UPLOAD_STAGED"] = "1"
from mitoflex_amd import mitofilter as mf
T="$T"
ks = mf.KmerSet.from_fasta(T+"/s.bait.fa", 31)
for tag, f in (("gz", T+"/s.fq.gz"), ("plain", T+"/s_1.fq")):
    ts = []
    for _ in range(4):
        t0 = time.time(); kept, total = mf.filter_fastq_files(ks, f, None, T+"/o.fq", None); ts.append(time.time()-t0)
    st = mf.last_ingest_stats()
    print(f"{mode:10s} {tag:6s} kept {kept}/{total}  " + " ".join(f"{t:.3f}" for t in ts) + f" s   best {total/min(ts)/1e6:7.2f} M reads/s  device bytes peak {st['device_bytes_peak']/1e9:.2f} GB", flush=True)
PY
done
rm -rf $T
