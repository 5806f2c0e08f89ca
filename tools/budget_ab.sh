#!/bin/bash
# configs[4] (pgzip control, 5.15 GB) by device-memory budget: seconds of a warm call and everything in use on the device
cd $GRAFT_REPO_ROOT; T=/tmp/bdab; mkdir -p $T
python tools/make_fastq.py $T/s --pairs ${1:-33333334} --mates 1 --block 2000000 > /dev/null
python tools/pgzip.py $T/s_1.fq $T/s.fq.gz --level 6; rm $T/s_1.fq
for b in default 12 16 20 24 32 default; do
python - $b <<PY
import time, os, sys
sys.path.insert(0, ".")
b = sys.argv[1]
if b != "default": os.environ["MF_INGEST_BUDGET_GB"] = b
from mitoflex_amd import mitofilter as mf
T="$T"
ks = mf.KmerSet.from_fasta(T+"/s.bait.fa", 31)
ts = []
for _ in range(4):
    t0 = time.time(); kept, total = mf.filter_fastq_files(ks, T+"/s.fq.gz", None, T+"/o.fq", None); ts.append(time.time()-t0)
st = mf.last_ingest_stats()
print(f"budget {b:8s} kept {kept}/{total}  " + " ".join(f"{t:.3f}" for t in ts) + f" s   buffers {st['pool_bytes_peak']/1e9:.2f} GB, device in use at most {st['device_bytes_peak']/1e9:.2f} GB", flush=True)
PY
done
rm -rf $T
