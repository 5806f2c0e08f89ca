#!/bin/bash
# configs[1]'s shape, plain files through the device path, under a few knobs
cd $GRAFT_REPO_ROOT; T=/tmp/pepk; mkdir -p $T
python tools/make_fastq.py $T/p --pairs ${1:-16666667} --block 2000000 > /dev/null
run() {
python - "$@" <<PY
import time, os, sys
sys.path.insert(0, ".")
for kv in sys.argv[1:]:
    k, v = kv.split("=", 1); os.environ[k] = v
from mitoflex_amd import mitofilter as mf
T="$T"
ks = mf.KmerSet.from_fasta(T+"/p.bait.fa", 31)
ts = []
for _ in range(6):
    t0 = time.time(); kept, total = mf.filter_fastq_files(ks, T+"/p_1.fq", T+"/p_2.fq", T+"/o1.fq", T+"/o2.fq"); ts.append(time.time()-t0)
st = mf.last_ingest_stats()
print(f"{' '.join(sys.argv[1:]) or 'default':56s} kept {kept}/{total}  " + " ".join(f"{t:.3f}" for t in ts) + f" s   device in use at most {st['device_bytes_peak']/1e9:.2f} GB", flush=True)
PY
}
run
run MF_INGEST_TEXT_BUFS=10
run MF_INGEST_TEXT_BUFS=10 MF_INGEST_CONSUMERS=6
run MF_INGEST_CONSUMERS=6
run MF_INGEST_SLAB_BYTES=134217728
run MF_INGEST_SLAB_BYTES=134217728 MF_INGEST_TEXT_BUFS=12
run MF_UPLOAD_THREADS=4
run MF_UPLOAD_THREADS=16
run
rm -rf $T
