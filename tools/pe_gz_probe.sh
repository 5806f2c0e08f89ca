#!/bin/bash
# a paired .gz input of 2 x 1.23 GB (8 M pairs): warm calls under a few knobs, and where the time goes (MF_PIPE_TIMING)
cd $GRAFT_REPO_ROOT; T=/tmp/pegz; mkdir -p $T
python tools/make_fastq.py $T/p --pairs ${1:-8000000} --block 2000000 > /dev/null; for m in 1 2; do python tools/pgzip.py $T/p_$m.fq $T/p_$m.fq.gz --level 6; done
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
for _ in range(5):
    t0 = time.time(); kept, total = mf.filter_fastq_files(ks, T+"/p_1.fq.gz", T+"/p_2.fq.gz", T+"/o1.fq", T+"/o2.fq"); ts.append(time.time()-t0)
st = mf.last_ingest_stats()
print(f"{' '.join(sys.argv[1:]) or 'default':48s} kept {kept}/{total}  " + " ".join(f"{t:.3f}" for t in ts) + f" s   chunks {st['chunks']} device in use at most {st['device_bytes_peak']/1e9:.2f} GB", flush=True)
PY
}
run
run MF_INGEST_BUDGET_GB=40
run MF_GZDEV_CHUNK_BYTES=65536
run MF_GZDEV_CHUNK_BYTES=196608
run MF_GZDEV_DEC_STREAMS=4
run MF_INGEST_TEXT_BUFS=8
run MF_INGEST_CONSUMERS=6
run
run MF_PIPE_TIMING=1 2>&1 | tail -3 | cut -c1-1800
rm -rf $T
