#!/bin/bash
# configs[4] (pgzip control) by consumers x text buffers, after the CRC slots
cd $GRAFT_REPO_ROOT; T=/tmp/cnab; mkdir -p $T gpurun_out/r05
python tools/make_fastq.py $T/s --pairs ${1:-33333334} --mates 1 --block 2000000 > /dev/null
python tools/pgzip.py $T/s_1.fq $T/s.fq.gz --level 6; rm $T/s_1.fq
run() {
python - "$@" <<PY
import time, os, sys
sys.path.insert(0, ".")
for kv in sys.argv[1:]:
    k, v = kv.split("=", 1); os.environ[k] = v
from mitoflex_amd import mitofilter as mf
T="$T"
ks = mf.KmerSet.from_fasta(T+"/s.bait.fa", 31)
ts = []
for _ in range(5):
    t0 = time.time(); kept, total = mf.filter_fastq_files(ks, T+"/s.fq.gz", None, T+"/o.fq", None); ts.append(time.time()-t0)
st = mf.last_ingest_stats()
print(f"{' '.join(sys.argv[1:]) or 'default':60s} kept {kept}/{total}  " + " ".join(f"{t:.3f}" for t in ts) + f" s   device in use at most {st['device_bytes_peak']/1e9:.2f} GB", flush=True)
PY
}
run
run MF_INGEST_CONSUMERS=4 MF_INGEST_TEXT_BUFS=8
run MF_INGEST_CONSUMERS=6 MF_INGEST_TEXT_BUFS=10
run MF_INGEST_CONSUMERS=6 MF_INGEST_TEXT_BUFS=12
run MF_INGEST_CONSUMERS=8 MF_INGEST_TEXT_BUFS=14
run MF_INGEST_CONSUMERS=6 MF_INGEST_TEXT_BUFS=12 MF_GZDEV_SLAB_CHUNKS=320
run MF_INGEST_CONSUMERS=6 MF_INGEST_TEXT_BUFS=12 MF_INGEST_BUDGET_GB=32
run
MF_PIPE_TIMING=1 run MF_INGEST_CONSUMERS=6 MF_INGEST_TEXT_BUFS=12 2>&1 | grep "mf device ingest\] wall" | tail -1 | cut -c1-1600
rm -rf $T
