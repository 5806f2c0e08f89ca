#!/bin/bash
# configs[4] (pgzip control) and its plain text by number of consumer threads; first (cold) call of a process by number of decode streams
cd $GRAFT_REPO_ROOT; T=/tmp/cnab; mkdir -p $T
python tools/make_fastq.py $T/s --pairs ${1:-33333334} --mates 1 --block 2000000 > /dev/null
python tools/pgzip.py $T/s_1.fq $T/s.fq.gz --level 6
run() {
python - "$@" <<PY
import time, os, sys
sys.path.insert(0, ".")
for kv in sys.argv[1:]:
    k, v = kv.split("=", 1); os.environ[k] = v
from mitoflex_amd import mitofilter as mf
T="$T"
ks = mf.KmerSet.from_fasta(T+"/s.bait.fa", 31)
for tag, f in (("gz", T+"/s.fq.gz"), ("plain", T+"/s_1.fq")):
    ts = []
    for _ in range(4):
        t0 = time.time(); kept, total = mf.filter_fastq_files(ks, f, None, T+"/o.fq", None); ts.append(time.time()-t0)
    st = mf.last_ingest_stats()
    print(f"{' '.join(sys.argv[1:]) or 'default':40s} {tag:6s} kept {kept}/{total}  " + " ".join(f"{t:.3f}" for t in ts) + f" s   consumers {st['consumers']} device in use at most {st['device_bytes_peak']/1e9:.2f} GB", flush=True)
PY
}
run
run MF_INGEST_CONSUMERS=4
run MF_INGEST_CONSUMERS=6
run MF_INGEST_CONSUMERS=8
run MF_GZDEV_DEC_STREAMS=4
run MF_GZDEV_DEC_STREAMS=4 MF_INGEST_CONSUMERS=6
run MF_INGEST_CONSUMERS=6 MF_INGEST_TEXT_BUFS=10
run
MF_PIPE_TIMING=1 run MF_INGEST_CONSUMERS=6 2>&1 | grep "mf device ingest\] wall" | tail -2 | cut -c1-1500
rm -rf $T
