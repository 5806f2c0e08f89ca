#!/bin/bash
# configs[4] (pgzip control): MF_PIPE_TIMING of a warm call, and warm calls by slab size / slabs in flight
cd $GRAFT_REPO_ROOT; T=/tmp/c4t; mkdir -p $T gpurun_out/r05
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
for _ in range(4):
    t0 = time.time(); kept, total = mf.filter_fastq_files(ks, T+"/s.fq.gz", None, T+"/o.fq", None); ts.append(time.time()-t0)
st = mf.last_ingest_stats()
print(f"{' '.join(sys.argv[1:]) or 'default':60s} kept {kept}/{total}  " + " ".join(f"{t:.3f}" for t in ts) + f" s   device in use at most {st['device_bytes_peak']/1e9:.2f} GB", flush=True)
PY
}
MF_PIPE_TIMING=1 run 2>&1 | grep "mf device ingest\] wall\|default" | tail -2 | cut -c1-2200
run MF_GZDEV_SLAB_CHUNKS=320
run MF_GZDEV_SLAB_CHUNKS=320 MF_GZDEV_SLABS_IN_FLIGHT=12
run MF_GZDEV_SLAB_CHUNKS=640 MF_GZDEV_SLABS_IN_FLIGHT=8
run MF_GZDEV_SLAB_CHUNKS=1280 MF_GZDEV_SLABS_IN_FLIGHT=4
run MF_GZDEV_CHUNK_BYTES=131072
run MF_GZDEV_CHUNK_BYTES=262144
run MF_GZDEV_RESERVED_CUS=16
run MF_GZDEV_RESERVED_CUS=48
run
rm -rf $T
