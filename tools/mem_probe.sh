#!/bin/bash
# device memory of a call by input size: pairs of .gz files, the call's own buffers against everything in use on the device (MF_DEVINGEST_TRACE)
cd $GRAFT_REPO_ROOT; T=/tmp/memp; mkdir -p $T
for pairs in ${1:-300000 1200000 4000000}; do
python tools/make_fastq.py $T/q --pairs $pairs > /dev/null
for m in 1 2; do python tools/pgzip.py $T/q_$m.fq $T/q_$m.fq.gz --level 6; done
python - <<PY 2>&1 | grep "^call\|device memory in use as\|mf device ingest" | sed -e 's/ | set-up.*survivors [0-9.]* / /' -e 's/ | first text.*//' | cut -c1-700
import time, os, sys
sys.path.insert(0, ".")
os.environ["MF_DEVINGEST_TRACE"] = "1"; os.environ["MF_PIPE_TIMING"] = "1"
from mitoflex_amd import mitofilter as mf
T="$T"
ks = mf.KmerSet.from_fasta(T+"/q.bait.fa", 31)
gz = os.path.getsize(T+"/q_1.fq.gz") + os.path.getsize(T+"/q_2.fq.gz")
for i in range(3):
    t0 = time.time(); r = mf.filter_fastq_files(ks, T+"/q_1.fq.gz", T+"/q_2.fq.gz", T+"/o1.fq", T+"/o2.fq"); dt = time.time() - t0
    st = mf.last_ingest_stats()
    print(f"call {i}: $pairs pairs, {gz/1e9:.3f} GB of .gz: {dt:.3f} s, buffers of this call at most {st['pool_bytes_peak']/1e9:.2f} GB, device in use at most {st['device_bytes_peak']/1e9:.2f} GB", flush=True)
    if i == 1: print("call: released", mf.release_cached() / 1e9, "GB", flush=True)
PY
done
rm -rf $T
