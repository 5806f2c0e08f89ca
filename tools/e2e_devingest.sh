#!/bin/bash
# End-to-end timing of the device ingest path against the host pipeline (files in -> files out), on the GPU box.
#   tools/e2e_devingest.sh [pairs=4000000] [levels="1 6"]
cd $GRAFT_REPO_ROOT; T=/tmp/e2ed; mkdir -p $T
PAIRS=${1:-4000000}; LEVELS=${2:-"1 6"}
python tools/make_fastq.py $T/s --pairs $PAIRS > /dev/null
for l in $LEVELS; do ( gzip -$l -c $T/s_1.fq > $T/s_1.l$l.fq.gz ) & ( gzip -$l -c $T/s_2.fq > $T/s_2.l$l.fq.gz ) & done; wait
ls -l $T | awk '{print $5, $9}'
python - <<PY
import time, os, sys
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
T="$T"; levels="$LEVELS".split()
ks = mf.KmerSet.from_fasta(T+"/s.bait.fa", 31)
def run(tag, f1, f2, o1, o2, reps=2):
    best = 1e9
    for _ in range(reps):
        t0 = time.time(); kept, total = mf.filter_fastq_files(ks, f1, f2, o1, o2); dt = time.time()-t0; best = min(best, dt)
    n = total * (2 if f2 else 1)
    print(f"{tag:44s} kept {kept}/{total}  {best:6.3f} s  {n/best/1e6:7.2f} M reads/s", flush=True)
    return kept, total
for mode in ("device", "host"):
    os.environ["MF_INGEST"] = mode
    print("== MF_INGEST=" + mode, flush=True)
    run("SE plain", T+"/s_1.fq", None, T+f"/o_{mode}_se.fq", None)
    run("PE plain", T+"/s_1.fq", T+"/s_2.fq", T+f"/o_{mode}_1.fq", T+f"/o_{mode}_2.fq")
    for l in levels:
        run(f"SE gz -{l} (configs[4] shape)", T+f"/s_1.l{l}.fq.gz", None, T+f"/og{l}_{mode}_se.fq", None)
        run(f"PE gz -{l}", T+f"/s_1.l{l}.fq.gz", T+f"/s_2.l{l}.fq.gz", T+f"/og{l}_{mode}_1.fq", T+f"/og{l}_{mode}_2.fq")
same = lambda a, b: open(a, "rb").read() == open(b, "rb").read()
assert same(T+"/o_device_se.fq", T+"/o_host_se.fq") and same(T+"/o_device_1.fq", T+"/o_host_1.fq") and same(T+"/o_device_2.fq", T+"/o_host_2.fq")
for l in levels:
    assert same(T+f"/og{l}_device_se.fq", T+"/o_host_se.fq") and same(T+f"/og{l}_device_1.fq", T+"/o_host_1.fq") and same(T+f"/og{l}_device_2.fq", T+"/o_host_2.fq")
print("device and host outputs identical; cpu cores", os.cpu_count())
PY
rm -rf $T
