#!/bin/bash
# configs[4] at its stated size: 33 333 334 single-end reads of 150 bases, gzipped (single member), 1 GPU.
#   tools/e2e_full_se_gz.sh [reads=33333334] [levels="1 6"]
cd $GRAFT_REPO_ROOT; T=/tmp/e2ef; mkdir -p $T
READS=${1:-33333334}; LEVELS=${2:-"1 6"}
t0=$(date +%s)
python tools/make_fastq.py $T/s --pairs $READS --mates 1 --block 2000000 > /dev/null
for l in $LEVELS; do ( gzip -$l -c $T/s_1.fq > $T/s.l$l.fq.gz ) & done; python tools/pgzip.py $T/s_1.fq $T/s.pgzip6.fq.gz --level 6; wait          # (GNU gzip: minutes, single-threaded; the same text through tools/pgzip.py as the control)
echo "generated and compressed in $(( $(date +%s) - t0 )) s"; ls -l $T | awk '{print $5, $9}'
python - <<PY
import time, os, sys, hashlib
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
T="$T"; levels="$LEVELS".split()
ks = mf.KmerSet.from_fasta(T+"/s.bait.fa", 31)
def md5(p):
    h = hashlib.md5()
    with open(p, "rb") as f:
        for b in iter(lambda: f.read(1 << 24), b""): h.update(b)
    return h.hexdigest()
def run(tag, f1, o1, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.time(); kept, total = mf.filter_fastq_files(ks, f1, None, o1, None); dt = time.time()-t0; best = min(best, dt)
    print(f"{tag:40s} kept {kept}/{total}  {best:6.3f} s  {total/best/1e6:7.2f} M reads/s  output md5 {md5(o1)}", flush=True)
    return kept, total
os.environ["MF_INGEST"] = "host"
ref = run("plain, host pipeline", T+"/s_1.fq", T+"/o_host.fq", reps=1)
for l in levels:
    os.environ.pop("MF_INGEST", None)
    r = run(f"SE gz -{l}, device ingest (configs[4])", T+f"/s.l{l}.fq.gz", T+f"/o_dev{l}.fq")
    assert r == ref
    os.environ["MF_INGEST"] = "host"
    run(f"SE gz -{l}, host pipeline", T+f"/s.l{l}.fq.gz", T+f"/o_hostgz{l}.fq", reps=1)
os.environ.pop("MF_INGEST", None)
r = run("SE pgzip -6 (control), device ingest", T+"/s.pgzip6.fq.gz", T+"/o_devp.fq")
assert r == ref
print("cpu cores", os.cpu_count(), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?")
PY
rm -rf $T
