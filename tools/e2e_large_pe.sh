#!/bin/bash
# A 50 Gbp-shaped paired-end input -- two .gz files of PAIRS x 150-base reads each (default 62.5 M pairs: 2 x ~9.6 GB .gz, 2 x 20 GB of
# text; the box's scratch decides how far one can go) -- through the device ingest path: bounded device memory, output equal to the host
# pipeline's.      tools/e2e_large_pe.sh [pairs=62500000] [devices=1]
cd $GRAFT_REPO_ROOT; T=${TMPDIR_BIG:-/tmp}/e2el; mkdir -p $T gpurun_out
PAIRS=${1:-62500000}; NDEV=${2:-1}
df -h $T | tail -1
t0=$(date +%s)
for m in 1 2; do
  python tools/make_fastq.py $T/p --pairs $PAIRS --mates 2 --only-mate $m --block 2000000 > /dev/null || exit 1
  python tools/pgzip.py $T/p_$m.fq $T/p_$m.fq.gz --level 6 || exit 1
  ls -l $T/p_$m.fq $T/p_$m.fq.gz | awk '{print $5, $9}'
  rm -f $T/p_$m.fq
done
echo "generated and compressed in $(( $(date +%s) - t0 )) s"
MF_PIPE_TIMING=1 python - <<PY
import time, os, sys, hashlib
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
T="$T"; ndev=$NDEV
ks = mf.KmerSet.from_fasta(T+"/p.bait.fa", 31)
def md5(p):
    h = hashlib.md5()
    with open(p, "rb") as f:
        for b in iter(lambda: f.read(1 << 24), b""): h.update(b)
    return h.hexdigest()
def run(tag, reps):
    best = 1e9
    for _ in range(reps):
        t0 = time.time(); kept, total = mf.filter_fastq_files(ks, T+"/p_1.fq.gz", T+"/p_2.fq.gz", T+"/o1.fq", T+"/o2.fq", 1, mf.PAIR_EITHER, ndev); dt = time.time()-t0; best = min(best, dt)
    st = mf.last_ingest_stats()
    print(f"{tag:28s} kept {kept}/{total} pairs  {best:7.3f} s  {2*total/best/1e6:7.2f} M reads/s  md5 {md5(T+'/o1.fq')} {md5(T+'/o2.fq')}  path {st['path']} device memory in use at most {st['device_bytes_peak']/1e9:.1f} GB, buffers of the call {st['pool_bytes_peak']/1e9:.1f} GB, {st['input_bytes']/1e9:.1f} GB in, {st['text_bytes']/1e9:.1f} GB of text", flush=True)
    return kept, total, md5(T+"/o1.fq"), md5(T+"/o2.fq")
a = run("device ingest, %d device(s)" % ndev, 2)
os.environ["MF_INGEST"] = "host"
b = run("host pipeline", 1)
print("outputs equal:", a == b)
assert a == b
PY
rm -rf $T
