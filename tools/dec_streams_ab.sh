#!/bin/bash
# configs[4] (pgzip control, 5.15 GB): how many CU-masked decode streams does the chip need?  warm calls by MF_GZDEV_DEC_STREAMS; then the CLI
# as a cold process under the two stream sets and a masked set of four decode streams; then the plain text, warm.
cd $GRAFT_REPO_ROOT; T=/tmp/dsab; mkdir -p $T
python tools/make_fastq.py $T/s --pairs ${1:-33333334} --mates 1 --block 2000000 > /dev/null
python tools/pgzip.py $T/s_1.fq $T/s.fq.gz --level 6
for n in 10 3 4 6 10; do
python - $n <<PY
import time, os, sys
sys.path.insert(0, ".")
os.environ["MF_GZDEV_DEC_STREAMS"] = sys.argv[1]
from mitoflex_amd import mitofilter as mf
T="$T"
ks = mf.KmerSet.from_fasta(T+"/s.bait.fa", 31)
ts = []
for _ in range(4):
    t0 = time.time(); kept, total = mf.filter_fastq_files(ks, T+"/s.fq.gz", None, T+"/o.fq", None); ts.append(time.time()-t0)
st = mf.last_ingest_stats()
print(f"decode streams {sys.argv[1]:3s} kept {kept}/{total}  " + " ".join(f"{t:.3f}" for t in ts) + f" s   device in use at most {st['device_bytes_peak']/1e9:.2f} GB", flush=True)
PY
done
python - <<PY
import time, os, sys
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
T="$T"
ks = mf.KmerSet.from_fasta(T+"/s.bait.fa", 31)
ts = []
for _ in range(5):
    t0 = time.time(); kept, total = mf.filter_fastq_files(ks, T+"/s_1.fq", None, T+"/o.fq", None); ts.append(time.time()-t0)
print(f"plain text, warm: kept {kept}/{total}  " + " ".join(f"{t:.3f}" for t in ts), flush=True)
PY
wall() { python3 - "$@" <<'PY'
import subprocess, sys, time
t0 = time.time(); r = subprocess.run(sys.argv[1:], stdout=subprocess.PIPE); dt = time.time() - t0
print("   wall %.3f s rc %d stdout %s" % (dt, r.returncode, r.stdout.decode().strip()[:60]), flush=True)
PY
}
B=mitoflex_amd/assemble/fastfilter
for setting in "" "MF_GZDEV_LARGE_MB=256" "MF_GZDEV_LARGE_MB=256 MF_GZDEV_DEC_STREAMS=4" "MF_GZDEV_LARGE_MB=256 MF_GZDEV_DEC_STREAMS=3"; do
  for rep in 1 2 3; do
    echo "== [$setting] fastfilter bait configs[4] .gz (cold process) rep $rep"; rm -f $T/o.fq
    if [ $rep = 1 ]; then env $setting MF_PIPE_TIMING=1 MF_COLD_TRACE=1 bash -c "$(declare -f wall); wall $B bait --bait $T/s.bait.fa --fq1 $T/s.fq.gz --out1 $T/o.fq" 2>&1 | grep "wall\|HIP runtime answered\|files filtered\|first piece of text\|consumers done" | cut -c1-330
    else env $setting bash -c "$(declare -f wall); wall $B bait --bait $T/s.bait.fa --fq1 $T/s.fq.gz --out1 $T/o.fq"; fi
  done
done
rm -rf $T
