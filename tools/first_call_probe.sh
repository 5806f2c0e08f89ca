#!/bin/bash
# configs[4] (pgzip control): a process's FIRST call and the calls after it (library: CU-masked set), three processes; then the CLI cold, default
# (plain set below 8 GB) and with the masked set, with the time between "files filtered" and the process being gone
cd $GRAFT_REPO_ROOT; T=/tmp/fcp; mkdir -p $T
python tools/make_fastq.py $T/s --pairs ${1:-33333334} --mates 1 --block 2000000 > /dev/null
python tools/pgzip.py $T/s_1.fq $T/s.fq.gz --level 6; rm $T/s_1.fq
for rep in 1 2 3; do
python - <<PY
import time, os, sys
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
T="$T"
ks = mf.KmerSet.from_fasta(T+"/s.bait.fa", 31)
ts = []
for _ in range(4):
    t0 = time.time(); kept, total = mf.filter_fastq_files(ks, T+"/s.fq.gz", None, T+"/o.fq", None); ts.append(time.time()-t0)
st = mf.last_ingest_stats()
print(f"library, process $rep: kept {kept}/{total}  first call {ts[0]:.3f} s, then " + " ".join(f"{t:.3f}" for t in ts[1:]) + f"   device in use at most {st['device_bytes_peak']/1e9:.2f} GB", flush=True)
PY
done
run() { python3 - "$@" <<'PY'
import subprocess, sys, time, os, re
env = dict(os.environ, MF_COLD_TRACE="1")
t0 = time.time(); r = subprocess.run(sys.argv[1:], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env); dt = time.time() - t0
marks = re.findall(r"\[cold \+([0-9.]+)\] (.*)", r.stderr.decode())
def at(what):
    m = [x for x in marks if what in x[1]]; return m[0][0] if m else "?"
print("   caller saw %.3f s | HIP answered +%s | first text +%s | consumers done +%s | files filtered +%s | rc %d kept %s" % (dt, at("HIP runtime answered"), at("first piece of text"), at("consumers done"), at("files filtered"), r.returncode, r.stdout.decode().strip()), flush=True)
PY
}
B=mitoflex_amd/assemble/fastfilter
for setting in "" "MF_GZDEV_LARGE_MB=256"; do
  for rep in 1 2 3; do
    echo "== [$setting] fastfilter bait configs[4] .gz (cold process) rep $rep"; rm -f $T/o.fq
    env $setting bash -c "$(declare -f run); run $B bait --bait $T/s.bait.fa --fq1 $T/s.fq.gz --out1 $T/o.fq"
  done
done
rm -rf $T
