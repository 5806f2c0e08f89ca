#!/bin/bash
# End-to-end (files in -> files out) timing on the GPU box; host bound by construction.
cd $GRAFT_REPO_ROOT; T=/tmp/e2e; mkdir -p $T
PAIRS=${1:-4000000}
python tools/make_fastq.py $T/s --pairs $PAIRS
ls -la $T | head
( gzip -1 -c $T/s_1.fq > $T/s_1.fq.gz ) & ( gzip -1 -c $T/s_2.fq > $T/s_2.fq.gz ) & wait
python - <<PY
import time, os, sys
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
T="$T"; pairs=$PAIRS
ks = mf.KmerSet.from_fasta(T+"/s.bait.fa", 31)
def run(tag, f1, f2, o1, o2):
    best = 1e9
    for _ in range(4 if not f1.endswith('.gz') else 1):
        t0 = time.time(); kept, total = mf.filter_fastq_files(ks, f1, f2, o1, o2); dt = time.time()-t0; best = min(best, dt)
    n = total * (2 if f2 else 1)
    print(f"{tag:28s} kept {kept}/{total}  {best:6.2f} s  {n/best/1e6:7.2f} M reads/s  {n*150/best/1e9:6.2f} Gbp/s")
run("PE plain -> plain", T+"/s_1.fq", T+"/s_2.fq", T+"/o_1.fq", T+"/o_2.fq")
run("SE plain -> plain", T+"/s_1.fq", None, T+"/o_se.fq", None)
run("PE gz -> plain", T+"/s_1.fq.gz", T+"/s_2.fq.gz", T+"/og_1.fq", T+"/og_2.fq")
run("SE gz -> plain (configs[4])", T+"/s_1.fq.gz", None, T+"/og_se.fq", None)
assert open(T+"/o_1.fq","rb").read() == open(T+"/og_1.fq","rb").read()
# the same with four-bin quality strings (the default draws eighteen equiprobable values: twice the compressed size, mostly literals)
import subprocess
subprocess.check_call([sys.executable, "tools/make_fastq.py", T+"/b", "--pairs", str(pairs), "--qual", "binned"], stdout=subprocess.DEVNULL)
subprocess.check_call("gzip -1 -c %s/b_1.fq > %s/b_1.fq.gz" % (T, T), shell=True)
print("binned-quality .gz is %.2f x smaller than its text (uniform: %.2f x)" % (os.path.getsize(T+"/b_1.fq") / os.path.getsize(T+"/b_1.fq.gz"), os.path.getsize(T+"/s_1.fq") / os.path.getsize(T+"/s_1.fq.gz")))
run("SE gz -> plain, binned quals", T+"/b_1.fq.gz", None, T+"/ob_se.fq", None)
run("SE gz -> plain, binned quals", T+"/b_1.fq.gz", None, T+"/ob_se.fq", None)
print("cpu cores", os.cpu_count())
PY
# the CLI path (what shell_call would run)
time mitoflex_amd/assemble/fastfilter bait --bait $T/s.bait.fa --fq1 $T/s_1.fq --fq2 $T/s_2.fq --out1 $T/c_1.fq --out2 $T/c_2.fq
cmp $T/c_1.fq $T/o_1.fq && echo "CLI output identical"
rm -rf $T
