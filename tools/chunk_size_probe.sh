#!/bin/bash
# The decoder's chunk size against the call's time, over the file sizes the default rule has to serve: 0.1 / 0.3 / 1 / 3 GB of .gz per mate
# x {64, 96, 128, 192, 256} KiB chunks, single-end and paired bait filter and the paired quality filter (-d), calls of a warm process, best of 3.
#   tools/chunk_size_probe.sh [tag]  -> gpurun_out/<tag>/chunk_size_probe.txt        (the rule in mf_devingest.cpp, GzStream::open, cites the committed copy)
R=$GRAFT_REPO_ROOT; TAG=${1:-r05}; O=$R/gpurun_out/$TAG; mkdir -p $O; T=/tmp/csp; mkdir -p $T
cd $R
python - <<PY > $O/chunk_size_probe.txt 2>&1
import os, subprocess, sys, time
sys.path.insert(0, ".")
T = "$T"
# pairs per size: 0.3 GB a mate is 2 M pairs of 150-base reads (gzip level 6 of this synthetic text: 154 bytes a record)
sizes = [("0.1 GB", 650_000), ("0.3 GB", 2_000_000), ("1 GB", 6_500_000), ("3 GB", 19_500_000)]
print("# tools/chunk_size_probe.sh: MF_GZDEV_CHUNK_BYTES against the call's seconds (best of 3, a warm process); '*' = the fastest of the row", flush=True)
from mitoflex_amd import mitofilter as mf
ks = None
def best(f, reps=3):
    b = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f(); b = min(b, time.perf_counter() - t0)
    return b
for name, pairs in sizes:
    for f in os.listdir(T):
        os.unlink(os.path.join(T, f))
    subprocess.check_call([sys.executable, "tools/make_fastq.py", T + "/q", "--pairs", str(pairs), "--block", "2000000"], stdout=subprocess.DEVNULL)
    for m in "12":
        subprocess.check_call([sys.executable, "tools/pgzip.py", f"{T}/q_{m}.fq", f"{T}/q_{m}.fq.gz", "--level", "6"], stdout=subprocess.DEVNULL)
        os.unlink(f"{T}/q_{m}.fq")
    if ks is None:
        ks = mf.KmerSet.from_fasta(T + "/q.bait.fa", k=31)
    gz = os.path.getsize(T + "/q_1.fq.gz")
    rows = {"bait filter SE": [], "bait filter PE": [], "quality filter PE -d": []}
    cks = [65536, 98304, 131072, 196608, 262144]
    for ck in cks:
        os.environ["MF_GZDEV_CHUNK_BYTES"] = str(ck)
        rows["bait filter SE"].append(best(lambda: mf.filter_fastq_files(ks, T + "/q_1.fq.gz", None, T + "/b1.fq", None)))
        rows["bait filter PE"].append(best(lambda: mf.filter_fastq_files(ks, T + "/q_1.fq.gz", T + "/q_2.fq.gz", T + "/b1.fq", T + "/b2.fq")))
        def q():
            for o in ("/o1.fq", "/o2.fq"):
                if os.path.exists(T + o): os.unlink(T + o)
            mf.qualfilter_files(T + "/q_1.fq.gz", T + "/q_2.fq.gz", T + "/o1.fq", T + "/o2.fq", dedup=True)
        rows["quality filter PE -d"].append(best(q, 2))
    os.environ.pop("MF_GZDEV_CHUNK_BYTES", None)
    dflt = {"bait filter SE": best(lambda: mf.filter_fastq_files(ks, T + "/q_1.fq.gz", None, T + "/b1.fq", None)),
            "bait filter PE": best(lambda: mf.filter_fastq_files(ks, T + "/q_1.fq.gz", T + "/q_2.fq.gz", T + "/b1.fq", T + "/b2.fq"))}
    print(f"== {name} a mate ({gz / 1e6:.0f} MB of .gz, {pairs} pairs); chunks of " + " / ".join(f"{c >> 10} KiB" for c in cks), flush=True)
    for what, v in rows.items():
        m = min(v)
        print(f"   {what:22s} " + "  ".join(f"{x:.4f}{'*' if x == m else ' '}" for x in v) + (f"   | the default rule: {dflt[what]:.4f}" if what in dflt else ""), flush=True)
PY
rm -rf $T
cat $O/chunk_size_probe.txt
