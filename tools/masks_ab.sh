#!/bin/bash
# Are the CU-masked decode / post streams still worth their price (16 ms apiece to make, a quarter of a second of the process's exit) now that the
# link step is a few short kernels?  configs[4] (33.3 M reads, one 5.15 GB gzip member) and a paired 2 x 1.2 GB input, calls of a warm process,
# the masked set against the plain set in turn (MF_GZDEV_LARGE_MB = 0 / a huge value).  Also: plain FASTQ, device path against host pipeline,
# over file sizes (where should the default change over?).       tools/masks_ab.sh [tag] -> gpurun_out/<tag>/masks_ab.txt
R=$GRAFT_REPO_ROOT; TAG=${1:-r05}; O=$R/gpurun_out/$TAG; mkdir -p $O; T=/tmp/mab; mkdir -p $T; cd $R
python tools/make_fastq.py $T/f --pairs 33333334 --mates 1 --block 2000000 > /dev/null; python tools/pgzip.py $T/f_1.fq $T/f_1.fq.gz --level 6
python tools/make_fastq.py $T/p --pairs 8000000 --block 2000000 > /dev/null; for m in 1 2; do python tools/pgzip.py $T/p_$m.fq $T/p_$m.fq.gz --level 6; done
python - <<PY > $O/masks_ab.txt 2>&1
import os, sys, time
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
T = "$T"
ks = mf.KmerSet.from_fasta(T + "/f.bait.fa", k=31)
def call(f1, f2, n):
    t0 = time.perf_counter(); r = mf.filter_fastq_files(ks, f1, f2, T + "/o1.fq", T + "/o2.fq" if f2 else None); dt = time.perf_counter() - t0
    return dt, r
print("# tools/masks_ab.sh: seconds of a warm call; 'masked' = the CU-masked stream set (MF_GZDEV_LARGE_MB=0), 'plain' = plain streams only", flush=True)
for name, f1, f2, n in (("configs[4] SE 5.15 GB .gz", T + "/f_1.fq.gz", None, 33333334), ("PE 2 x 1.23 GB .gz", T + "/p_1.fq.gz", T + "/p_2.fq.gz", 16000000)):
    call(f1, f2, n)
    res = {"masked": [], "plain": []}
    for rep in range(4):
        for tag, v in (("masked", "0"), ("plain", "100000000")):
            os.environ["MF_GZDEV_LARGE_MB"] = v
            res[tag].append(call(f1, f2, n)[0])
    os.environ.pop("MF_GZDEV_LARGE_MB")
    print(f"{name}: masked " + " ".join(f"{x:.4f}" for x in res["masked"]) + " | plain " + " ".join(f"{x:.4f}" for x in res["plain"]) + f" | best {min(res['masked']):.4f} vs {min(res['plain']):.4f}", flush=True)
# plain FASTQ: where does the device path overtake the host pipeline?
import subprocess
print("# plain FASTQ, device path against host pipeline (best of 3 warm calls), by size", flush=True)
for pairs in (500_000, 2_000_000, 4_000_000, 8_000_000):
    for se in (True, False):
        f1, f2 = T + "/p_1.fq", (None if se else T + "/p_2.fq")
        # (a prefix of the 8 M-pair files: whole records of 321 bytes)
        g1, g2 = T + "/x_1.fq", (None if se else T + "/x_2.fq")
        for src, dst in ((f1, g1), (f2, g2)):
            if src: subprocess.check_call(["head", "-c", str(pairs * 321), src], stdout=open(dst, "wb"))
        out = {}
        for which in ("host", "device"):
            os.environ["MF_INGEST"] = which
            best = min(call(g1, g2, pairs)[0] for _ in range(3))
            out[which] = best
        os.environ.pop("MF_INGEST")
        print(f"   {'SE' if se else 'PE'} {pairs * 321 * (1 if se else 2) / 1e9:6.2f} GB of text: host pipeline {out['host']:.4f} s | device path {out['device']:.4f} s", flush=True)
PY
cat $O/masks_ab.txt
echo "== registered file mapping probe" >> $O/masks_ab.txt; PROBE_MMAP=$T/p_1.fq tools/coldstart_probe 2>&1 | grep -v "hipMalloc\|hipMemset\|hipFree \|hipHostMalloc\|hipHostFree" | tee -a $O/masks_ab.txt
rm -rf $T
