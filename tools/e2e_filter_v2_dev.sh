#!/bin/bash
# filter_v2 end to end on the GPU box: plain and .gz inputs, device ingest path against the host pipeline (MF_QUAL_INGEST=host);
# process start to exit, outputs not pre-existing.  usage: [QUICK=1] [BIG=n] tools/e2e_filter_v2_dev.sh [pairs]
cd $GRAFT_REPO_ROOT; T=/tmp/fv2; mkdir -p $T
PAIRS=${1:-8000000}
python tools/make_fastq.py $T/s --pairs $PAIRS --block 2000000
python tools/pgzip.py $T/s_1.fq $T/s_1.fq.gz 6 2>/dev/null || gzip -6 -k $T/s_1.fq
python tools/pgzip.py $T/s_2.fq $T/s_2.fq.gz 6 2>/dev/null || gzip -6 -k $T/s_2.fq
ls -la $T
QUICK=${QUICK:-0}
if [ "$QUICK" = 0 ]; then g++ -O2 -pthread tools/write_bench.cpp -o $T/wb && for nt in 1 2 4 8; do $T/wb $T/wbf 4 $nt 0; done; for nt in 1 4 8; do $T/wb $T/wbf 4 $nt 2; done; fi
python - <<PY
import os, subprocess, time, hashlib
F="mitoflex_amd/filter/filter_v2"; T="$T"; pairs=$PAIRS; quick=$QUICK != 0
def md5(p):
    h=hashlib.md5()
    with open(p,'rb') as f:
        for b in iter(lambda: f.read(1<<24), b''): h.update(b)
    return h.hexdigest()
sums={}
def t(tag, args, reads, env=None, reps=3, check=None):
    global quick
    if quick and env and env.get("MF_QUAL_INGEST") == "host": return
    best=1e9
    e=dict(os.environ); e.update(env or {})
    for r in range(reps):
        for o in ('/o_1.fq', '/o_2.fq', '/o_se.fq'):
            if os.path.exists(T+o): os.unlink(T+o)
        if r == reps-1: e["MF_PIPE_TIMING"]="1"
        t0=time.time(); subprocess.check_call([F]+args, env=e); best=min(best,time.time()-t0)
    print(f"{tag:34s} {best:6.3f} s  {reads/best/1e6:7.2f} M reads/s  {reads*321/best/1e9:6.2f} GB/s of FASTQ", flush=True)
    if check:
        m=[md5(T+o) for o in (('/o_1.fq','/o_2.fq') if '-2' in args else ('/o_se.fq',))]
        if check in sums: print("    outputs equal to the first run's:", sums[check]==m, flush=True)
        else: sums[check]=m
H={"MF_QUAL_INGEST":"host"}
for sfx in ("", ".gz"):
    pe=["-1",T+"/s_1.fq"+sfx,"-2",T+"/s_2.fq"+sfx,"-3",T+"/o_1.fq","-4",T+"/o_2.fq"]
    se=["-1",T+"/s_1.fq"+sfx,"-3",T+"/o_se.fq","-q","60","-l","0.3"]
    kind="plain" if not sfx else ".gz"
    D={"MF_QUAL_INGEST":"device"}
    t(f"PE default {kind} device", pe, 2*pairs, D, check="pe")
    t(f"PE default {kind} host", pe, 2*pairs, H, reps=2, check="pe")
    t(f"PE dedup {kind} device", pe+["-d"], 2*pairs, D, check="ped")
    t(f"PE dedup {kind} host", pe+["-d"], 2*pairs, H, reps=2, check="ped")
    t(f"SE q60 l0.3 {kind} device", se, pairs, D, check="se")
    t(f"SE q60 l0.3 {kind} host", se, pairs, H, reps=2, check="se")
big = int(os.environ.get("BIG", "0"))
if big > 1:
    # the same files BIG times over, as gzip members one behind the other: a longer input without the minutes it takes to make one
    for m in ("1", "2"):
        with open(f"{T}/b_{m}.fq.gz", "wb") as o:
            for _ in range(big):
                with open(f"{T}/s_{m}.fq.gz", "rb") as i:
                    while True:
                        blk = i.read(1 << 24)
                        if not blk: break
                        o.write(blk)
    bpe = ["-1", T+"/b_1.fq.gz", "-2", T+"/b_2.fq.gz", "-3", T+"/o_1.fq", "-4", T+"/o_2.fq"]
    t(f"PE default .gz x{big} device", bpe, 2*pairs*big, reps=2, check="big")
    quick_saved, quick = quick, False
    t(f"PE default .gz x{big} host", bpe, 2*pairs*big, H, reps=1, check="big")
    quick = quick_saved
for c in (() if quick else (3, 10)):
    t(f"PE dedup .gz device, {c} consumers", ["-1",T+"/s_1.fq.gz","-2",T+"/s_2.fq.gz","-3",T+"/o_1.fq","-4",T+"/o_2.fq","-d"], 2*pairs, {"MF_INGEST_CONSUMERS":str(c)})
PY
wc -l $T/o_1.fq
# the bait filter on the same .gz files, three calls in one process (first call: streams, pool and staging buffers are made)
MF_PIPE_TIMING=1 python - <<PY 2>&1 | cut -c1-700
import time
from mitoflex_amd import mitofilter as mf
T="$T"
ks = mf.KmerSet.from_fasta(T + "/s.bait.fa", k=31)
for i in range(3):
    t0 = time.time(); kept, total = mf.filter_fastq_files(ks, T + "/s_1.fq.gz", None, T + "/b_1.fq", None); dt = time.time() - t0
    print(f"bait filter SE .gz call {i}: {dt:.3f} s  {total/dt/1e6:.1f} M reads/s  kept {kept} of {total}", flush=True)
for i in range(2):
    t0 = time.time(); kept, total = mf.filter_fastq_files(ks, T + "/s_1.fq.gz", T + "/s_2.fq.gz", T + "/b_1.fq", T + "/b_2.fq"); dt = time.time() - t0
    print(f"bait filter PE .gz call {i}: {dt:.3f} s  {2*total/dt/1e6:.1f} M reads/s  kept {kept} of {total}", flush=True)
PY
rm -rf $T
