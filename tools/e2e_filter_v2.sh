#!/bin/bash
# filter_v2 drop-in end-to-end timing + kernel profile on the GPU box
cd $GRAFT_REPO_ROOT; T=/tmp/fv2; mkdir -p $T
PAIRS=${1:-8000000}
python tools/make_fastq.py $T/s --pairs $PAIRS
python - <<PY
import os, subprocess, time
F="mitoflex_amd/filter/filter_v2"; T="$T"; pairs=$PAIRS
def t(tag, args, reads):
    best=1e9
    for _ in range(2):
        for o in ('/o_1.fq', '/o_2.fq', '/o_se.fq'):
            if os.path.exists(T+o): os.unlink(T+o)          # (truncating a 2.6 GB file that sits in the page cache costs half a second of system time)
        t0=time.time(); subprocess.check_call([F]+args); best=min(best,time.time()-t0)
    print(f"{tag:26s} {best:6.2f} s  {reads/best/1e6:7.2f} M reads/s  {reads*321/best/1e9:6.2f} GB/s of FASTQ")
t("PE default", ["-1",T+"/s_1.fq","-2",T+"/s_2.fq","-3",T+"/o_1.fq","-4",T+"/o_2.fq"], 2*pairs)
t("PE dedup", ["-1",T+"/s_1.fq","-2",T+"/s_2.fq","-3",T+"/o_1.fq","-4",T+"/o_2.fq","-d"], 2*pairs)
t("SE q60 l0.3", ["-1",T+"/s_1.fq","-3",T+"/o_se.fq","-q","60","-l","0.3"], pairs)
PY
wc -l $T/o_1.fq $T/o_se.fq
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fv2prof -- $R/mitoflex_amd/filter/filter_v2 -1 $T/s_1.fq -2 $T/s_2.fq -3 $T/o_1.fq -4 $T/o_2.fq -d > /dev/null 2>&1
cat $(find $R/gpurun_out/fv2prof -name "*kernel_stats.csv" | head -1)
rm -rf $T $R/gpurun_out/fv2prof
