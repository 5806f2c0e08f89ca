#!/bin/bash
# The path as the reference calls it -- a process per call (utility/helper.py:78-86): what a COLD call costs, and where.
#   tools/cold_calls.sh [tag]   -> gpurun_out/<tag>/cold_calls.log
# (files: s = 500 k reads / 77 MB .gz, m = 4 M reads / 0.6 GB, l = 9 M reads / 1.4 GB: the last one is a "large" input -- CU-masked post streams, every decode stream)
# (1) tools/coldstart_probe: HIP start-up, allocation, pinned memory, streams; (2) `fastfilter bait` on a 500 k-read .gz and on a 4 M-read .gz,
# process start to exit, three times each, MF_PIPE_TIMING on the first; (3) filter_v2 on a 2 M-pair .gz pair with -d, device path and host pipeline.
cd $GRAFT_REPO_ROOT; TAG=${1:-r05}; O=gpurun_out/$TAG; mkdir -p $O; T=/tmp/cold; mkdir -p $T
L=$O/cold_calls.log; : > $L
python tools/make_fastq.py $T/s --pairs 500000 --mates 1 > /dev/null; gzip -6 -c $T/s_1.fq > $T/s.fq.gz
python tools/make_fastq.py $T/m --pairs 4000000 --mates 1 --block 2000000 > /dev/null; python tools/pgzip.py $T/m_1.fq $T/m.fq.gz --level 6
python tools/make_fastq.py $T/p --pairs 2000000 --block 2000000 > /dev/null; python tools/pgzip.py $T/p_1.fq $T/p_1.fq.gz --level 6; python tools/pgzip.py $T/p_2.fq $T/p_2.fq.gz --level 6
python tools/make_fastq.py $T/l --pairs 9000000 --mates 1 --block 3000000 > /dev/null; python tools/pgzip.py $T/l_1.fq $T/l.fq.gz --level 6; rm $T/l_1.fq
ls -l $T >> $L
echo "== coldstart_probe" >> $L
tools/coldstart_probe mitoflex_amd/libmitofilter_hip.so >> $L 2>&1
tools/coldstart_probe mitoflex_amd/libmitofilter_hip.so | grep -E "hipGetDeviceCount|first kernel|dlopen|done" >> $L 2>&1
wall() { python3 - "$@" <<'PY'
import subprocess, sys, time
t0 = time.time(); r = subprocess.run(sys.argv[1:], stdout=subprocess.PIPE); dt = time.time() - t0
print("   wall %.3f s rc %d stdout %s" % (dt, r.returncode, r.stdout.decode().strip()[:60]), flush=True)
PY
}
B=mitoflex_amd/assemble/fastfilter; F=mitoflex_amd/filter/filter_v2
for f in s m l; do
  for rep in 1 2 3; do
    echo "== fastfilter bait $f.fq.gz (cold process) rep $rep" >> $L
    rm -f $T/o.fq
    if [ $rep = 1 ]; then MF_PIPE_TIMING=1 MF_COLD_TRACE=1 wall $B bait --bait $T/$f.bait.fa --fq1 $T/$f.fq.gz --out1 $T/o.fq >> $L 2>&1; else wall $B bait --bait $T/$f.bait.fa --fq1 $T/$f.fq.gz --out1 $T/o.fq >> $L 2>&1; fi
  done
done
echo "== fastfilter bait s_1.fq PLAIN (cold process)" >> $L
for rep in 1 2; do rm -f $T/o.fq; MF_PIPE_TIMING=1 wall $B bait --bait $T/s.bait.fa --fq1 $T/s_1.fq --out1 $T/o.fq >> $L 2>&1; done
for ing in device host; do
  for rep in 1 2; do
    echo "== filter_v2 -d 2 M pairs .gz, MF_QUAL_INGEST=$ing rep $rep" >> $L
    rm -f $T/o_1.fq $T/o_2.fq
    MF_COLD_TRACE=$([ $rep = 1 ] && echo 1) MF_QUAL_INGEST=$ing MF_PIPE_TIMING=1 wall $F -1 $T/p_1.fq.gz -2 $T/p_2.fq.gz -3 $T/o_1.fq -4 $T/o_2.fq -d >> $L 2>&1
  done
done
rm -rf $T
grep -E "^==|wall|set-up" $L | cut -c1-400
