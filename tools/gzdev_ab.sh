#!/bin/bash
cd $GRAFT_REPO_ROOT; T=/tmp/gzd; mkdir -p $T gpurun_out
python tools/make_fastq.py $T/s --pairs 1000000 > /dev/null
python tools/make_fastq.py $T/b --pairs 1000000 --qual binned > /dev/null
gzip -6 -c $T/s_1.fq > $T/s6.gz & gzip -6 -c $T/b_1.fq > $T/b6.gz & gzip -1 -c $T/s_1.fq > $T/s1.gz & gzip -9 -c $T/b_1.fq > $T/b9.gz & wait
for v in $@; do for f in s6 b6 s1 b9; do for ck in ${CHUNKS:-64}; do echo "== $v $f $ck"; timeout 120 tools/gzdev_check_$v $T/$f.gz $ck 24 2 | grep -v "decode kernel:"; done; done; done
