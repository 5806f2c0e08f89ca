#!/bin/bash
cd $GRAFT_REPO_ROOT; T=/tmp/gzp; mkdir -p $T gpurun_out
python tools/make_fastq.py $T/s --pairs ${1:-12000000} --mates 1 --block 2000000 > /dev/null
python tools/pgzip.py $T/s_1.fq $T/s6.gz --level 6
head -c 1300000000 $T/s_1.fq | gzip -6 -c > $T/g6.gz
for b in $BINS; do echo "== $b"; timeout 300 $b $T/g6.gz 256 4 2; timeout 600 $b $T/s6.gz 256 4 2; done
rm -rf $T
