#!/bin/bash
cd $GRAFT_REPO_ROOT; T=/tmp/gzp; mkdir -p $T gpurun_out
python tools/make_fastq.py $T/s --pairs 500000 --mates 1 > /dev/null
gzip -6 -c $T/s_1.fq > $T/g6.gz; gzip -1 -c $T/s_1.fq > $T/g1.gz
for f in g6 g1; do for ck in 64 32 128; do timeout 300 tools/gzdev_check_prof $T/$f.gz $ck 6 2; done; done
rm -rf $T
