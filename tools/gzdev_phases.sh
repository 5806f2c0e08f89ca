#!/bin/bash
# per-phase cycle counts of the lane-parallel decode kernel (tools/gzdev_check built with -DGZ_PROFILE) and its rate with the chip full
# (make -C mitoflex_amd/csrc tools builds tools/gzdev_check and tools/gzdev_check_prof before this is sent to the GPU box)
cd $GRAFT_REPO_ROOT; T=/tmp/gzp; mkdir -p $T gpurun_out
python tools/make_fastq.py $T/s --pairs ${1:-12000000} --mates 1 --block 2000000 > /dev/null
python tools/pgzip.py $T/s_1.fq $T/s6.gz --level 6
head -c 1300000000 $T/s_1.fq | gzip -6 -c > $T/g6.gz &
head -c 1300000000 $T/s_1.fq | gzip -1 -c > $T/g1.gz &
wait
ls -l $T/*.gz
for f in g6 g1; do timeout 300 ${BIN:-tools/gzdev_check_prof} $T/$f.gz 256 4 2; done
timeout 600 ${BIN:-tools/gzdev_check_prof} $T/s6.gz 256 4 2
timeout 600 ${BIN:-tools/gzdev_check_prof} $T/s6.gz 512 4 2
rm -rf $T
