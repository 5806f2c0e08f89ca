#!/bin/bash
# rocprofv3 --kernel-trace --stats of the decode kernel alone with the chip full: tools/gzdev_check on a 1.85 GB gzip member
# (12 M reads, 3.85 GB of text, 7 069 chunks of 256 KiB in ONE launch), byte for byte against zlib
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05}; mkdir -p $O; T=/tmp/gzp; mkdir -p $T
python $R/tools/make_fastq.py $T/s --pairs 12000000 --mates 1 --block 2000000 > /dev/null
python $R/tools/pgzip.py $T/s_1.fq $T/s6.gz --level 6; rm $T/s_1.fq
cd /tmp; export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/gzt -- $R/tools/gzdev_check $T/s6.gz 256 6 2 > $O/h_gzdev_check_under_rocprof.log 2>&1
cp $(find $O/gzt -name "*kernel_stats.csv" | head -1) $O/h_gzdev_check_kernel_stats.csv; rm -rf $O/gzt $T
tail -4 $O/h_gzdev_check_under_rocprof.log; cut -c1-60,400-520 $O/h_gzdev_check_kernel_stats.csv | head -5
