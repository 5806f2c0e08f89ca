#!/bin/bash
# The decode kernel at other occupancies: LDS per wavefront sets how many fit a CU (16.4 KB: 9; 15.2 KB: 10; 12.4 KB: 12 -- then registers
# bound it), bought with shorter spans / smaller expansion rounds.  Same files, same box, byte for byte against zlib each.
cd $GRAFT_REPO_ROOT; T=/tmp/gzp; mkdir -p $T
python tools/make_fastq.py $T/s --pairs 12000000 --mates 1 --block 2000000 > /dev/null
python tools/pgzip.py $T/s_1.fq $T/s6.gz --level 6
head -c 1300000000 $T/s_1.fq | gzip -6 -c > $T/g6.gz
for v in "" _span896 _span640 _s768r192 _span1024 ""; do for f in s6 g6; do echo "== gzdev_check$v $f"; timeout 300 tools/gzdev_check$v $T/$f.gz 256 4 3 | grep -E "kernel .* ms:|PASS|FAIL"; done; done
rm -rf $T
