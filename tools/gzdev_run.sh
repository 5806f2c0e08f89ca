#!/bin/bash
# device DEFLATE decoder: correctness against zlib and kernel rate, on the GPU box
cd $GRAFT_REPO_ROOT; T=/tmp/gzd; mkdir -p $T gpurun_out
PAIRS=${1:-1000000}
python tools/make_fastq.py $T/s --pairs $PAIRS > /dev/null
python tools/make_fastq.py $T/b --pairs $PAIRS --qual binned > /dev/null
gzip -1 -c $T/s_1.fq > $T/s1.gz & gzip -6 -c $T/s_1.fq > $T/s6.gz & gzip -6 -c $T/b_1.fq > $T/b6.gz & gzip -9 -c $T/b_1.fq > $T/b9.gz & wait
head -c 3000000 $T/s_1.fq | python3 -c "import sys,zlib; d=sys.stdin.buffer.read(); c=zlib.compressobj(0,zlib.DEFLATED,31); sys.stdout.buffer.write(c.compress(d)+c.flush())" > $T/stored.gz
head -c 200 $T/s_1.fq | gzip -9 -c > $T/tiny.gz
ls -l $T/*.gz
for f in tiny stored s1 s6 b6 b9; do
  for ck in ${CHUNKS:-256}; do
    timeout 300 tools/gzdev_check $T/$f.gz $ck ${EXP:-8} 2
  done
done
rm -rf $T
