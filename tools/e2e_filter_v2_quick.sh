#!/bin/bash
# filter_v2: PE default / PE dedup timing with the stage breakdown (no profiler)
cd $GRAFT_REPO_ROOT; T=/tmp/fv2q; mkdir -p $T
PAIRS=${1:-8000000}
python tools/make_fastq.py $T/s --pairs $PAIRS > /dev/null
F=mitoflex_amd/filter/filter_v2
for rep in 1 2; do
  for mode in "" "-d"; do
    rm -f $T/o_1.fq $T/o_2.fq; echo "filter_v2 PE $mode"; time MF_PIPE_TIMING=1 $F -1 $T/s_1.fq -2 $T/s_2.fq -3 $T/o_1.fq -4 $T/o_2.fq $mode
  done
done
rm -rf $T
