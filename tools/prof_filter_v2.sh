#!/bin/bash
# filter_v2 kernels: rocprofv3 kernel stats of one PE run with de-duplication (2 M-record batches of 150-base reads)
R=$GRAFT_REPO_ROOT; T=/tmp/fv2p; mkdir -p $T; cd $R
PAIRS=${1:-4000000}
python tools/make_fastq.py $T/s --pairs $PAIRS > /dev/null
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fv2prof -- $R/mitoflex_amd/filter/filter_v2 -1 $T/s_1.fq -2 $T/s_2.fq -3 $T/o_1.fq -4 $T/o_2.fq -d > /dev/null 2>&1
cat $(find $R/gpurun_out/fv2prof -name "*kernel_stats.csv" | head -1)
rm -rf $T $R/gpurun_out/fv2prof
