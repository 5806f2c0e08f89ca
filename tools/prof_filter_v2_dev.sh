#!/bin/bash
# rocprofv3 --kernel-trace --stats of the quality filter's drop-in on a 2 M-pair .gz pair with -d, device ingest path
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; T=/tmp/fvp; mkdir -p $T
python $R/tools/make_fastq.py $T/q --pairs 2000000 > /dev/null
for m in 1 2; do python $R/tools/pgzip.py $T/q_$m.fq $T/q_$m.fq.gz --level 6; done
cd /tmp; export TMPDIR=/tmp
timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fvt -- $R/mitoflex_amd/filter/filter_v2 -1 $T/q_1.fq.gz -2 $T/q_2.fq.gz -3 $T/o_1.fq -4 $T/o_2.fq -d > /dev/null 2>&1
cp $(find $O/fvt -name "*kernel_stats.csv" | head -1) $O/i_filter_v2_device_kernel_stats.csv; rm -rf $O/fvt $T
python3 - $O/i_filter_v2_device_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.reader(open(sys.argv[1])))[:16]:
    print(r[0].split('(')[0][-40:], r[1:5])
PY
