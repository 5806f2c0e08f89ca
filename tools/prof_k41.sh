#!/bin/bash
# rocprofv3 kernel stats of the resident pass at k = 41 (two-word keys): pipelined and with every kernel alone
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05}; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
Q="--k 41 --cpu-sample 0 --no-exhaustive --e2e-pairs 0 --e2e-full-reads 0 --no-group-a --no-live-traffic --real-gz-reads 0 --fv2-pairs 0 --plain-pairs 0 --k-sweep none"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 5 $Q > $OUT/k41_under_rocprof.json 2> /dev/null
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/k41_kernel_stats.csv; rm -rf $OUT/trace
MF_ENV_KNOBS=1 MF_PASS=serial timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 5 $Q > /dev/null 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/k41_kernel_stats_serial.csv; rm -rf $OUT/trace
head -5 $OUT/k41_kernel_stats.csv | cut -c1-60,150-330; head -5 $OUT/k41_kernel_stats_serial.csv | cut -c1-60,150-330
python3 -c "
import json; d=json.loads(open('$OUT/k41_under_rocprof.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['ms_per_step'], r['avg_kernel_ms'], r['frac'], r.get('whole_pass_frac'))"
