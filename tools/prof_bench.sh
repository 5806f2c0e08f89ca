#!/bin/bash
# rocprofv3 --kernel-trace --stats of the driver's bench command (pipelined passes) and of the serial form of the pass, plus the
# FETCH_SIZE / WRITE_SIZE counters of the screen kernel: the numbers `roofline` in the bench line has to agree with
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/${1:-r05}; mkdir -p $OUT; rm -f $OUT/bench_traffic.txt; cd /tmp; export TMPDIR=/tmp
Q="--cpu-sample 0 --no-exhaustive --e2e-pairs 0 --e2e-full-reads 0 --no-group-a --no-live-traffic --real-gz-reads 0 --fv2-pairs 0 --plain-pairs 0 --k-sweep none"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 5 $Q > $OUT/bench_under_rocprof.json 2> /dev/null
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv; rm -rf $OUT/trace
MF_ENV_KNOBS=1 MF_PASS=serial timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 5 $Q > /dev/null 2>&1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats_serial.csv; rm -rf $OUT/trace
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc -- python3 $R/bench.py --steps 3 --warmup 1 $Q > /dev/null 2>&1
  python3 - $(find $OUT/pmc -name "*counter_collection.csv" | head -1) $c <<'PY' >> $OUT/bench_traffic.txt
import csv, sys
v = [float(r['Counter_Value']) for r in csv.DictReader(open(sys.argv[1])) if 'mf::screen_kernel<' in r['Kernel_Name'] and r['Counter_Name'] == sys.argv[2]]
print(sys.argv[2], 'screen_kernel mean per launch: %.6g KiB over %d launches' % (sum(v) / len(v), len(v)))
PY
  rm -rf $OUT/pmc
done
head -6 $OUT/bench_kernel_stats.csv | cut -c1-200; head -3 $OUT/bench_kernel_stats_serial.csv | cut -c1-200; cat $OUT/bench_traffic.txt
python3 -c "
import json; d=json.load(open('$OUT/bench_under_rocprof.json')); r=d['roofline']; print(d['ms_per_step'], r['avg_kernel_ms'], r['frac'], r['kernel_alone_frac'], r['traffic'])"
