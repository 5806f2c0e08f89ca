#!/bin/bash
# Round-end evidence run on the GPU box: gpu tests, bench line, rocprofv3 kernel stats, PMC traffic.
# Usage (from the build container):  gpurun --timeout 2400 -- ./tools/profile_round.sh r01
R=$GRAFT_REPO_ROOT; TAG=${1:-r02}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2>> $OUT/bench.err; cat $OUT/bench_driver_cmd.json
for k in 21 41; do timeout 300 python bench.py --k $k --steps 20 --cpu-sample 0 --e2e-pairs 0 --no-live-traffic > $OUT/bench_k$k.json 2>> $OUT/bench.err; done
MF_ENV_KNOBS=1 MF_PASS=split timeout 300 python bench.py --steps 30 --cpu-sample 0 --e2e-pairs 0 --no-live-traffic --no-exhaustive > $OUT/bench_split_pass.json 2>> $OUT/bench.err
MF_ENV_KNOBS=1 MF_SCREEN_STREAMS=1 timeout 300 python bench.py --steps 30 --cpu-sample 0 --e2e-pairs 0 --no-live-traffic --no-exhaustive > $OUT/bench_one_screen_stream.json 2>> $OUT/bench.err
MF_ENV_KNOBS=1 MF_SPLIT_PIPE=0 timeout 300 python bench.py --k 21 --steps 20 --cpu-sample 0 --e2e-pairs 0 --no-live-traffic --no-exhaustive > $OUT/bench_k21_one_stream.json 2>> $OUT/bench.err
MF_ENV_KNOBS=1 MF_PASS=serial timeout 300 python bench.py --steps 30 --cpu-sample 0 --e2e-pairs 0 --no-live-traffic --no-exhaustive > $OUT/bench_serial_pass.json 2>> $OUT/bench.err
timeout 300 python tools/bait_fraction_sweep.py > $OUT/bait_fraction.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-exhaustive --e2e-pairs 0 --no-live-traffic > $OUT/trace_bench.json 2> $OUT/trace.err
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null; cat $OUT/kernel_stats.csv
MF_ENV_KNOBS=1 MF_PASS=serial timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serial -- python3 $R/bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-exhaustive --e2e-pairs 0 --no-live-traffic > $OUT/trace_serial_bench.json 2> $OUT/trace_serial.err
cp $(find $OUT/trace_serial -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_serial.csv 2>/dev/null; head -4 $OUT/kernel_stats_serial.csv
pmc() { name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-exhaustive --e2e-pairs 0 --no-live-traffic > /dev/null 2> $OUT/pmc_$name.err
  python3 - $(find $OUT/pmc_$name -name "*counter_collection.csv" | head -1) <<'PY' | tee -a $OUT/pmc_summary.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0].split('::')[-1][:40]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if any(x in k for x in ('screen_kernel', 'exact_kernel', 'mark_kernel', 'finish_kernel')):
        print(k, {c: '%.5g' % (sum(v)/len(v)) for c, v in d.items()}, 'launches=%d' % len(next(iter(d.values()))))
PY
}
: > $OUT/pmc_summary.txt
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS
pmc grbm GRBM_GUI_ACTIVE
cd $R
timeout 600 ./tools/e2e_bench.sh 8000000 > $OUT/e2e_files.log 2>&1; tail -12 $OUT/e2e_files.log
timeout 600 ./tools/e2e_filter_v2.sh > $OUT/filter_v2_e2e.log 2>&1; tail -12 $OUT/filter_v2_e2e.log
timeout 600 python tools/bench_protein.py > $OUT/protein_bench.json 2> $OUT/protein_bench.err; cat $OUT/protein_bench.json
rm -rf $OUT/trace $OUT/trace_serial $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1 $OUT/pmc_sq2 $OUT/pmc_grbm
