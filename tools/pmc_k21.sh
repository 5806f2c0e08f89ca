#!/bin/bash
# PMC counters of the candidate-bitmap pass at k = 21 (one stream, so that every kernel has the device to itself)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_k21; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp MF_ENV_KNOBS=1 MF_SPLIT_PIPE=0
pmc() { name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/bench.py --k 21 --steps 3 --warmup 1 --prewarm-ms 0 --cpu-sample 0 --no-exhaustive --e2e-pairs 0 --no-live-traffic > /dev/null 2> $OUT/$name.err
  python3 - $(find $OUT/$name -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0].split('::')[-1][:40]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if any(x in k for x in ('screen_kernel', 'exact_kernel', 'mark_kernel')):
        print(k, {c: '%.4g' % (sum(v)/len(v)) for c, v in d.items()}, 'launches=%d' % len(next(iter(d.values()))))
PY
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD
pmc sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
rm -rf $OUT/fetch $OUT/write $OUT/sq1 $OUT/sq2
