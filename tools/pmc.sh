#!/bin/bash
# ad-hoc: per-kernel PMC counters of one bench run (counters only: no --kernel-trace/--stats mixing with other traces)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
pass() {  # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$name -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-exhaustive > /dev/null 2> $R/gpurun_out/pmc_$name.err
  f=$(find $R/gpurun_out/pmc_$name -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name'].split('(')[0][-40:]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if 'screen' in k or 'exact' in k:
        print(k, {c: '%.4g' % (sum(v)/len(v)) for c, v in d.items()}, 'n=%d' % len(next(iter(d.values()))))
PY
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS
pass tcc1 FETCH_SIZE
pass tcc2 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pass grbm GRBM_GUI_ACTIVE
