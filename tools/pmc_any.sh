#!/bin/bash
# Per-kernel PMC counters (and, first, kernel-trace statistics) of tools/run_passes.py with the given arguments:
#   bash tools/pmc_any.sh <name> [run_passes.py arguments]      ->  gpurun_out/r06/pmc_<name>.txt
# Counters are collected in passes of their own (no trace domain beside --pmc).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; name=$1; shift
OUT=$R/gpurun_out/r06; mkdir -p $OUT; T=$OUT/pmc_$name.txt; : > $T
cd /tmp; export TMPDIR=/tmp
echo "# python3 tools/run_passes.py $*" >> $T
D=/tmp/pmc_$name.$$; rm -rf $D
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- python3 $R/tools/run_passes.py "$@" >> $T 2>$D.err
f=$(find $D/stats -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && { echo "# kernel stats (rocprofv3 --kernel-trace --stats)" >> $T; python3 - "$f" >> $T <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].split('(')[0].replace('void ', '')
    print(f"{n[:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:10.1f} us  min {float(r['MinNs'])/1e3:10.1f}  max {float(r['MaxNs'])/1e3:10.1f}  {r['Percentage']}%")
PY
}
pmc() {
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $D/p -- python3 $R/tools/run_passes.py "${ARGS[@]}" > /dev/null 2>>$D.err
  python3 - $(find $D/p -name "*counter_collection.csv" | head -1) >> $T <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:60]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    if any(x in k for x in ('screen', 'exact_kernel', 'mark_kernel', 'finish_kernel')) and 'build' not in k:
        print(k, {c: '%.5g' % (sum(v)/len(v)) for c, v in d.items()}, 'launches=%d' % len(next(iter(d.values()))))
PY
  rm -rf $D/p
}
ARGS=("$@")
[ -n "$STATS_ONLY" ] && { rm -rf $D $D.err; cat $T; exit 0; }          # STATS_ONLY=1: the kernel statistics alone (a PMC pass of a 33 M-read set takes minutes)
echo "# counters (per launch averages; rocprofv3 --pmc, one group a run)" >> $T
pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS
pmc FETCH_SIZE
pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pmc GRBM_GUI_ACTIVE
rm -rf $D $D.err
cat $T
