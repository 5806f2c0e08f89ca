#!/bin/bash
# counters of the finish kernels on bait-rich input (10 % bait reads, serial passes: every kernel alone on the device)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_finish; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp MF_ENV_KNOBS=1 MF_PASS=serial
PPM=${1:-100000}
pass() { name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $R/tools/bait_fraction_sweep.py 33333334 $PPM > /dev/null 2> $OUT/$name.err
  python3 - $(find $OUT/$name -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0].split('::')[-1][:44]
    if 'finish_kernel' in k or 'screen_kernel' in k:
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items()):
    for c, v in sorted(d.items()):
        print('%-40s %-34s %14.5g  (%d launches)' % (k, c, sum(v) / len(v), len(v)))
PY
  rm -rf $OUT/$name
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM
pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_WRITE_REQ_sum
pass tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_EA0_RDREQ_sum
pass grbm GRBM_GUI_ACTIVE
