#!/bin/bash
# Where the time between screen_kernel<1, 2> and the bare read loop of its own shape goes: the same counters for both
# (rocprofv3 --pmc, one small group of counters per pass; serial passes so that every screen launch has the device to itself).
#   tools/pmc_gap.sh  ->  gpurun_out/pmc_gap/summary.txt
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_gap; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/readbw $R/tools/readbw.hip 2> /dev/null
rocprofv3 --list-avail 2> /dev/null | grep -o "\b\(TCP\|TA\|TD\|TCC\|SQ\|SQC\|GRBM\|SPI\)_[A-Za-z0-9_]*" | sort -u > $OUT/avail.txt
READBW_SHAPE_ONLY=1 /tmp/readbw > $OUT/readbw_times.txt
sum() { python3 - "$1" "$2" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0].split('::')[-1][:40]
    if 'screen_kernel' in k or 'rd_screen_shape' in k:
        agg[k][r['Counter_Name']].append((int(r.get('Grid_Size', 0) or 0), float(r['Counter_Value'])))
for k, d in agg.items():
    for c, v in sorted(d.items()):
        # readbw launches two grids: keep the 224-workgroup launches (grid size 224 * 1024)
        vv = [x for g, x in v if 'rd_screen' not in k or g == 224 * 1024] or [x for g, x in v]
        print('%-8s %-28s %-34s %14.5g  (%d launches)' % (sys.argv[2], k[:28], c, sum(vv) / len(vv), len(vv)))
PY
}
pass() { name=$1; shift
  MF_ENV_KNOBS=1 MF_PASS=serial timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/s_$name -- python3 $R/bench.py --steps 3 --warmup 1 --prewarm-ms 0 --cpu-sample 0 --no-exhaustive --e2e-pairs 0 --e2e-full-reads 0 --no-group-a --no-live-traffic > /dev/null 2> $OUT/s_$name.err
  f=$(find $OUT/s_$name -name "*counter_collection.csv" | head -1); [ -n "$f" ] && sum "$f" screen
  READBW_SHAPE_ONLY=1 timeout 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/r_$name -- /tmp/readbw > /dev/null 2> $OUT/r_$name.err
  f=$(find $OUT/r_$name -name "*counter_collection.csv" | head -1); [ -n "$f" ] && sum "$f" readloop
  rm -rf $OUT/s_$name $OUT/r_$name
}
{
cat $OUT/readbw_times.txt
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
pass sq3 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
pass tcp2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pass ta TA_BUSY_avr TA_BUSY_max TA_TA_BUSY_sum TA_DATA_STALL_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
pass tcc1 TCC_REQ_sum TCC_READ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass tcc2 TCC_TAG_STALL_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_BUSY_avr TCC_BUSY_sum
pass tcc3 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_IO_CREDIT_STALL_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
} > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
