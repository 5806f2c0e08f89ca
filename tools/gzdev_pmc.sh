#!/bin/bash
# PMC counters of gz_decode_kernel (counters only; one rocprofv3 pass per counter group)
R=$GRAFT_REPO_ROOT; T=/tmp/gzd; OUT=$R/gpurun_out/pmc_gz; mkdir -p $T $OUT
cd $R
python tools/make_fastq.py $T/s --pairs 1000000 > /dev/null
gzip -6 -c $T/s_1.fq > $T/s6.gz
BIN=${1:-tools/gzdev_check_v1}; CK=${2:-64}
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $OUT/sq_counters.txt
pmc() { name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $R/$BIN $T/s6.gz $CK 24 1 > /dev/null 2> $OUT/$name.err
  python3 - $(find $OUT/$name -name "*counter_collection.csv" | head -1) <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0].split('::')[-1][:40]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k, {c: '%.4g' % (sum(v)/len(v)) for c, v in d.items()}, 'launches=%d' % len(next(iter(d.values()))))
PY
}
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc sq2 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
pmc sq3 SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE
rm -rf $OUT/sq1 $OUT/sq2 $OUT/sq3
