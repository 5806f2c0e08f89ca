#!/bin/bash
# SQ counters of the decode kernel alone (tools/gzdev_check on one gzip -6 file): where a wavefront's cycles go
cd $GRAFT_REPO_ROOT; T=/tmp/gzp; mkdir -p $T gpurun_out/pmc
python tools/make_fastq.py $T/s --pairs 4000000 --mates 1 --block 2000000 > /dev/null
gzip -6 -c $T/s_1.fq > $T/g6.gz
cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  rm -rf /tmp/pmc_out
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_out -- $GRAFT_REPO_ROOT/tools/gzdev_check $T/g6.gz 256 4 1 > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("/tmp/pmc_out/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gz_decode" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in sorted(acc): print(f"{k:28s} {acc[k]/max(1,n[k]):16.0f}  (launches {n[k]})")
PY
done
rm -rf $T
