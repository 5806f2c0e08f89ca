#!/bin/bash
# rocprofv3 kernel stats of one device-ingest call (SE .gz)
cd $GRAFT_REPO_ROOT; T=/tmp/e2ep; mkdir -p $T gpurun_out
READS=${1:-8000000}; L=${2:-1}
python tools/make_fastq.py $T/s --pairs $READS --mates 1 --block 2000000 > /dev/null
if [ "${L#p}" != "$L" ]; then python tools/pgzip.py $T/s_1.fq $T/s.fq.gz --level ${L#p}; else gzip -$L -c $T/s_1.fq > $T/s.fq.gz; fi      # (p6: one member written on many cores, seconds instead of minutes)
cat > $T/run.py <<PY
import sys, time
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from mitoflex_amd import mitofilter as mf
ks = mf.KmerSet.from_fasta("$T/s.bait.fa", 31)
for i in range(2):
    t0 = time.time(); r = mf.filter_fastq_files(ks, "$T/s.fq.gz", None, "$T/o.fq", None); print(r, time.time() - t0, flush=True)
PY
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_dev -- python3 $T/run.py 2>&1 | grep -v "^W\|^E" | tail -5
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_dev -name "*kernel_stats.csv" | head -1); cat $f | head -30
# timeline of the decode kernels: start/end relative to the first
python3 - $(find $GRAFT_REPO_ROOT/gpurun_out/prof_dev -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
dec = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if 'gz_decode' in r['Kernel_Name']]
dec.sort()
half = len(dec) // 2
for part in (dec[:half], dec[half:]):
    t0 = part[0][0]
    busy = 0; cur_s, cur_e = part[0]
    for s, e in part[1:]:
        if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print("decode launches %d: first start -> last end %.1f ms, union busy %.1f ms, sum of durations %.1f ms" % (len(part), (max(e for s, e in part) - t0) / 1e6, busy / 1e6, sum(e - s for s, e in part) / 1e6))
    print("  per launch (start ms, dur ms):", " ".join("%.0f/%.0f" % ((s - t0) / 1e6, (e - s) / 1e6) for s, e in part[:24]))
PY
rm -rf $T $GRAFT_REPO_ROOT/gpurun_out/prof_dev
