#!/bin/bash
# does a process that used the device ingest path exit cleanly under rocprofv3?  (a backtrace from __cxa_finalize was seen)
cd $GRAFT_REPO_ROOT; T=/tmp/exitp; mkdir -p $T
python tools/make_fastq.py $T/s --pairs 200000 --mates 1 > /dev/null; gzip -1 -c $T/s_1.fq > $T/s.fq.gz
cat > $T/run.py <<PY
import sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from mitoflex_amd import mitofilter as mf
ks = mf.KmerSet.from_fasta("$T/s.bait.fa", 31)
print(mf.filter_fastq_files(ks, "$T/s.fq.gz", None, "$T/o.fq", None), flush=True)
PY
cd /tmp; export TMPDIR=/tmp
for v in ${VARIANTS:-"MF_KEEP_BUFFERS=1" "MF_KEEP_BUFFERS=0" "MF_INGEST=host"}; do
  echo "== $v, under rocprofv3 --kernel-trace"; env $v rocprofv3 --kernel-trace --stats --output-format csv -d $T/prof -- python3 $T/run.py 2>&1 | grep -v "^W2\|^E2\|^I2" | tail -8; echo "rc ${PIPESTATUS[0]}"
done
rm -rf $T
