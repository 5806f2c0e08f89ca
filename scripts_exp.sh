#!/bin/bash
cd $GRAFT_REPO_ROOT; T=/tmp/e2e; mkdir -p $T
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_inflate.py tests/test_filter_v2.py -x -q -k "fastq or files or inflate or v2" 2>&1 | tail -3
python tools/make_fastq.py $T/s --pairs 4000000 > /dev/null
( gzip -6 -c $T/s_1.fq > $T/s_1.fq.gz ) & ( gzip -1 -c $T/s_2.fq > $T/s_2.fq.gz ) & wait
g++ -O3 -std=c++17 -I mitoflex_amd/csrc tests/native/inflate_check.cpp mitoflex_amd/csrc/build/mf_inflate.o mitoflex_amd/csrc/build/mf_pinflate.o -lz -lpthread -o $T/ic
for cfg in "16 2097152" "32 2097152" "32 1048576" "64 1048576" "48 2097152"; do echo "threads/chunk $cfg"; $T/ic $T/s_1.fq.gz --ptime 2 $cfg | tail -1; done
MF_PIPE_TIMING=1 python - <<PY
import time, os, sys
sys.path.insert(0, ".")
from mitoflex_amd import mitofilter as mf
T="$T"
ks = mf.KmerSet.from_fasta(T+"/s.bait.fa", 31)
for i in range(2):
    t0=time.time(); r=mf.filter_fastq_files(ks, T+"/s_1.fq.gz", T+"/s_2.fq.gz", T+"/o_1.fq", T+"/o_2.fq"); print(r, "PE gz %.2f s" % (time.time()-t0), file=sys.stderr)
for i in range(2):
    t0=time.time(); r=mf.filter_fastq_files(ks, T+"/s_1.fq.gz", None, T+"/o_1.fq", None); print(r, "SE gz %.2f s" % (time.time()-t0), file=sys.stderr)
PY
rm -rf $T
