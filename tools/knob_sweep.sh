#!/bin/bash
# Warm calls of the bait filter on one input shape under sets of environment knobs, a fresh process per set (the knobs are read when a process
# first uses the path).  Made this round's profiles/r05/g_*_ab.txt, g_*_knobs.txt, g_*_probe.txt files.
#   tools/knob_sweep.sh SHAPE ["KNOB=V KNOB=V" ...]        (no set, or the set "default": the defaults; MF_PIPE_TIMING=1 in a set prints the stage times)
#   SHAPE: c4gz    configs[4]: 33 333 334 single-end reads, one gzip member (tools/pgzip.py -6, 5.15 GB)
#          c4plain the same text, plain (10.7 GB)
#          pe_gz   configs[1]'s shape, 16 666 667 pairs, two .gz files        pe_plain   the same, plain
#          pe8_gz  8 M pairs, two .gz files of 1.23 GB
cd $GRAFT_REPO_ROOT; SHAPE=$1; shift; T=/tmp/knobs_$SHAPE; mkdir -p $T
case $SHAPE in
  c4gz|c4plain) python tools/make_fastq.py $T/s --pairs 33333334 --mates 1 --block 2000000 > /dev/null
                [ $SHAPE = c4gz ] && { python tools/pgzip.py $T/s_1.fq $T/s_1.fq.gz --level 6; rm $T/s_1.fq; }; F1=$T/s_1.fq; F2=; BAIT=$T/s.bait.fa ;;
  pe_gz|pe_plain|pe8_gz) P=16666667; [ $SHAPE = pe8_gz ] && P=8000000
                python tools/make_fastq.py $T/s --pairs $P --block 2000000 > /dev/null
                [ $SHAPE != pe_plain ] && { for m in 1 2; do python tools/pgzip.py $T/s_$m.fq $T/s_$m.fq.gz --level 6 & done; wait; rm $T/s_1.fq $T/s_2.fq; }; F1=$T/s_1.fq; F2=$T/s_2.fq; BAIT=$T/s.bait.fa ;;
  *) echo "unknown shape $SHAPE"; exit 1 ;;
esac
case $SHAPE in *gz) F1=$F1.gz; [ -n "$F2" ] && F2=$F2.gz ;; esac
[ $# = 0 ] && set -- default
for set in "$@"; do
python - "$SHAPE" "$F1" "$F2" "$BAIT" $set <<'PY' 2>&1 | grep "^\[mf device ingest\] wall\|^[a-z0-9_]* *|" | cut -c1-2400
import time, os, sys
sys.path.insert(0, ".")
shape, f1, f2, bait = sys.argv[1:5]; knobs = [k for k in sys.argv[5:] if k != "default"]
for kv in knobs:
    k, v = kv.split("=", 1); os.environ[k] = v
from mitoflex_amd import mitofilter as mf
ks = mf.KmerSet.from_fasta(bait, 31)
ts = []
for _ in range(5):
    t0 = time.time(); kept, total = mf.filter_fastq_files(ks, f1, f2 or None, "/tmp/knobs_" + shape + "/o1.fq", "/tmp/knobs_" + shape + "/o2.fq" if f2 else None); ts.append(time.time() - t0)
st = mf.last_ingest_stats()
print(f"{shape:8s}| {' '.join(knobs) or 'default':64s} | kept {kept}/{total} | first call {ts[0]:.3f} s, then " + " ".join(f"{t:.3f}" for t in ts[1:]) + f" | device in use at most {st['device_bytes_peak'] / 1e9:.2f} GB", flush=True)
PY
done
rm -rf $T
