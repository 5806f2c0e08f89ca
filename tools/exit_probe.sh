#!/bin/bash
# Where does the time between "files filtered" (the outputs are closed) and the process being gone go?  filter_v2 -d on a 2 M-pair .gz pair as a
# cold process: wall time seen by the caller against the time of the last mark, under settings that change how many hardware queues the
# process owns when it exits (every CU-masked stream is one; plain streams share GPU_MAX_HW_QUEUES of them).
cd $GRAFT_REPO_ROOT; T=/tmp/xp; mkdir -p $T
python tools/make_fastq.py $T/p --pairs 2000000 --block 2000000 > /dev/null; python tools/pgzip.py $T/p_1.fq $T/p_1.fq.gz --level 6; python tools/pgzip.py $T/p_2.fq $T/p_2.fq.gz --level 6
python tools/make_fastq.py $T/s --pairs 500000 --mates 1 > /dev/null; gzip -6 -c $T/s_1.fq > $T/s.fq.gz
F=mitoflex_amd/filter/filter_v2; B=mitoflex_amd/assemble/fastfilter
run() { python3 - "$@" <<'PY'
import subprocess, sys, time, os, re
env = dict(os.environ, MF_COLD_TRACE="1")
t0 = time.time(); r = subprocess.run(sys.argv[1:], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env); dt = time.time() - t0
marks = re.findall(r"\[cold \+([0-9.]+)\] (.*)", r.stderr.decode())
last = marks[-1] if marks else ("?", "?")
hip = [m for m in marks if "HIP runtime answered" in m[1]]
print("   caller saw %.3f s | last mark +%s (%s) | HIP answered +%s | rc %d" % (dt, last[0], last[1], hip[0][0] if hip else "?", r.returncode), flush=True)
PY
}
for rep in 1 2; do
for setting in "" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=12" "MF_GZDEV_LARGE_MB=0"; do
  echo "== [$setting] filter_v2 -d 2 M pairs"; rm -f $T/o_1.fq $T/o_2.fq
  env $setting bash -c "$(declare -f run); run $F -1 $T/p_1.fq.gz -2 $T/p_2.fq.gz -3 $T/o_1.fq -4 $T/o_2.fq -d"
  echo "== [$setting] fastfilter bait 500 k reads"; rm -f $T/o.fq
  env $setting bash -c "$(declare -f run); run $B bait --bait $T/s.bait.fa --fq1 $T/s.fq.gz --out1 $T/o.fq"
done; done
rm -rf $T
