#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; K=""; for a in "$@"; do case $a in --k) K="--k";; [0-9]*) [ -n "$K" ] && K="--k $a";; esac; done; env $(for a in "$@"; do case $a in *=*) echo $a;; esac; done) timeout 300 python bench.py --steps 20 --warmup 3 --cpu-sample 0 $K 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); e=d['extra']; print('value %.3e  screen %.4f ms  mark %.4f  exact %.4f ms  pass_ev %.4f  roofline %.3f  cand %d pass %d exh %.3e' % (d['value'], e['ms_screen_kernel'], e['ms_mark_kernel'], e['ms_exact_kernel'], e['ms_pass_events'], d['roofline']['frac'], e['candidates'], e['passed'], e.get('exhaustive_reads_per_s',0)))
    else: print(l.rstrip())
"; }
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run A=1
run MF_DEBUG_MARK=16
run A=1 --k 21
run A=1 --k 41
