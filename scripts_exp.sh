#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; K=""; for a in "$@"; do case $a in --k) K="--k";; [0-9]*) [ -n "$K" ] && K="--k $a";; esac; done; env $(for a in "$@"; do case $a in *=*) echo $a;; esac; done) timeout 300 python bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-exhaustive $K 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); e=d['extra']; print('value %.3e  screen %.4f ms  mark %.4f  exact %.4f ms  pass_ev %.4f  roofline %.3f  cand %d pass %d' % (d['value'], e['ms_screen_kernel'], e['ms_mark_kernel'], e['ms_exact_kernel'], e['ms_pass_events'], d['roofline']['frac'], e['candidates'], e['passed']))
    else: print(l.rstrip())
"; }
run A=1
run MF_DEBUG_MARK=1
run MF_DEBUG_MARK=2
run MF_DEBUG_MARK=3
