#!/bin/bash
# ad-hoc tuning runs on the GPU box (not part of the product)
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 20 --warmup 3 --cpu-sample 0 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); e=d['extra']; print('value %.3e  screen %.4f ms  exact %.4f ms  pass_ev %.4f  roofline %.3f  cand %d pass %d exh %.3e' % (d['value'], e['ms_screen_kernel'], e['ms_exact_kernel'], e['ms_pass_events'], d['roofline']['frac'], e['candidates'], e['passed'], e.get('exhaustive_reads_per_s',0)))
    else: print(l.rstrip())
"; }
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
run A=1
run MF_DEBUG_SCREEN=1
