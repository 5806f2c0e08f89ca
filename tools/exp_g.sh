#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for k in 21 27 29 41; do python bench.py --k $k --steps 30 --cpu-sample 0 --no-exhaustive --e2e-pairs 0 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extra']; print('k=$k value %.3e ms/step %.4f screen %.4f mark %.4f finish0 %.4f items %d pass %d' % (d['value'], d['ms_per_step'], e['ms_screen_kernel'], e['ms_mark_kernel'], e['ms_finish_kernel_phase0'], e['work_items'], e['passed']))"; done
python tools/bait_fraction_sweep.py
