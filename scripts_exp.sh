#!/bin/bash
cd $GRAFT_REPO_ROOT
for s in 1 4 16 1000; do
  echo "stride $s"; MF_EVENT_STRIDE=$s python bench.py --steps 100 --warmup 10 --cpu-sample 0 --no-exhaustive | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['extra']['ms_screen_kernel'], d['extra']['ms_mark_kernel'], d['extra']['ms_exact_kernel'], d['extra']['ms_pass_events'])"
done
