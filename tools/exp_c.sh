#!/bin/bash
# round-2 experiment C: streaming / finisher wave split sweep
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02c; mkdir -p $OUT; cd $R
B="python bench.py --steps 40 --warmup 5 --no-exhaustive --cpu-sample 0"
short() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
e=d['extra']; print(sys.argv[1].split('/')[-1], 'ms/step %.4f kernel %.4f pass %d cand %d' % (d['ms_per_step'], e['ms_screen_kernel'], e['passed'], e['candidates']))" $1; }
for w in ${WAVES:-8 9 10 11 12}; do
  ( MF_STREAM_WAVES=$w MF_FUSED_DEBUG=1 timeout 300 $B > $OUT/fused_w$w.json 2> $OUT/fused_w$w.err ); short $OUT/fused_w$w.json; grep "mf fused" $OUT/fused_w$w.err | tail -2
done
