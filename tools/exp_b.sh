#!/bin/bash
# round-2 experiment B: fused pass with sample-mode finishing; stream-only timing; per-role timestamps
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02b; mkdir -p $OUT; cd $R
B="python bench.py --steps 40 --warmup 5 --no-exhaustive"
short() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
e=d['extra']; print(sys.argv[1].split('/')[-1], 'ms/step %.4f kernel %.4f pass %d cand %d match %s' % (d['ms_per_step'], e['ms_screen_kernel'], e['passed'], e['candidates'], e.get('sample_bits_match_oracle')))" $1; }
for w in 12 13 14; do
  ( MF_STREAM_WAVES=$w MF_FUSED_DEBUG=1 timeout 300 $B > $OUT/fused_w$w.json 2> $OUT/fused_w$w.err ); short $OUT/fused_w$w.json; grep "mf fused" $OUT/fused_w$w.err | tail -2
done
for w in 12 14 15; do
  ( MF_STREAM_WAVES=$w MF_FUSED_DROP=1 MF_FUSED_DEBUG=1 timeout 300 $B --cpu-sample 0 > $OUT/drop_w$w.json 2> $OUT/drop_w$w.err ); short $OUT/drop_w$w.json; grep "mf fused" $OUT/drop_w$w.err | tail -1
done
( MF_PASS=split timeout 300 $B --cpu-sample 0 > $OUT/split.json 2> $OUT/split.err ); short $OUT/split.json
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
