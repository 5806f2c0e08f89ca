#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02e; mkdir -p $OUT; cd $R
B="python bench.py --steps 40 --warmup 5 --no-exhaustive"
short() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
e=d['extra']; print(sys.argv[1].split('/')[-1], 'ms/step %.4f screen %.4f mark %.4f exact/finish %.4f pass %d cand %d match %s' % (d['ms_per_step'], e['ms_screen_kernel'], e['ms_mark_kernel'], e['ms_exact_kernel'], e['passed'], e['candidates'], e.get('sample_bits_match_oracle')))" $1; }
( timeout 300 $B > $OUT/default.json 2> $OUT/default.err ); short $OUT/default.json; tail -2 $OUT/default.err
( MF_PASS=serial timeout 300 $B --cpu-sample 0 > $OUT/serial.json 2> $OUT/serial.err ); short $OUT/serial.json
( MF_PASS=split timeout 300 $B --cpu-sample 0 > $OUT/split.json 2> $OUT/split.err ); short $OUT/split.json
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
