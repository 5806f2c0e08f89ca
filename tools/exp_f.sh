#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02f; mkdir -p $OUT; cd $R
B="python bench.py --steps 40 --warmup 5 --no-exhaustive"
short() { python3 -c "
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
e=d['extra']; print(sys.argv[1].split('/')[-1], 'ms/step %.4f screen %.4f mark %.4f exact/finish %.4f pass %d cand %d match %s' % (d['ms_per_step'], e['ms_screen_kernel'], e['ms_mark_kernel'], e['ms_exact_kernel'], e['passed'], e['candidates'], e.get('sample_bits_match_oracle')))" $1; }
for v in ${VARIANTS:-b64}; do
  L=$R/mitoflex_amd/csrc/build/variants/libmitofilter_hip_$v.so
  ( MITOFILTER_LIB=$L MF_PASS=split timeout 300 $B > $OUT/split_$v.json 2> $OUT/split_$v.err ); short $OUT/split_$v.json; tail -2 $OUT/split_$v.err
  ( MITOFILTER_LIB=$L timeout 300 $B --cpu-sample 0 > $OUT/default_$v.json 2> $OUT/default_$v.err ); short $OUT/default_$v.json
done
( MF_PASS=split timeout 300 $B --cpu-sample 0 > $OUT/split.json 2> $OUT/split.err ); short $OUT/split.json
