#!/bin/bash
# same-box A/B of library variants: tools/ab.sh "old new" [bench args]   (alternates, three rounds)
R=$GRAFT_REPO_ROOT; cd $R; V="$1"; shift
for rep in 1 2 3; do for v in $V; do
  MITOFILTER_LIB=$R/mitoflex_amd/csrc/build/variants/libmitofilter_hip_$v.so python bench.py --steps 40 --warmup 5 --no-exhaustive --cpu-sample 0 --e2e-pairs 0 --no-live-traffic "$@" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extra']; print('$v rep $rep: ms/step %.4f screen %.4f finish0 %.4f single %.4f' % (d['ms_per_step'], e['ms_screen_kernel'], e['ms_finish_kernel_phase0'], e.get('ms_single_pass_latency', 0)))"
done; done
