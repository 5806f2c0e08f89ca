#!/bin/bash
# same-box A/B of one environment switch: tools/ab_env.sh VAR "v1 v2" [bench args]   (alternates, three rounds)
R=$GRAFT_REPO_ROOT; cd $R; VAR="$1"; V="$2"; shift; shift
for rep in 1 2 3; do for v in $V; do
  export $VAR=$v
  python bench.py --steps 40 --warmup 5 --no-exhaustive --cpu-sample 0 --e2e-pairs 0 --no-live-traffic "$@" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extra']; print('$VAR=$v rep $rep: ms/step %.4f sampled %.4f screen %.4f finish0 %.4f single %.4f ok %s' % (d['ms_per_step'], e['ms_per_step_sampled_loop'], e['ms_screen_kernel'], e['ms_finish_kernel_phase0'], e.get('ms_single_pass_latency', 0), e.get('sample_bits_match_oracle')))"
done; done
