#!/bin/bash
# round-4 collection, second part (after tools/round4_collect.sh): the whole GPU suite, filter_v2 end to end (device ingest against host pipeline, same box), the bench line with its filter_v2 leg
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest_gpu_tail.txt
BIG=4 ./tools/e2e_filter_v2_dev.sh 2>&1 | cut -c1-1100 > $O/f_filter_v2_e2e.log
grep -v "^\[mf" $O/f_filter_v2_e2e.log | tail -30
grep "put away" $O/f_filter_v2_e2e.log | sort | uniq -c | sort -rn | head -5
python bench.py > $O/bench_shape_default.json 2> $O/bench_default.err; tail -2 $O/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04/bench_shape_default.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'])
print(json.dumps(d['extra'].get('filter_v2'))[:1800])
print(d['extra']['e2e_files']['configs4_se_gz']['seconds'], d['extra']['e2e_files']['se_gz_reads_per_s'], d['extra']['e2e_files']['pe_plain_reads_per_s'])
PY
