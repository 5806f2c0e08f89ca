#!/bin/bash
# final: the whole GPU suite and the bench line on the final code
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest_gpu_tail.txt
python bench.py > $O/bench_shape_default.json 2> $O/bench_default.err; tail -2 $O/bench_default.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04/bench_shape_default.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'])
q=d['extra']['filter_v2']; print(q['cli_process_start_to_exit'], q['library_call'])
e=d['extra']['e2e_files']; print(e['configs4_se_gz']['seconds'], e['configs4_se_gz']['roofline'], e['se_gz_reads_per_s'], e['pe_plain_reads_per_s'], e['real_compressors'])
PY
