#!/bin/bash
cd $GRAFT_REPO_ROOT
QUICK=1 ./tools/e2e_filter_v2_dev.sh 2>&1 | grep -v "^\[mf device ingest\] streams" | cut -c1-900
python bench.py 2>gpurun_out/bench16.err | tail -1 > gpurun_out/bench16.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench16.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'])
e=d['extra']['e2e_files']
print(json.dumps({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk in ('seconds','reads_per_s','roofline')}) for k,v in e.items()})[:1500])
PY
