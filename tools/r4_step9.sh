#!/bin/bash
# round 4, step 9: the whole GPU suite, then the driver's bench command
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r4s9_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4s9_pytest.log
tail -6 gpurun_out/r4s9_pytest.log
timeout 900 python bench.py > gpurun_out/r4s9_bench.json 2> gpurun_out/r4s9_bench.err
echo "bench rc $?"; tail -c 600 gpurun_out/r4s9_bench.err
python - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r4s9_bench.json") if l.startswith("{")][-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"]["whole_pass_frac"])
e = d["extra"]
print(json.dumps(e.get("k_sweep"), indent=1)[:900])
print(json.dumps(e.get("e2e_files"), indent=1)[:3500])
PY
