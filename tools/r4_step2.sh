#!/bin/bash
# round 4, step 2: every device ingest test incl. configs[4] at full size and the knob variants; the other GPU test files
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_devingest.py -m gpu -x -q -s > gpurun_out/r4s2_devingest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4s2_devingest.log
grep -E "configs\[4\]|passed|failed|rc " gpurun_out/r4s2_devingest.log | tail -12
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_devingest.py > gpurun_out/r4s2_rest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4s2_rest.log
tail -5 gpurun_out/r4s2_rest.log
