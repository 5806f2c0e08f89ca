#!/bin/bash
# round 4, step 1: the streaming device ingest -- parity tests (without the full-size ones), then a mid-size and a small timing
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_devingest.py -m gpu -x -q -k "not configs4 and not knobs" > gpurun_out/r4s1_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/r4s1_pytest.log
tail -15 gpurun_out/r4s1_pytest.log
MF_PIPE_TIMING=1 timeout 600 bash tools/e2e_small_gz.sh 500000 > gpurun_out/r4s1_small.log 2>&1
tail -12 gpurun_out/r4s1_small.log
