#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02d; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3
WAVES="${WAVES:-12 14 16}" ./tools/exp_c.sh
