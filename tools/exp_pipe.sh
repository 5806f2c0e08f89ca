#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
echo "--- k21 default"; SWEEP_K=21 python tools/sweep2.py 5000
echo "--- k21 no split pipe"; MF_SPLIT_PIPE=0 SWEEP_K=21 python tools/sweep2.py 5000
echo "--- k31 split pipe"; MF_PASS=split python tools/sweep2.py 5000 20000 100000
echo "--- k31 split serial"; MF_PASS=split MF_SPLIT_PIPE=0 python tools/sweep2.py 5000 20000 100000
echo "--- default"; python tools/sweep2.py 5000 20000 100000
