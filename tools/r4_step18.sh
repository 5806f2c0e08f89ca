#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_devingest.py -x -q -m gpu -k "not configs4" 2>&1 | tail -3
QUICK=1 ./tools/e2e_filter_v2_dev.sh 2>&1 | cut -c1-1100 > $O/f_filter_v2_e2e_quick.log
grep -v "^\[mf" $O/f_filter_v2_e2e_quick.log | tail -24
grep "quality filter: wall" $O/f_filter_v2_e2e_quick.log | cut -c1-600 | tail -4
