#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for v in base prio; do export MITOFILTER_LIB=$R/mitoflex_amd/csrc/build/variants/libmitofilter_hip_$v.so; echo "--- $v"; SWEEP_K=21 python tools/sweep2.py 5000; MF_PASS=split python tools/sweep2.py 5000 100000; done
