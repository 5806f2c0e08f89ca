#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for cfg in "1024 1" "512 2" "256 4" "256 8" "512 4"; do set -- $cfg; echo "--- mark block $1 split $2"; MF_MARK_BLOCK=$1 MF_MARK_SPLIT=$2 SWEEP_K=21 python tools/sweep2.py 5000; MF_PASS=split MF_MARK_BLOCK=$1 MF_MARK_SPLIT=$2 python tools/sweep2.py 5000 100000; done
