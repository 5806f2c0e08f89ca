#!/bin/bash
# parity of the pass kinds + same-box step time
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "filter_matches or edge or golden or synth or bench_bits or full" 2>&1 | tail -3
./tools/ab_env.sh MF_SCREEN_STREAMS "2 1"
