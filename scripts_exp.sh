#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -k "fastq" 2>&1 | tail -3
./tools/e2e_bench.sh 4000000 2>&1 | tail -12
