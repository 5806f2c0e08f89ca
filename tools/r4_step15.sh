#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_filter_v2.py -x -q -m gpu -k "bulk or random or device_path" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_devingest.py -x -q -m gpu -k "not configs4" 2>&1 | tail -5
QUICK=1 ./tools/e2e_filter_v2_dev.sh 2>&1 | grep -v "^\[mf device ingest\] streams" | cut -c1-1300
