#!/bin/bash
# filter_v2 through the device ingest path: the reference's vectors, then a first timing
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_filter_v2.py -x -q -m gpu -k "bulk or random or device_path" 2>&1 | tail -40
