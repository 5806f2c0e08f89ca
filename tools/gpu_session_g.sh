#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
python -m pytest tests/test_gpu_devingest.py -q -x -k "device_memory_follows" 2>&1 | tail -30 > $O/g_memtest.txt
bash tools/upload_ab.sh > $O/g_upload_ab.txt 2>&1
