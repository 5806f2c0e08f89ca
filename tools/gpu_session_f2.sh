cd $GRAFT_REPO_ROOT; O=gpurun_out/r05i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_devingest.py -x -q -m gpu 2>&1 | tail -25 | cut -c1-300 | tee $O/pytest_tail.txt
MF_PIPE_TIMING=1 bash tools/e2e_full_se_gz.sh 33333334 "6" > $O/configs4_gnu_gzip6.log 2>&1; grep -v "^\[mf" $O/configs4_gnu_gzip6.log | cut -c1-200; grep "^\[mf device ingest\] wall" $O/configs4_gnu_gzip6.log | cut -c1-1500 | tail -4
