cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c; mkdir -p $O
(CHUNKS="256 64" timeout 600 bash tools/gzdev_run.sh 1000000) > $O/gzdev_run.log 2>&1; grep -E "PASS|FAIL|device link" $O/gzdev_run.log | cut -c1-260
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $O/pytest_gpu_tail.txt
bash tools/cold_calls.sh r05c > /dev/null 2>&1; grep -E "^==|wall|cold \+" $O/cold_calls.log | cut -c1-200
