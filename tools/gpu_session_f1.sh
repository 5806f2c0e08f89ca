cd $GRAFT_REPO_ROOT; O=gpurun_out/r05h; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_devingest.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_tail.txt
bash tools/exit_probe.sh > $O/exit_probe.log 2>&1; cat $O/exit_probe.log | cut -c1-200
bash tools/chunk_size_probe.sh r05h > /dev/null 2>&1; cat $O/chunk_size_probe.txt | cut -c1-250
