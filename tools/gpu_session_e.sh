cd $GRAFT_REPO_ROOT; O=gpurun_out/r05g; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_devingest.py tests/test_filter_v2.py tests/test_gpu_parity.py tests/test_callsite.py tests/test_bim.py tests/test_gpu_protein.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_tail.txt
bash tools/gzdev_variants.sh > $O/gzdev_variants.txt 2>&1; cat $O/gzdev_variants.txt | cut -c1-200
bash tools/cold_calls.sh r05g > /dev/null 2>&1; grep -E "^==|wall " $O/cold_calls.log | cut -c1-200; grep -A45 "fastfilter bait l.fq.gz (cold process) rep 1" $O/cold_calls.log | grep -E "cold \+" | cut -c1-200
bash tools/exit_probe.sh > $O/exit_probe.log 2>&1; head -8 $O/exit_probe.log | cut -c1-200
