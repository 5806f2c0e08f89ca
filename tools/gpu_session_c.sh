cd $GRAFT_REPO_ROOT; O=gpurun_out/r05e; mkdir -p $O
# the decode kernel with the compacted search: byte for byte against zlib, per-phase cycles, rate with the chip full
(CHUNKS="256 64" timeout 600 bash tools/gzdev_run.sh 1000000) > $O/gzdev_run.log 2>&1; grep -E "PASS|FAIL|kernel .* ms:" $O/gzdev_run.log | cut -c1-200
timeout 900 bash tools/gzdev_phases.sh 12000000 > $O/gzdev_phases.log 2>&1; grep -E "cycles per chunk|kernel .* ms:|PASS|FAIL|^/tmp" $O/gzdev_phases.log | cut -c1-330
bash tools/exit_probe.sh > $O/exit_probe.log 2>&1; cat $O/exit_probe.log | cut -c1-220
timeout 900 python -m pytest tests/test_gpu_devingest.py -x -q -m gpu 2>&1 | tail -3
bash tools/cold_calls.sh r05e > /dev/null 2>&1; grep -E "^==|wall " $O/cold_calls.log | cut -c1-200; grep -A45 "fastfilter bait l.fq.gz (cold process) rep 1" $O/cold_calls.log | grep -E "cold \+|mf device ingest" | cut -c1-700
