#!/bin/bash
# round-2 experiment A: fused pass parity + timing against the split pass
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02a; mkdir -p $OUT; cd $R
B="python bench.py --steps 40 --warmup 5 --no-exhaustive"
( MF_PASS=split timeout 300 $B --cpu-sample 0 > $OUT/bench_split.json 2> $OUT/bench_split.err ) ; cat $OUT/bench_split.json | cut -c1-600
for w in 14 15 12; do
  ( MF_STREAM_WAVES=$w timeout 300 $B > $OUT/bench_fused_w$w.json 2> $OUT/bench_fused_w$w.err ); cut -c1-400 $OUT/bench_fused_w$w.json; grep -o '"extra".*' $OUT/bench_fused_w$w.json | cut -c1-700; tail -3 $OUT/bench_fused_w$w.err
done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/pytest_gpu.txt; cat $OUT/pytest_gpu.txt
