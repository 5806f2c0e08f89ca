#!/bin/bash
# round 3: the bench line shapes and the profiles of the device ingest path, on the GPU box
cd $GRAFT_REPO_ROOT; O=gpurun_out/r03; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_shape_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 --k 21 --e2e-pairs 0 --e2e-full-reads 0 --no-group-a --cpu-sample 0 > $O/bench_shape_k21.json 2> $O/bench_k21.err
MF_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --reads 8000000 --no-exhaustive --cpu-sample 0 --e2e-pairs 200000 2> $O/bench_2ranks.err | grep "^{" > $O/bench_shape_2ranks_one_gpu.json
tools/prof_devingest.sh 16000000 p6 > $O/devingest_kernel_stats.txt 2>&1
tail -c 600 $O/bench_shape_default.json; echo; tail -c 300 $O/bench_shape_k21.json; echo; tail -c 400 $O/bench_shape_2ranks_one_gpu.json
python tools/bait_fraction_sweep.py 33333334 > $O/bait_fraction.txt 2>/dev/null; cat $O/bait_fraction.txt
bash tools/e2e_filter_v2.sh 2>&1 | grep "M reads/s" > $O/filter_v2_e2e.txt; cat $O/filter_v2_e2e.txt
