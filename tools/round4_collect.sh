#!/bin/bash
# round 4: the bench line shapes and the profiles of the round, on the GPU box -> gpurun_out/r04 (copied to profiles/r04)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_shape_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 --k 21 --e2e-pairs 0 --e2e-full-reads 0 --real-gz-reads 0 --fv2-pairs 0 --k-sweep none --no-group-a --cpu-sample 0 > $O/bench_shape_k21.json 2> $O/bench_k21.err
MF_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --reads 8000000 --no-exhaustive --cpu-sample 0 --e2e-pairs 200000 2> $O/bench_2ranks.err | grep "^{" > $O/bench_shape_2ranks_one_gpu.json
tail -c 400 $O/bench_shape_2ranks_one_gpu.json; echo
# rocprofv3 of the driver's bench command: kernel stats (pipelined and serial), FETCH_SIZE / WRITE_SIZE of the screen kernel
sed -e 's#gpurun_out/r03#gpurun_out/r04#' tools/prof_bench.sh > /tmp/prof_bench_r04.sh; sed -i 's/--no-group-a --no-live-traffic"/--no-group-a --no-live-traffic --real-gz-reads 0 --fv2-pairs 0 --k-sweep none"/' /tmp/prof_bench_r04.sh; bash /tmp/prof_bench_r04.sh > $O/prof_bench.log 2>&1; tail -4 $O/prof_bench.log
# rocprofv3 kernel stats of the device ingest path on configs[4]'s shape, and the decode launches on the time axis
tools/prof_devingest.sh 33333334 p6 > $O/devingest_kernel_stats.txt 2>&1; head -22 $O/devingest_kernel_stats.txt | cut -c1-160
# the decode kernel alone: counters, phases
bash tools/r4_pmc.sh > $O/gzdev_pmc.txt 2>&1; cat $O/gzdev_pmc.txt
python tools/bait_fraction_sweep.py 33333334 > $O/bait_fraction.txt 2>/dev/null; cat $O/bait_fraction.txt
MF_PIPE_TIMING=1 bash tools/e2e_small_gz.sh 500000 > $O/small_gz_500k.log 2>&1; grep -E "^call" $O/small_gz_500k.log
