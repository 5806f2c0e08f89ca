#!/bin/bash
# Round-end evidence on the GPU box -> gpurun_out/<tag>/ (copied to profiles/<tag>/ afterwards):
#   the whole GPU test suite; the bench line's shapes (default run, k = 21, two ranks on one GPU); rocprofv3 kernel stats of the driver's bench command
#   (pipelined and serial) + the FETCH_SIZE / WRITE_SIZE passes; kernel stats of the device ingest path on configs[4]'s shape; the decode kernel alone
#   under rocprofv3; the cold calls through the CLIs.
#   gpurun --timeout 3600 -- tools/collect_round.sh r05
cd $GRAFT_REPO_ROOT; TAG=${1:-r05}; O=gpurun_out/$TAG; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/pytest_gpu_tail.txt
python bench.py > $O/bench_shape_default.json 2> $O/bench_default.err; tail -2 $O/bench_default.err
python bench.py --steps 20 --warmup 5 --k 21 --e2e-pairs 0 --e2e-full-reads 0 --real-gz-reads 0 --fv2-pairs 0 --plain-pairs 0 --k-sweep none --no-group-a --cpu-sample 0 > $O/bench_shape_k21.json 2> $O/bench_k21.err
MF_BENCH_SHARE_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --reads 8000000 --no-exhaustive --cpu-sample 0 --e2e-pairs 200000 2> $O/bench_2ranks.err | grep "^{" > $O/bench_shape_2ranks_one_gpu.json
bash tools/prof_bench.sh $TAG > $O/prof_bench.log 2>&1; tail -4 $O/prof_bench.log
tools/prof_devingest.sh 33333334 p6 > $O/devingest_kernel_stats.txt 2>&1; head -14 $O/devingest_kernel_stats.txt | cut -c1-160
bash tools/prof_gzdev_check.sh $TAG > /dev/null 2>&1; tail -3 $O/h_gzdev_check_under_rocprof.log; head -3 $O/h_gzdev_check_kernel_stats.csv | cut -c1-60,400-520
bash tools/cold_calls.sh $TAG > /dev/null 2>&1; grep -E "^==|wall " $O/cold_calls.log | cut -c1-160
# round 6: the bait-size axis with the screened == exhaustive check per size, and rocprofv3 kernel statistics of the 350 kbp leg
SWEEP_CHECK=1 timeout 600 python tools/bait_sweep.py > $O/bait_sweep_final.txt 2>&1; cat $O/bait_sweep_final.txt | cut -c1-200
STATS_ONLY=1 bash tools/pmc_any.sh s2_350k_stats --bait-size 350000 > /dev/null 2>&1; cp gpurun_out/r06/pmc_s2_350k_stats.txt $O/ 2>/dev/null
python - <<PY
import json
d = json.load(open("$O/bench_shape_default.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["kernel_alone_frac"], d["roofline"]["whole_pass_frac"], d["roofline"]["traffic"])
e = d["extra"]["e2e_files"]
for k in ("se_gz_seconds", "se_gz_first_call_seconds", "se_gz_cli_cold"): print(k, e.get(k))
c = e.get("configs4_se_gz", {})
print("configs4", {k: c.get(k) for k in ("seconds", "first_call_seconds", "reads_per_s", "cli_cold", "device_memory_in_use_peak_GB", "output_equals_host_pipeline_on_plain_text")}, c.get("roofline", {}).get("frac"), c.get("inflate_kernels", {}).get("text_GB_per_s"))
for k in ("configs4_se_plain", "configs1_pe_plain"): print(k, json.dumps(e.get(k))[:900])
x = d["extra"]
print("bait_sweep", {k: (round(v["ms_per_step"], 4), v["whole_pass_frac_of_hbm_peak"], v["roofline"]["bound"], v["roofline"]["frac"], v.get("window_bits_match_oracle")) for k, v in x.get("bait_sweep", {}).items() if "ms_per_step" in v})
print("thresholds", {k: (round(v["ms_per_step"], 4), v.get("with_hit_counts", {}).get("ms_per_step"), v.get("roofline", {}).get("frac")) for k, v in x.get("threshold_sweep", {}).items() if "ms_per_step" in v})
print("ragged", {k: x.get("ragged", {}).get(k) for k in ("ms_per_step", "ms_per_step_over_uniform", "window_bits_match_oracle")}, "realistic", {k: x.get("realistic", {}).get(k) for k in ("ms_per_step", "ms_per_step_over_iid", "whole_pass_frac_of_hbm_peak", "window_bits_match_oracle")})
print("real", json.dumps(e.get("real_compressors"))[:900])
print("fv2", json.dumps(d["extra"].get("filter_v2"))[:1200])
print("k_sweep", {k: (v.get("reads_per_s"), v.get("whole_pass_frac_of_hbm_peak"), v.get("roofline", {}).get("frac")) for k, v in d["extra"].get("k_sweep", {}).items()})
PY
