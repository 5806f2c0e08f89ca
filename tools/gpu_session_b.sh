cd $GRAFT_REPO_ROOT; O=gpurun_out/r05d; mkdir -p $O
bash tools/cold_calls.sh r05d > /dev/null 2>&1; grep -E "^==|wall |cold \+|buffers of this call" $O/cold_calls.log | cut -c1-330
timeout 1200 python -m pytest tests/test_gpu_devingest.py tests/test_filter_v2.py tests/test_callsite.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_ingest_tail.txt
timeout 300 python tools/k41_probe.py 2>&1 | tee $O/k41_probe.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -3 $O/bench_default.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r05d/bench_default.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["whole_pass_frac"])
e = d["extra"]["e2e_files"]
for k in ("se_gz_seconds", "se_gz_first_call_seconds", "se_gz_cli_cold"): print(k, e.get(k))
c = e.get("configs4_se_gz", {})
print("configs4", {k: c.get(k) for k in ("seconds", "first_call_seconds", "reads_per_s", "cli_cold", "device_memory_in_use_peak_GB", "output_equals_host_pipeline_on_plain_text")}, c.get("roofline", {}).get("frac"), c.get("inflate_kernels", {}).get("text_GB_per_s"))
for k in ("configs4_se_plain", "configs1_pe_plain"): print(k, json.dumps(e.get(k))[:900])
print("real", json.dumps(e.get("real_compressors"))[:1200])
print("fv2", json.dumps(d["extra"].get("filter_v2"))[:1500])
print("k_sweep", json.dumps(d["extra"].get("k_sweep"))[:1800])
PY
