"""Multi-GPU path: ranks are independent -- every rank filters its own shard of whole pairs and there
is no collective -- so all bench.py needs from the launcher is RANK / WORLD_SIZE plus a barrier and a max over ranks,
which go through a directory in /dev/shm (no torch, no process group).  Covered here on CPU with two and four
processes; the GPU suite runs bench.py itself under torch.distributed.run with two ranks on one device
(tests/test_gpu_parity.py::test_bench_ranks_under_torchrun, two and eight ranks)."""
import multiprocessing as mp
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_PORT"] = str(port)
    os.environ["TORCHELASTIC_RUN_ID"] = "pytest%d" % port
    import bench
    r = bench.ShmRendezvous(rank, world)
    got = []
    for i in range(5):
        got.append(r.allmax(float(rank * 10 + i)))       # max over ranks of a rank-dependent value
        r.barrier()
    got.append(r.gather(float(rank) + 0.5))              # every rank's value, by rank (per-GPU figures of the bench line)
    r.close()
    q.put((rank, got))


@pytest.mark.parametrize("world", [2, 4])
def test_shm_rendezvous(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 20000 + os.getpid() % 20000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for r in range(world):
        assert res[r] == [float((world - 1) * 10 + i) for i in range(5)] + [[q + 0.5 for q in range(world)]]
    assert not any(n.startswith("mf_bench_%d_pytest%d" % (port, port)) for n in os.listdir("/dev/shm"))


def test_bench_workloads_follow_baseline_configs():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.READS_5GBP * 150 == pytest.approx(5e9, rel=1e-6) and bench.READS_5GBP % 2 == 0
    assert bench.READS_50GBP_8 * 8 * 150 == pytest.approx(50e9, rel=1e-6) and bench.READS_50GBP_8 % 2 == 0


def test_bench_line_shapes():
    """The three shapes of the bench line's roofline object (HBM-bound default, VALU-bound k < 28, multi-rank) as committed
    under profiles/r03/ carry the fields the review asks for."""
    import glob
    import json
    d = os.path.join(ROOT, "profiles", "r03")
    shapes = {os.path.basename(f): json.load(open(f)) for f in glob.glob(os.path.join(d, "bench_shape_*.json"))}
    if not shapes:
        pytest.skip("profiles/r03/bench_shape_*.json not committed yet")
    for name, line in shapes.items():
        r = line["roofline"]
        assert {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_alone_frac", "whole_pass_frac"} <= set(r), name
        assert len(line["extra"]["per_gpu_reads_per_s"]) == line["n_gpus"] == len(line["extra"]["per_rank_ms_per_step"]), name
        if "k21" in name:
            assert r["bound"] == "valu" and "hbm" in r and r["hbm"]["unit"] == "GB/s", name
        else:
            assert r["bound"] == "hbm" and r["unit"] == "GB/s", name


def test_bench_line_shapes_round4():
    """Round 4's additions to the default line as committed under profiles/r04/: the file-level leg against its own roof, files written by
    real compressors, the k sweep, the quality filter through the device ingest path."""
    import json
    f = os.path.join(ROOT, "profiles", "r04", "bench_shape_default.json")
    if not os.path.exists(f):
        pytest.skip("profiles/r04/bench_shape_default.json not committed yet")
    x = json.load(open(f))["extra"]
    c4 = x["e2e_files"]["configs4_se_gz"]
    assert c4["roofline"]["bound"] == "pcie_h2d" and 0 < c4["roofline"]["frac"] < 1 and c4["output_equals_host_pipeline_on_plain_text"] and c4["ingest_path"] == "device"
    assert all(v["output_equals_host_pipeline_on_plain_text"] for v in x["e2e_files"]["real_compressors"]["files"].values())
    assert {"21", "41"} <= set(x["k_sweep"])
    q = x["filter_v2"]
    assert q["cli_outputs_equal"] and q["library_outputs_equal"] and q["library_call"]["device"]["ingest_path"] == "device"
    assert {"device", "host"} <= set(q["cli_process_start_to_exit"])


def test_bench_line_shapes_round5():
    """Round 5's additions to the default line as committed under profiles/r05/: the cold calls through the CLI (a process per call is the
    reference's boundary), first-call times, plain files through the device path against their PCIe roof, the same-size control beside the
    real compressors' files, a roofline object in every leg of the k sweep."""
    import json
    f = os.path.join(ROOT, "profiles", "r05", "bench_shape_default.json")
    if not os.path.exists(f):
        pytest.skip("profiles/r05/bench_shape_default.json not committed yet")
    x = json.load(open(f))["extra"]
    e = x["e2e_files"]
    c4 = e["configs4_se_gz"]
    assert c4["cli_cold"]["kept_equals_library"] and c4["cli_cold"]["seconds"] > 0 and c4["first_call_seconds"] >= c4["seconds"] > 0
    assert e["se_gz_first_call_seconds"] > 0 and e["se_gz_cli_cold"]["seconds"] > 0
    for leg in ("configs4_se_plain", "configs1_pe_plain"):
        p = e[leg]
        assert p["outputs_equal"] and p["device_path"]["ingest_path"] == "device" and p["roofline"]["bound"] == "pcie_h2d" and 0 < p["roofline"]["frac"] < 1.05, leg
    assert any("pgzip" in k for k in e["real_compressors"]["files"]) and "gzip-6" in e["real_compressors"]["files"]
    for k, leg in x["k_sweep"].items():
        r = leg["roofline"]
        assert r["bound"] == ("valu" if int(k) < 28 else "hbm") and r["peak"] > 0 and "frac" in r, k
