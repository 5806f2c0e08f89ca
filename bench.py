#!/usr/bin/env python3
"""Headline benchmark: filtered reads/sec on 5 Gbp synthetic PE150, k=31 (BASELINE.json configs[1]).

A "step" is one pass of the hot path (screen kernel + exact kernel) over the
whole read set, already packed and resident in HBM.  `value` = reads of all
ranks x steps / wall time (barrier + device sync on both sides, max over ranks).
`roofline` prices the dominant kernel (the screen kernel, which streams every
packed byte once) with ALGORITHMIC bytes: 37.5 B of packed bases + 1 result bit
per 150-base read, divided by the kernel's average duration
measured with HIP events on the library's stream.  `cpu_baseline` times the
oracle (a port: the reference has no k-mer filter) on a bounded sample of the
same reads on this box's host cores.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

READ_LEN = 150
READS_5GBP = 33_333_334          # 16 666 667 pairs x 2 mates
K = 31
THRESHOLD = 1
HBM_PEAK_GBPS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reads", type=int, default=READS_5GBP, help="reads per GPU (default: the 5 Gbp set)")
    ap.add_argument("--k", type=int, default=K)
    ap.add_argument("--cpu-sample", type=int, default=READS_5GBP,
                    help="reads timed on the CPU oracle, all host threads (default: the whole set, a few seconds; 0 = skip)")
    ap.add_argument("--no-exhaustive", action="store_true", help="skip the extra exhaustive-mode measurement")
    return ap.parse_args()


def committed_traffic():
    """HBM bytes per screen-kernel launch from the newest committed PMC profile (profiles/rNN/traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    gfx950).  PMC counters cannot be collected from inside this process, so this is the profiled value of the
    same command, not a live reading; None when no profile is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        return int(d["screen_kernel"]["hbm_bytes_per_launch"]), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def cpu_quota():
    """CPUs the container may actually use (cgroup v2 cpu.max), or None when unlimited / unknown: the oracle runs on
    every visible hardware thread, but a quota caps what those threads get."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(p)
    except Exception:
        return None


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world

    # product library first (system ROCm runtime), torch only afterwards and only for the rendezvous
    from mitoflex_amd import mitofilter as mf
    mf.load()
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    from tests.util_data import make_bait
    bait = make_bait()
    dev = local_rank % max(1, mf.device_count())      # one rank per GPU; wraps only when ranks outnumber GPUs (tests)
    ks = mf.KmerSet.from_text(bait, a.k, dev)
    want_cpu = rank == 0 and world == 1 and a.cpu_sample > 0
    t0 = time.time()
    # every rank owns its own shard of whole pairs (weak scaling: fixed reads per GPU), no collective
    reads = mf.Reads.synth(a.reads, READ_LEN, seed=20261003 + rank, bait_text=bait,
                           mito_ppm=5000, sub_ppm=10000, n_read_ppm=10000, n_base_ppm=1000,
                           device=dev, keep_host=want_cpu)
    t_gen = time.time() - t0

    def barrier():
        mf.device_synchronize(dev)
        if dist is not None:
            dist.barrier()

    if a.warmup > 0:
        mf.filter_resident(ks, reads, THRESHOLD, mf.MODE_SCREENED, a.warmup)
    barrier()
    t0 = time.perf_counter()
    st = mf.filter_resident(ks, reads, THRESHOLD, mf.MODE_SCREENED, a.steps)
    mf.device_synchronize(dev)
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()

    total_reads = a.reads * world
    value = total_reads * a.steps / elapsed
    alg_bytes = st.algorithmic_bytes                     # per launch: ceil(2*bases/8) + ceil(reads/8)
    screen_s = st.ms_screen / 1e3
    achieved = alg_bytes / screen_s / 1e9 if screen_s > 0 else 0.0

    extra = {"ms_screen_kernel": round(st.ms_screen, 4), "ms_mark_kernel": round(st.ms_mark, 4), "ms_exact_kernel": round(st.ms_exact, 4),
             "ms_pass_events": round(st.ms_total, 4), "candidates": int(st.n_candidates), "passed": int(st.n_pass),
             "reads_per_gpu": a.reads, "synth_seconds": round(t_gen, 2), "device": mf.device_name(dev)}
    if rank == 0 and not a.no_exhaustive:
        ex = mf.filter_resident(ks, reads, THRESHOLD, mf.MODE_EXHAUSTIVE, 1)
        extra["exhaustive_reads_per_s"] = a.reads / (ex.ms_total / 1e3)
        extra["exhaustive_passed"] = int(ex.n_pass)

    cpu = None
    if want_cpu:
        import numpy as np
        from oracle import oracle_lib as ol
        n = min(a.cpu_sample, a.reads)
        cores = os.cpu_count() or 1
        off = np.arange(n + 1, dtype=np.uint64) * READ_LEN
        R = ol.OracleReads.from_arrays(reads.host_words, off, reads.host_npos[reads.host_npos < n * READ_LEN])
        T = ol.OracleTable(bait, a.k)
        c0 = time.perf_counter()
        obits, _ = ol.filter_reads(T, R, THRESHOLD, threads=cores)
        cs = time.perf_counter() - c0
        what = "all" if n == a.reads else "first"
        quota = cpu_quota()
        cpu = {"value": n / cs, "unit": "reads/s", "cores": cores, "kind": "port",
               "sample": f"{what} {n} reads of the same synthetic set, oracle/kmer_bait_oracle.c, {cores} threads, {cs:.1f}s"
                         + (f"; the container's cgroup quota is {quota:g} CPUs" if quota else ""),
               "cpu_quota": quota}
        # the same sample doubles as a checker: GPU bits of the sample == oracle bits
        gbits, _, _ = mf.filter_reads(ks, reads, THRESHOLD, mf.MODE_SCREENED)
        nw = n // 32
        extra["sample_bits_match_oracle"] = bool(np.array_equal(gbits[:nw], obits[:nw]))
        extra["sample_reads_checked"] = nw * 32

    traffic, traffic_src = committed_traffic()
    if a.reads != READS_5GBP or a.k != K:
        traffic, traffic_src = None, None              # the profile was taken on the default workload
    if rank == 0:
        out = {
            "metric": "filtered reads/sec on 5 Gbp PE150 k=31; achieved HBM GB/s vs peak",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64" if a.k <= 32 else "u128", "data": "synthetic",
            "config": {"workload": f"{a.reads} reads x {READ_LEN} b per GPU ({a.reads * READ_LEN / 1e9:.2f} Gbp, "
                                   f"synthetic PE150: 0.5% bait-derived reads with 1% substitutions, N in 1% of reads), "
                                   f"k={a.k}, threshold={THRESHOLD}, bait = synthetic 16 569 bp mitogenome + 1.2 kbp record; "
                                   "reads packed 2 bit/base and resident in HBM",
                       "sharding": f"{world} rank(s), one per GPU, independent shards of whole pairs, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "screen_kernel", "algorithmic_bytes_per_launch": int(alg_bytes),
                         "avg_kernel_ms": st.ms_screen},
            "cpu_baseline": cpu,
            "extra": extra,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
