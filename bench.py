#!/usr/bin/env python3
"""Headline benchmark: filtered reads/sec on 5 Gbp synthetic PE150, k=31 (BASELINE.json configs[1]).

A "step" is one pass of the hot path over the whole read set, already packed and resident in HBM:
screen_kernel (streams every packed byte once) + finish_kernel x 2 (settle the stage-1 positives, set the pass
bits).  Consecutive steps are pipelined the way consecutive batches of a file are: the finish kernels of step i
run on a second stream under the screen kernel of step i + 1; every step does all of its work and the timed
region ends when the last finish kernel has.  `value` = reads of all ranks x steps / wall time (barrier + device
sync on both sides, max over ranks).  `roofline` prices the dominant kernel (screen_kernel) with ALGORITHMIC
bytes: 37.5 B of packed bases + 1 result bit per 150-base read, divided by the kernel's average
duration over a separate, fully sampled loop of the same K steps (HIP events attached to every dispatch on the
library's streams; sampling every dispatch costs a little, so the timed loop itself is not sampled).
`cpu_baseline` times the oracle (a port: the reference has no k-mer filter) on the same reads on this box's
host cores and checks every GPU pass bit against it.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Ranks are independent (each filters its own shard, no collective); the launcher's RANK / WORLD_SIZE are all this
script needs, so torch is never imported: the barrier and the max over ranks go through a directory in /dev/shm.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

READ_LEN = 150
READS_5GBP = 33_333_334          # 16 666 667 pairs x 2 mates: configs[1]
READS_50GBP_8 = 41_666_668       # configs[3]: 50 Gbp = 333 333 334 reads over 8 GPUs, whole pairs per GPU
K = 31
THRESHOLD = 1
HBM_PEAK_GBPS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--prewarm-ms", type=float, default=100.0, help="keep the device busy with untimed passes for this long before the warm-up steps (clock ramp)")
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: the 5 Gbp set; at 8 GPUs a 1/8 shard of the 50 Gbp set)")
    ap.add_argument("--k", type=int, default=K)
    ap.add_argument("--cpu-sample", type=int, default=READS_5GBP,
                    help="reads timed on the CPU oracle, all host threads (default: the whole set, a few seconds; 0 = skip)")
    ap.add_argument("--no-exhaustive", action="store_true", help="skip the extra exhaustive-mode measurement")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from the committed profile instead of two rocprofv3 --pmc child runs")
    ap.add_argument("--e2e-pairs", type=int, default=500_000, help="pairs of the bounded files-in/files-out run reported in extra (0 = skip)")
    return ap.parse_args()


class ShmRendezvous:
    """Barrier and max-reduce for the ranks of one node through files in /dev/shm (one directory per launch)."""

    def __init__(self, rank, world):
        self.rank, self.world, self.n = rank, world, 0
        # one directory per launch: the launcher (parent of every rank) is part of the name, so that what a crashed earlier launch
        # with the same port and run id left behind is never read
        key = "%s_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", ""), os.getppid())
        self.dir = "/dev/shm/mf_bench_%s_%d" % (key, os.getuid())
        os.makedirs(self.dir, exist_ok=True)

    def allmax(self, value=0.0, timeout=600.0):
        self.n += 1
        mine = os.path.join(self.dir, "%d.%d" % (self.n, self.rank))
        with open(mine + ".tmp", "w") as f:
            f.write(repr(float(value)))
        os.rename(mine + ".tmp", mine)
        vals, t0 = [], time.time()
        for r in range(self.world):
            p = os.path.join(self.dir, "%d.%d" % (self.n, r))
            while not os.path.exists(p):
                if time.time() - t0 > timeout:
                    raise RuntimeError("rank %d: rank %d did not reach barrier %d" % (self.rank, r, self.n))
                time.sleep(0.0005)
            vals.append(float(open(p).read()))
        return max(vals)

    barrier = allmax

    def close(self, timeout=120.0):
        self.allmax()
        open(os.path.join(self.dir, "done.%d" % self.rank), "w").close()       # this rank has read everything it will ever read here
        if self.rank == 0:                                                      # ... so rank 0 may take the directory away once all have
            t0 = time.time()
            while not all(os.path.exists(os.path.join(self.dir, "done.%d" % r)) for r in range(self.world)) and time.time() - t0 < timeout:
                time.sleep(0.001)
            import shutil
            shutil.rmtree(self.dir, ignore_errors=True)


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


def live_traffic(timeout_s=75.0):
    """HBM bytes per screen-kernel launch of THIS workload on THIS box: two short child runs of this script under
    `rocprofv3 --pmc` (FETCH_SIZE, then WRITE_SIZE -- one counter a pass, as MI355X_MICROARCH.md prescribes; counters cannot be
    read from inside a process), FETCH_SIZE doubled for gfx950.  (bytes, description) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process is being profiled itself"
    kib = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="mf_pmc_", dir="/tmp")
        try:
            # (the profiled program comes right after `--`: no shell, no env wrapper in between)
            cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
                   "--steps", "2", "--warmup", "0", "--prewarm-ms", "0", "--cpu-sample", "0", "--no-exhaustive", "--e2e-pairs", "0", "--no-live-traffic"]
            subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", timeout=timeout_s, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            vals = [float(r["Counter_Value"]) for f in files for r in csv.DictReader(open(f))
                    if "mf::screen_kernel<" in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter]      # (not build_screen_kernel)
            if not vals:
                return None, f"no {counter} rows for screen_kernel"
            kib[counter] = sum(vals) / len(vals)
        except Exception as e:
            return None, f"{counter} pass failed: {str(e)[:120]}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return int((2 * kib["FETCH_SIZE"] + kib["WRITE_SIZE"]) * 1024), \
        "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this script on this box (FETCH_SIZE x 2 on gfx950, KiB counters)"


def cpu_quota():
    """CPUs the container may actually use (cgroup v2 cpu.max), or None when unlimited / unknown: the oracle runs on
    every visible hardware thread, but a quota caps what those threads get."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(p)
    except Exception:
        return None


def e2e_files(mf, ks, pairs):
    """Bounded files-in / files-out run (host bound by construction: parse, pack, PCIe, write): reads/s of the
    whole mf_filter_fastq_files call, best of three, for plain PE and gzipped SE input (configs[4] shape)."""
    import tempfile
    out = {}
    with tempfile.TemporaryDirectory(prefix="mf_e2e_") as t:
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_fastq.py"), os.path.join(t, "s"), "--pairs", str(pairs)],
                              stdout=subprocess.DEVNULL)
        f1, f2 = os.path.join(t, "s_1.fq"), os.path.join(t, "s_2.fq")
        with open(f1 + ".gz", "wb") as g:
            subprocess.check_call(["gzip", "-1", "-c", f1], stdout=g)

        def run(a, b, o1, o2, n_reads):
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                mf.filter_fastq_files(ks, a, b, o1, o2)
                best = min(best, time.perf_counter() - t0)
            return n_reads / best
        out["pe_plain_reads_per_s"] = run(f1, f2, os.path.join(t, "o1.fq"), os.path.join(t, "o2.fq"), 2 * pairs)
        out["se_gz_reads_per_s"] = run(f1 + ".gz", None, os.path.join(t, "og.fq"), None, pairs)
        out["pairs"] = pairs
    return out


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world
    if a.reads <= 0:
        a.reads = READS_50GBP_8 if world == 8 else READS_5GBP
    rdv = ShmRendezvous(rank, world) if world > 1 else None

    # roofline.traffic, live (two short counter runs of this script as child processes, before this process touches the GPU)
    live = (None, None)
    if rank == 0 and world == 1 and not a.no_live_traffic and a.reads == READS_5GBP and a.k == K:
        live = live_traffic()

    from mitoflex_amd import mitofilter as mf
    mf.load()
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
        if rdv is not None:
            rdv.barrier()

    os.environ["MF_EVENT_STRIDE"] = "1000000000"        # the timed loop carries no per-kernel events
    # W steps of this workload last about a millisecond, far less than the device needs to come out of its idle clocks after the
    # (host-side) set-up above: keep it busy with the same passes for a fixed time first, untimed like the warm-up steps
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < a.prewarm_ms:
        mf.filter_resident(ks, reads, THRESHOLD, mf.MODE_SCREENED, 20)
    if a.warmup > 0:
        mf.filter_resident(ks, reads, THRESHOLD, mf.MODE_SCREENED, a.warmup)
    barrier()
    t0 = time.perf_counter()
    st = mf.filter_resident(ks, reads, THRESHOLD, mf.MODE_SCREENED, a.steps)
    mf.device_synchronize(dev)
    elapsed = time.perf_counter() - t0
    if rdv is not None:
        elapsed = rdv.allmax(elapsed)

    # kernel durations: the same K steps again, every dispatch sampled
    os.environ["MF_EVENT_STRIDE"] = "1"
    sp = mf.filter_resident(ks, reads, THRESHOLD, mf.MODE_SCREENED, a.steps)
    os.environ["MF_EVENT_STRIDE"] = "8"

    total_reads = a.reads * world
    value = total_reads * a.steps / elapsed
    alg_bytes = st.algorithmic_bytes                     # per launch: ceil(2*bases/8) + ceil(reads/8)
    screen_s = sp.ms_screen / 1e3
    achieved = alg_bytes / screen_s / 1e9 if screen_s > 0 else 0.0

    extra = {"ms_screen_kernel": round(sp.ms_screen, 4), "ms_mark_kernel": round(sp.ms_mark, 4), "ms_finish_kernel_phase0": round(sp.ms_exact, 4),
             "ms_per_step_sampled_loop": round(sp.ms_total, 4), "kernel_samples": a.steps,
             "ms_pass_events": round(st.ms_total, 4), "work_items": int(st.n_candidates), "passed": int(st.n_pass),
             "reads_per_gpu": a.reads, "synth_seconds": round(t_gen, 2), "prewarm_ms": a.prewarm_ms, "device": mf.device_name(dev),
             "pipelined": "consecutive steps overlap: finish kernels of step i run under the screen kernel of step i+1 (second stream) and "
                          "consecutive screen kernels go to two streams in turn, so the duration of a screen launch (ms_screen_kernel, "
                          "roofline.achieved) includes time it shares the device with its neighbours; a step completes every ms_per_step"}
    if rank == 0 and world == 1:
        one = mf.filter_resident(ks, reads, THRESHOLD, mf.MODE_SCREENED, 1)    # a single step: nothing to overlap with
        extra["ms_single_pass_latency"] = round(one.ms_total, 4)
        # the same kernel with the device to itself (MF_PASS=serial: one stream, no overlap between steps), every dispatch sampled
        prev = os.environ.get("MF_PASS")
        os.environ["MF_PASS"] = "serial"; os.environ["MF_EVENT_STRIDE"] = "1"
        alone = mf.filter_resident(ks, reads, THRESHOLD, mf.MODE_SCREENED, min(a.steps, 20))
        os.environ["MF_EVENT_STRIDE"] = "8"
        if prev is None:
            del os.environ["MF_PASS"]
        else:
            os.environ["MF_PASS"] = prev
        if alone.ms_screen > 0:
            extra["screen_kernel_alone"] = {"ms": round(alone.ms_screen, 4), "frac_of_hbm_peak": round(alg_bytes / (alone.ms_screen / 1e3) / 1e9 / HBM_PEAK_GBPS, 4),
                                            "ms_per_step_serial": round(alone.ms_total, 4)}
    extra["whole_pass_frac_of_hbm_peak"] = round(alg_bytes / (elapsed / a.steps) / 1e9 / HBM_PEAK_GBPS, 4)
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
    if rank == 0 and world == 1 and a.e2e_pairs > 0:
        try:
            extra["e2e_files"] = e2e_files(mf, ks, a.e2e_pairs)
        except Exception as e:           # the headline numbers do not depend on scratch space for files
            extra["e2e_files"] = {"error": str(e)[:200]}

    traffic, traffic_src = committed_traffic()
    if a.reads != READS_5GBP or a.k != K:
        traffic, traffic_src = None, None              # the profile was taken on the default workload
    elif live[0] is not None:
        traffic, traffic_src = live
    elif live[1] and traffic_src:
        traffic_src += f" (live collection unavailable: {live[1]})"
    if rank == 0:
        out = {
            "metric": "filtered reads/sec on 5 Gbp PE150 k=31; achieved HBM GB/s vs peak",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64" if a.k <= 32 else "u128", "data": "synthetic",
            "config": {"workload": f"{a.reads} reads x {READ_LEN} b per GPU ({a.reads * READ_LEN / 1e9:.2f} Gbp"
                                   + (", a 1/8 shard of the 50 Gbp set of configs[3]" if world == 8 and a.reads == READS_50GBP_8 else "")
                                   + f", synthetic PE150: 0.5% bait-derived reads with 1% substitutions, N in 1% of reads), "
                                   f"k={a.k}, threshold={THRESHOLD}, bait = synthetic 16 569 bp mitogenome + 1.2 kbp record; "
                                   "reads packed 2 bit/base and resident in HBM",
                       "sharding": f"{world} rank(s), one per GPU, independent shards of whole pairs, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "screen_kernel", "algorithmic_bytes_per_launch": int(alg_bytes),
                         "avg_kernel_ms": sp.ms_screen, "kernel_launches_averaged": a.steps},
            "cpu_baseline": cpu,
            "extra": extra,
        }
        print(json.dumps(out))
    if rdv is not None:
        rdv.close()


if __name__ == "__main__":
    main()
