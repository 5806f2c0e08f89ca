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

Beside the headline, at N = 1 (`extra`): files in / files out -- a bounded PE / SE run, configs[4] at its stated size
(33 333 334 single-end reads, one gzip member, device ingest: inflate, line index, pack, filter on the GPU; priced against the
box's host-to-device copy rate, its roof) and an eighth of it compressed by the compressors people use (gzip, pigz, bgzip where the
box has them) --, configs[2]'s other k (21 and 41) on the same resident reads (`extra.k_sweep`), and the Group-A row of
BASELINE.md (the contig filter CLI on the 1 M-record generator file of SURVEY.md 8c).  At N > 1 every rank also filters a
.gz shard of its own, file to file, on its own GPU (`extra.per_gpu_file_reads_per_s`).  Every input file is made BEFORE this
process touches the GPU, by child processes that do not inherit a profiler's environment.
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
    ap.add_argument("--e2e-full-reads", type=int, default=READS_5GBP,
                    help="reads of the single-end .gz file of configs[4] that is filtered file to file (default: its stated size, 5 Gbp; 0 = skip)")
    ap.add_argument("--no-group-a", action="store_true", help="skip the Group-A row (contig filter CLI on the 1 M-record generator file)")
    ap.add_argument("--fv2-pairs", type=int, default=2_000_000,
                    help="pairs of the .gz pair the reference's quality filter (filter_v2, SURVEY.md 8f next #2) is timed on, device ingest path against host pipeline (0 = skip)")
    ap.add_argument("--real-gz-reads", type=int, default=READS_5GBP // 8,
                    help="reads of the single-end file that gzip / pigz / bgzip compress themselves (default: an eighth of configs[4]; 0 = skip)")
    ap.add_argument("--plain-pairs", type=int, default=READS_5GBP // 2,
                    help="pairs of the plain paired files of configs[1] put through the file boundary (device path against host pipeline; 0 = skip)")
    ap.add_argument("--k-sweep", default="21,41", help="other k of configs[2] timed on the same resident reads, a few steps each ('none' = none)")
    ap.add_argument("--no-axes", action="store_true", help="skip extra.bait_sweep / threshold_sweep / ragged / realistic (the resident path off the headline's one point)")
    ap.add_argument("--rank-file-reads", type=int, default=500_000, help="N > 1: reads of the .gz shard every rank filters file to file on its own GPU (0 = skip)")
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

    def gather(self, value, timeout=600.0):
        """every rank's value, by rank"""
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
                    raise RuntimeError("rank %d: rank %d did not reach gather %d" % (self.rank, r, self.n))
                time.sleep(0.0005)
            vals.append(float(open(p).read()))
        return vals

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


def being_profiled():
    return any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")


def child_env(**extra):
    """environment for child processes: nothing of a profiler that may be wrapped around this process (a child that inherits
    its preloaded library is a GPU-initialised fork that execs -- the hop the pool forbids)"""
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER", "HSA_TOOLS"))}
    env.update(extra)
    return env


def live_counters(counters, k, timeout_s=75.0):
    """Per-launch averages of PMC counters of the dominant kernel of THIS workload on THIS box: one short child run of this
    script under `rocprofv3 --pmc` per counter (one counter a pass, as MI355X_MICROARCH.md prescribes; counters cannot be read
    from inside a process).  ({counter: value}, None) or (None, reason).  A child that overruns is killed with its whole
    process group, and waited for, before this process goes on to the GPU."""
    import csv
    import glob
    import shutil
    import signal
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    if being_profiled():
        return None, "this process is being profiled itself"
    out = {}
    for counter in counters:
        d = tempfile.mkdtemp(prefix="mf_pmc_", dir="/tmp")
        try:
            # (the profiled program comes right after `--`: no shell, no env wrapper in between)
            cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--k", str(k),
                   "--steps", "2", "--warmup", "0", "--prewarm-ms", "0", "--cpu-sample", "0", "--no-exhaustive", "--e2e-pairs", "0",
                   "--e2e-full-reads", "0", "--fv2-pairs", "0", "--no-group-a", "--no-live-traffic", "--k-sweep", "none", "--real-gz-reads", "0", "--plain-pairs", "0", "--no-axes"]
            p = subprocess.Popen(cmd, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                p.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, signal.SIGKILL)          # the launcher AND the profiled script: nothing of it may share the GPU with the timed loop
                except ProcessLookupError:
                    pass
                p.wait()
                return None, f"live collection timed out ({counter})"
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            vals = [float(r["Counter_Value"]) for f in files for r in csv.DictReader(open(f))
                    if "mf::screen_kernel<" in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter]      # (not build_screen_kernel)
            if not vals:
                return None, f"no {counter} rows for screen_kernel"
            out[counter] = sum(vals) / len(vals)
        except Exception as e:
            return None, f"{counter} pass failed: {str(e)[:120]}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return out, None


def live_traffic(k):
    """HBM bytes per screen-kernel launch: FETCH_SIZE doubled for gfx950 + WRITE_SIZE (KiB counters).  (bytes, description) or (None, reason)."""
    c, why = live_counters(("FETCH_SIZE", "WRITE_SIZE"), k)
    if c is None:
        return None, why
    return int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024), \
        "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this script on this box (FETCH_SIZE x 2 on gfx950, KiB counters)"


def cpu_throttled_usec():
    """microseconds the container's cgroup has been throttled by its CPU quota so far (cgroup v2 cpu.stat), or None"""
    try:
        for ln in open("/sys/fs/cgroup/cpu.stat"):
            if ln.startswith("throttled_usec"):
                return int(ln.split()[1])
    except Exception:
        pass
    return None


def cpu_quota():
    """CPUs the container may actually use (cgroup v2 cpu.max), or None when unlimited / unknown: the oracle runs on
    every visible hardware thread, but a quota caps what those threads get."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(p)
    except Exception:
        return None


G20_COUNT, G20_MD5 = 398188, "59347e825357644e5116d9f8b988cb82"       # SURVEY.md 8c, captured from the reference's ELF
G20_GENERATOR = """
import random, sys
random.seed(20261003)
with open(sys.argv[1], "w") as f:
    for i in range(1_000_000):
        L = random.randint(60, 600)
        multi = random.choice([1.0, 2.0, 3.5, 9.9999, 10.0, 25.25, 300.0]) * random.random() * 2
        f.write(f">k31_{i} flag={random.randint(0, 2)} multi={multi:.4f} len={L}\\n")
        f.write(''.join(random.choices('ACGT', k=L)) + "\\n")
"""


def md5_of(path):
    import hashlib
    h = hashlib.md5()
    with open(path, "rb") as f:
        for b in iter(lambda: f.read(1 << 24), b""):
            h.update(b)
    return h.hexdigest()


def prepare_inputs(a, tmp, solo=True):
    """Every file the file-level measurements read, made by child processes BEFORE this process initialises the GPU (and not
    at all when this process is being profiled: the children would inherit the profiler).  -> {name: path or note}"""
    import shutil
    files = {}
    if being_profiled():
        return {"skipped": "this process is being profiled: no child processes, no file-level runs"}
    env = child_env()
    run = lambda cmd, **kw: subprocess.check_call(cmd, env=env, stdout=subprocess.DEVNULL, **kw)
    try:
        if a.e2e_pairs > 0:
            run([sys.executable, os.path.join(ROOT, "tools", "make_fastq.py"), os.path.join(tmp, "s"), "--pairs", str(a.e2e_pairs)])
            with open(os.path.join(tmp, "s_1.fq.gz"), "wb") as g:
                subprocess.check_call(["gzip", "-1", "-c", os.path.join(tmp, "s_1.fq")], stdout=g, env=env)
            files["small"] = os.path.join(tmp, "s")
        real = []
        if a.real_gz_reads > 0 and solo and shutil.which("gzip"):
            # the same kind of reads, compressed by the tools themselves (single-threaded gzip takes its minute: started now, waited for below)
            run([sys.executable, os.path.join(ROOT, "tools", "make_fastq.py"), os.path.join(tmp, "r"), "--pairs", str(a.real_gz_reads), "--mates", "1", "--block", "2000000"])
            src = os.path.join(tmp, "r_1.fq")
            for name, cmd in (("gzip-6", ["gzip", "-6", "-c", src]), ("pigz-6", ["pigz", "-6", "-c", src]), ("bgzip", ["bgzip", "-c", src])):
                if shutil.which(cmd[0]):
                    out = os.path.join(tmp, "r_1.%s.fq.gz" % name)
                    real.append((name, out, subprocess.Popen(cmd, stdout=open(out, "wb"), env=env)))
            # the control: the SAME text through this repository's tools/pgzip.py (what the configs[4] leg is compressed with), so that a
            # difference between the two figures is the compressor's and not the size's
            out = os.path.join(tmp, "r_1.pgzip-6.fq.gz")
            real.append(("pgzip-6 (control: tools/pgzip.py, same text)", out, subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "pgzip.py"), src, out, "--level", "6"],
                                                                                                 stdout=subprocess.DEVNULL, env=env)))
        if a.e2e_full_reads > 0 and solo:                 # (configs[4] is a one-GPU configuration: the other ranks of a multi-rank run would only wait for the file)
            need = a.e2e_full_reads * 321 * 1.7
            if shutil.disk_usage(tmp).free < need:
                files["full_note"] = "not enough scratch space for the configs[4] file"
            else:
                t0 = time.time()
                run([sys.executable, os.path.join(ROOT, "tools", "make_fastq.py"), os.path.join(tmp, "f"), "--pairs", str(a.e2e_full_reads), "--mates", "1", "--block", "2000000"])
                t1 = time.time()
                run([sys.executable, os.path.join(ROOT, "tools", "pgzip.py"), os.path.join(tmp, "f_1.fq"), os.path.join(tmp, "f_1.fq.gz"), "--level", "6"])
                files["full"] = os.path.join(tmp, "f")
                files["full_prep_seconds"] = {"generate": round(t1 - t0, 1), "compress": round(time.time() - t1, 1)}
        # ---- the path as the reference calls it: a PROCESS per call (utility/helper.py:78-86, assemble_wrapper.py:326-343) -- `fastfilter bait`
        # from process start to exit, HIP start-up and all, before this process touches the GPU
        bait_cli = os.path.join(ROOT, "mitoflex_amd", "assemble", "fastfilter")

        def cold(bait, fq, out, reps):
            secs, kept = [], None
            for _ in range(reps):
                if os.path.exists(out):
                    os.unlink(out)
                t0 = time.perf_counter()
                kept = int(subprocess.check_output([bait_cli, "bait", "--bait", bait, "--fq1", fq, "--out1", out], env=env, stderr=subprocess.DEVNULL, timeout=300).decode())
                secs.append(round(time.perf_counter() - t0, 4))
            return {"seconds_each": secs, "seconds": min(secs), "kept": kept}
        for name, out, proc in real:          # (the compressors started above have had the time of the steps in between; nothing of ours runs beside the cold calls)
            proc.wait()
        try:
            if "small" in files:
                files["small_cli"] = cold(os.path.join(tmp, "s.bait.fa"), os.path.join(tmp, "s_1.fq.gz"), os.path.join(tmp, "s_cli.fq"), 3)
            if "full" in files:
                files["full_cli"] = cold(os.path.join(tmp, "f.bait.fa"), os.path.join(tmp, "f_1.fq.gz"), os.path.join(tmp, "f_cli.fq"), 3)
        except Exception as e:
            files["cli_error"] = str(e)[:200]
        if a.plain_pairs > 0 and solo:
            # configs[1] through the file boundary: the 5 Gbp PE150 set as two plain FASTQ files
            if shutil.disk_usage(tmp).free < a.plain_pairs * 2 * 321 * 1.3:
                files["plain_note"] = "not enough scratch space for the plain configs[1] pair"
            else:
                run([sys.executable, os.path.join(ROOT, "tools", "make_fastq.py"), os.path.join(tmp, "c"), "--pairs", str(a.plain_pairs), "--block", "2000000"])
                files["plain"] = os.path.join(tmp, "c")
        if a.fv2_pairs > 0 and solo:
          try:          # (a leg of its own: whatever goes wrong here costs the line this leg, not the file-level legs below)
            # the quality filter's drop-in CLI, process start to exit, on a .gz pair: device ingest path (the default) and host pipeline
            run([sys.executable, os.path.join(ROOT, "tools", "make_fastq.py"), os.path.join(tmp, "q"), "--pairs", str(a.fv2_pairs), "--block", "2000000"])
            for m in ("1", "2"):
                run([sys.executable, os.path.join(ROOT, "tools", "pgzip.py"), os.path.join(tmp, "q_%s.fq" % m), os.path.join(tmp, "q_%s.fq.gz" % m), "--level", "6"])
            exe = os.path.join(ROOT, "mitoflex_amd", "filter", "filter_v2")
            q = os.path.join(tmp, "q")
            cli = {}
            for tag, extra_env, reps in (("device", {}, 3), ("host", {"MF_QUAL_INGEST": "host"}, 2)):
                best = 1e9
                for _ in range(reps):
                    for o in ("_c1.fq", "_c2.fq"):
                        if os.path.exists(q + tag + o):
                            os.unlink(q + tag + o)
                    t0 = time.perf_counter()
                    subprocess.check_call([exe, "-1", q + "_1.fq.gz", "-2", q + "_2.fq.gz", "-3", q + tag + "_c1.fq", "-4", q + tag + "_c2.fq", "-d"], env=dict(env, **extra_env),
                                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=180)
                    best = min(best, time.perf_counter() - t0)
                cli[tag] = best
            files["fv2"] = {"prefix": q, "cli_seconds": cli}
          except Exception as e:
            files["fv2_error"] = str(e)[:200]
        if not a.no_group_a and solo:
            run([sys.executable, "-c", G20_GENERATOR, os.path.join(tmp, "g20.fa")])
            files["g20"] = os.path.join(tmp, "g20.fa")
        for name, out, proc in real:
            if proc.returncode == 0:
                files.setdefault("real", {"plain": os.path.join(tmp, "r_1.fq"), "bait": os.path.join(tmp, "r.bait.fa"), "gz": {}})["gz"][name] = out
    except Exception as e:
        files["error"] = str(e)[:200]
    return files


def e2e_files(mf, ks, files, a):
    """Files in / files out (mf_filter_fastq_files), reads/s of the whole call, best of three (configs[4]: of five)."""
    out = {}

    first_call = {}

    each = {}          # every call's seconds of the last run() (the legs print them all: a minimum alone hides the call-to-call spread)

    def run(f1, f2, o1, o2, n_reads, reps=3, tag=None):
        best, res, secs, thr = 1e9, None, [], []
        for i in range(reps):
            th0 = cpu_throttled_usec()
            t0 = time.perf_counter()
            res = mf.filter_fastq_files(ks, f1, f2, o1, o2)
            dt = time.perf_counter() - t0
            th1 = cpu_throttled_usec()
            if i == 0 and tag:
                first_call[tag] = round(dt, 4)
            secs.append(round(dt, 4))
            thr.append(round((th1 - th0) / 1e3, 1) if th0 is not None and th1 is not None else None)
            best = min(best, dt)
        each["last"], each["throttled"] = secs, thr
        return n_reads / best, best, res

    def spread():
        """seconds of every call of the last run(), their median and max / min"""
        v = sorted(each.get("last", []))
        return {"seconds_each": list(each.get("last", [])), "seconds_median": v[len(v) // 2] if v else None,
                "max_over_min": round(v[-1] / v[0], 3) if v and v[0] > 0 else None,
                # the container's CPU quota at work: milliseconds (summed over its threads) this cgroup was throttled during each call
                "cpu_throttled_ms_each": list(each.get("throttled", []))}

    def with_ingest(which, fn):
        prev = os.environ.get("MF_INGEST")
        os.environ["MF_INGEST"] = which
        try:
            return fn()
        finally:
            if prev is None:
                del os.environ["MF_INGEST"]
            else:
                os.environ["MF_INGEST"] = prev

    def plain_leg(f1, f2, prefix, n_reads, what):
        """plain FASTQ through the file boundary: the streaming device path (the file's bytes go up as they are) against the host pipeline
        (parse and pack on the host); roof: the text crosses PCIe once"""
        text_bytes = os.path.getsize(f1) + (os.path.getsize(f2) if f2 else 0)
        h_rate, h_secs, h_res = with_ingest("host", lambda: run(f1, f2, prefix + "_h1.fq", prefix + "_h2.fq" if f2 else None, n_reads, reps=2))
        h_spread = spread()
        d_rate, d_secs, d_res = with_ingest("device", lambda: run(f1, f2, prefix + "_d1.fq", prefix + "_d2.fq" if f2 else None, n_reads, reps=5, tag=what))
        d_spread = spread()
        ist = mf.last_ingest_stats()
        dflt_rate, dflt_secs, _ = run(f1, f2, prefix + "_x1.fq", prefix + "_x2.fq" if f2 else None, n_reads, reps=5)
        x_spread = spread()
        dflt_path = mf.last_ingest_stats()["path"]
        try:
            h2d = mf.h2d_bandwidth(0, 1 << 30, 3)
        except Exception:
            h2d = None
        same = md5_of(prefix + "_d1.fq") == md5_of(prefix + "_h1.fq") and (not f2 or md5_of(prefix + "_d2.fq") == md5_of(prefix + "_h2.fq")) and tuple(d_res) == tuple(h_res)
        return {"reads": n_reads, "text_bytes": text_bytes,
                "device_path": {"reads_per_s": d_rate, "seconds": round(d_secs, 4), **d_spread, "first_of_these_calls_seconds": first_call.get(what),
                                "ingest_path": "device" if ist["path"] == 1 else "host", "device_memory_in_use_peak_GB": ist["device_bytes_peak"] / 1e9},
                "host_pipeline": {"reads_per_s": h_rate, "seconds": round(h_secs, 4), **h_spread},
                "library_default": {"reads_per_s": dflt_rate, "seconds": round(dflt_secs, 4), **x_spread, "ingest_path": "device" if dflt_path == 1 else "host"},
                "roofline": {"bound": "pcie_h2d", "achieved": text_bytes / d_secs / 1e9, "peak": h2d, "unit": "GB/s", "frac": (text_bytes / d_secs / 1e9 / h2d) if h2d else None,
                             "note": "bytes of FASTQ text / seconds of the whole call on the device path (files in the page cache -> survivors written)"},
                "outputs_equal": bool(same), "kept": int(d_res[0]), "total": int(d_res[1])}
    if "small" in files:
        t = files["small"]
        out["pe_plain_reads_per_s"] = run(t + "_1.fq", t + "_2.fq", t + "_o1.fq", t + "_o2.fq", 2 * a.e2e_pairs)[0]
        out["se_gz_reads_per_s"] = run(t + "_1.fq.gz", None, t + "_og.fq", None, a.e2e_pairs, tag="se_gz")[0]
        out["se_gz_seconds"] = round(a.e2e_pairs / out["se_gz_reads_per_s"], 4)
        out["se_gz_first_call_seconds"] = first_call.get("se_gz")       # the process's first call of the device ingest path (HIP itself is up: the bait set has been built)
        if "small_cli" in files:
            out["se_gz_cli_cold"] = dict(files["small_cli"], note="`fastfilter bait` on the same file, process start to exit (HIP start-up included), run before this process touched the GPU")
        out["pairs"] = a.e2e_pairs
        out["note_small"] = "small inputs (the SE .gz has `pairs` reads): a call's fixed latencies dominate; configs4_se_gz is the throughput figure"
    if "full" in files:
        # configs[4] at its stated size: one gzip member of single-end reads, device ingest (inflate, line index, 2-bit pack, filter
        # and survivor copy on the GPU).  Checked against the host pipeline on the plain text of the same reads, byte for byte.
        t = files["full"]
        n = a.e2e_full_reads
        prev = os.environ.get("MF_INGEST")
        os.environ["MF_INGEST"] = "host"
        host_rate, _, host_res = run(t + "_1.fq", None, t + "_oh.fq", None, n, reps=1)
        host_gz_rate = run(t + "_1.fq.gz", None, t + "_ohg.fq", None, n, reps=1)[0]
        if prev is None:
            del os.environ["MF_INGEST"]
        else:
            os.environ["MF_INGEST"] = prev
        rate, secs, res = run(t + "_1.fq.gz", None, t + "_od.fq", None, n, reps=6, tag="configs4")          # (the first is the process's first call at this size)
        c4_spread = spread()
        warm = sorted(c4_spread["seconds_each"][1:])
        c4_spread["warm_calls_max_over_min"] = round(warm[-1] / warm[0], 3) if warm and warm[0] > 0 else None
        ist = mf.last_ingest_stats()                 # (of the last of the calls)
        md5 = md5_of(t + "_od.fq")
        gz_bytes = os.path.getsize(t + "_1.fq.gz")
        try:
            h2d = mf.h2d_bandwidth(0, 1 << 30, 3)
        except Exception:
            h2d = None
        out["configs4_se_gz"] = {
            # the leg's roof: its input crosses PCIe once, as it lies on disk -- compressed bytes over the time of the whole call, against a
            # pinned host-to-device copy of 1 GiB measured on this box in this process
            "roofline": {"bound": "pcie_h2d", "achieved": gz_bytes / secs / 1e9, "peak": h2d, "unit": "GB/s", "frac": (gz_bytes / secs / 1e9 / h2d) if h2d else None,
                         "note": "compressed input bytes / seconds of the whole call (file in the page cache -> survivors written)"},
            "inflate_kernels": {"busy_seconds": ist["decode_busy_seconds"], "text_GB_per_s": (ist["text_bytes"] / ist["decode_busy_seconds"] / 1e9) if ist["decode_busy_seconds"] > 0 else None,
                                "note": "time with at least one gz_decode kernel running (HIP events around every launch), while upload, link step and the consumers' kernels share the device",
                                "bound": "instruction issue and latency of divergent lane code, not memory: 6.6 wave-instructions per byte of text (VALU 3.5, scalar 2.7, LDS 0.26), wavefronts on "
                                         "s_waitcnt 56 % of their cycles, ~2.7 B of HBM traffic per byte of text (profiles/r04/c_gzdev_pmc.txt); alone with the chip full the kernel does "
                                         "88.5 GB/s of text (profiles/r05/h_gzdev_check_kernel_stats.csv: 43.5 ms for 3.85 GB)"},
            "ingest_path": "device" if ist["path"] == 1 else "host", "device_memory_in_use_peak_GB": ist["device_bytes_peak"] / 1e9, "call_buffers_peak_GB": ist["pool_bytes_peak"] / 1e9,
            "chunks": ist["chunks"], "chunks_linked": ist["chunks_linked"], "gaps_bridged_on_host": ist["gaps"],
            "reads": n, "reads_per_s": rate, "seconds": round(secs, 4), **c4_spread, "first_call_seconds": first_call.get("configs4"), "kept": int(res[0]), "total": int(res[1]),
            # the same file through the boundary the reference has: a process per call (HIP start-up, streams, pinned staging, the bait set's build
            # and the call), process start to exit, run before this process touched the GPU
            "cli_cold": dict(files["full_cli"], reads_per_s=n / files["full_cli"]["seconds"], kept_equals_library=bool(files["full_cli"]["kept"] == int(res[0]))) if "full_cli" in files else files.get("cli_error"),
            "gz_bytes": gz_bytes, "text_bytes": os.path.getsize(t + "_1.fq"),
            "input": "synthetic single-end FASTQ (tools/make_fastq.py), ONE gzip member written by tools/pgzip.py at level 6 (8 MiB slices)",
            "ingest": "device: compressed bytes uploaded as they are; inflate, line index, 2-bit pack, filter, survivor copy on the GPU",
            "output_md5": md5, "output_equals_host_pipeline_on_plain_text": bool(md5 == md5_of(t + "_oh.fq") and tuple(res) == tuple(host_res)),
            "host_pipeline_plain_reads_per_s": host_rate, "host_pipeline_gz_reads_per_s": host_gz_rate,
            "prep_seconds": files.get("full_prep_seconds")}
        try:          # the plain text of the same reads: single-end, 10.7 GB
            out["configs4_se_plain"] = plain_leg(t + "_1.fq", None, t + "_p", n, "se_plain")
        except Exception as e:
            out["configs4_se_plain"] = {"error": str(e)[:200]}
    elif "full_note" in files:
        out["configs4_se_gz"] = {"skipped": files["full_note"]}
    if "plain" in files:
        try:
            c = files["plain"]
            out["configs1_pe_plain"] = plain_leg(c + "_1.fq", c + "_2.fq", c + "_p", 2 * a.plain_pairs, "pe_plain")
        except Exception as e:
            out["configs1_pe_plain"] = {"error": str(e)[:200]}
    elif "plain_note" in files:
        out["configs1_pe_plain"] = {"skipped": files["plain_note"]}
    if "real" in files:
        # an eighth of configs[4], compressed by gzip / pigz / bgzip themselves (not by this repository's tools/pgzip.py): the same
        # call, checked against the host pipeline on the plain text of the same reads
        r = files["real"]
        n = a.real_gz_reads
        rks = mf.KmerSet.from_fasta(r["bait"], a.k)
        prev = os.environ.get("MF_INGEST")
        os.environ["MF_INGEST"] = "host"
        _, _, host_res = run2(mf, rks, r["plain"], r["plain"] + ".host.out", 1)
        if prev is None:
            del os.environ["MF_INGEST"]
        else:
            os.environ["MF_INGEST"] = prev
        host_md5 = md5_of(r["plain"] + ".host.out")
        legs = {}
        for name, path in sorted(r["gz"].items()):
            secs, _, res = run2(mf, rks, path, path + ".out", 3)
            ist = mf.last_ingest_stats()
            legs[name] = {"reads_per_s": n / secs, "seconds": round(secs, 4), "gz_bytes": os.path.getsize(path), "ingest_path": "device" if ist["path"] == 1 else "host (BGZF members are decoded side by side on the host)",
                          "output_equals_host_pipeline_on_plain_text": bool(md5_of(path + ".out") == host_md5 and tuple(res) == tuple(host_res))}
        out["real_compressors"] = {"reads": n, "files": legs}
    return out


def filter_v2_leg(mf, files, a):
    """The reference's FASTQ quality filter (filter/filter_v2, `-d`) on a .gz pair: the drop-in CLI from process start to exit (timed before
    this process touched the GPU) and the library call inside this process, device ingest path against host pipeline, outputs compared."""
    f = files["fv2"]
    q, n = f["prefix"], 2 * a.fv2_pairs
    out = {"pairs": a.fv2_pairs, "argv": "-1 a_1.fq.gz -2 a_2.fq.gz -3 o_1.fq -4 o_2.fq -d", "gz_bytes": os.path.getsize(q + "_1.fq.gz") + os.path.getsize(q + "_2.fq.gz"),
           "text_bytes": os.path.getsize(q + "_1.fq") + os.path.getsize(q + "_2.fq"),
           "cli_process_start_to_exit": {k: {"seconds": round(v, 4), "reads_per_s": n / v} for k, v in f["cli_seconds"].items()},
           "cli_outputs_equal": bool(md5_of(q + "device_c1.fq") == md5_of(q + "host_c1.fq") and md5_of(q + "device_c2.fq") == md5_of(q + "host_c2.fq"))}
    legs = {}
    for tag, reps in (("device", 3), ("host", 1)):
        prev = os.environ.get("MF_QUAL_INGEST")
        if tag == "host":
            os.environ["MF_QUAL_INGEST"] = "host"
        try:
            best, res = 1e9, None
            for _ in range(reps):
                for o in ("_l1.fq", "_l2.fq"):
                    if os.path.exists(q + tag + o):
                        os.unlink(q + tag + o)
                t0 = time.perf_counter()
                res = mf.qualfilter_files(q + "_1.fq.gz", q + "_2.fq.gz", q + tag + "_l1.fq", q + tag + "_l2.fq", dedup=True)
                best = min(best, time.perf_counter() - t0)
            legs[tag] = {"seconds": round(best, 4), "reads_per_s": n / best, "kept_pairs": int(res[0])}
            if tag == "device":
                ist = mf.last_ingest_stats()
                legs[tag].update({"ingest_path": "device" if ist["path"] == 1 else "host", "device_memory_in_use_peak_GB": ist["device_bytes_peak"] / 1e9,
                                  "output_bytes": os.path.getsize(q + tag + "_l1.fq") + os.path.getsize(q + tag + "_l2.fq")})
        finally:
            if prev is None:
                os.environ.pop("MF_QUAL_INGEST", None)
            else:
                os.environ["MF_QUAL_INGEST"] = prev
    out["library_call"] = legs
    out["library_outputs_equal"] = bool(md5_of(q + "device_l1.fq") == md5_of(q + "host_l1.fq") and md5_of(q + "device_l2.fq") == md5_of(q + "host_l2.fq")
                                        and md5_of(q + "device_l1.fq") == md5_of(q + "device_c1.fq"))
    out["note"] = ("nearly every record is written back: the call moves as many bytes down and out as it takes in -- its roof is the page cache taking the two output files "
                   "(one writer thread per file, 8-10 GB/s each), not the device")
    return out


def run2(mf, ks, f1, o1, reps):
    """best seconds of `reps` single-end file-to-file calls, rate placeholder, (kept, total)"""
    best, res = 1e9, None
    for _ in range(reps):
        t0 = time.perf_counter()
        res = mf.filter_fastq_files(ks, f1, None, o1, None)
        best = min(best, time.perf_counter() - t0)
    return best, None, res


def group_a(files):
    """the contig filter CLI (the reference's real `fastfilter`) on the 1 M-record generator file."""
    exe = os.path.join(ROOT, "mitoflex_amd", "assemble", "fastfilter")
    src = files["g20"]
    dst = src + ".filtered"
    best, count = 1e9, None
    for _ in range(3):
        t0 = time.perf_counter()
        count = int(subprocess.check_output([exe, "-i", src, "-o", dst, "-l", "0,20000", "-d", "10"], env=child_env()).decode())
        best = min(best, time.perf_counter() - t0)
    size = os.path.getsize(src)
    return {"seconds": round(best, 4), "MB_per_s": round(size / best / 1e6, 1), "input_bytes": size, "count": count,
            "count_matches_reference": count == G20_COUNT, "output_md5_matches_reference": md5_of(dst) == G20_MD5,
            "reference_elf_seconds_build_container": 0.571,
            "note": "argv: -l 0,20000 -d 10 (SURVEY.md 8c case G20); the reference figure was taken with its prebuilt ELF on one core of the build "
                    "container (the ELF cannot travel to this box), so the two are not same-host timings"}


VALU_PEAK_GINST = 1024 * 2.4 / 4        # wave-instructions per ns the chip can issue: 256 CUs x 4 SIMDs, one per 4 cycles per wave stream at 2.4 GHz

GATHER_PEAK_GLOOKUPS = 265.0     # G lookups/s: independent random 16-byte gathers from a table of <= 4 MiB (an XCD's L2), every CU issuing, nothing else
                                 # running -- measured by tools/gather_roof.hip (profiles/r06/a_gather_roof.txt); 217 G/s beside a 16-byte read stream


def committed_valu(kernel_tag, profile):
    """SQ_INSTS_VALU per launch of a kernel from a committed PMC profile of this round (profiles/r06/pmc_<profile>.txt, tools/pmc_any.sh):
    the count is a property of the code and the input, so the committed figure prices a live duration.  None when absent."""
    import ast
    import re
    try:
        for ln in open(os.path.join(ROOT, "profiles", "r06", "pmc_%s.txt" % profile)):
            if kernel_tag in ln and "SQ_INSTS_VALU" in ln:
                d = ast.literal_eval(re.search(r"\{.*\}", ln).group(0))
                return float(d["SQ_INSTS_VALU"])
    except Exception:
        pass
    return None


def oracle_window(bait, k, thr, host_words, host_npos, n_win, offsets=None):
    """pass bits of the first n_win reads from the CPU oracle (all host threads)"""
    import numpy as np
    from oracle import oracle_lib as ol
    off = np.arange(n_win + 1, dtype=np.uint64) * READ_LEN if offsets is None else offsets[:n_win + 1]
    lim = int(off[n_win])
    R = ol.OracleReads.from_arrays(host_words[:(lim + 15) // 16 + 8], off, host_npos[host_npos < lim])
    return ol.filter_reads(ol.OracleTable(bait, k), R, thr, threads=os.cpu_count() or 1)[0]


def axis_steps(a):
    """passes per timed call of the legs off the headline's point: the headline's own K (a call of K pipelined passes pays one ramp and one
    un-overlapped tail -- ~0.3 ms -- whatever K is: at K = 10 that was 10-15 % of a leg's figure and none of the headline's at K = 50)"""
    return max(2, min(int(a.steps), 50))


def timed_passes(mf, ks, reads, thr, dev, n_st, mode=None):
    """(seconds per pipelined pass over n_st passes, stats of the timed call, stats of a fully sampled loop)"""
    mode = mf.MODE_SCREENED if mode is None else mode
    os.environ["MF_EVENT_STRIDE"] = "1000000000"
    for _ in range(3):                                   # (the pass kind follows what the last calls of this read set saw)
        mf.filter_resident(ks, reads, thr, mode, 3)
    mf.device_synchronize(dev)
    c0 = time.perf_counter()
    st = mf.filter_resident(ks, reads, thr, mode, n_st)
    mf.device_synchronize(dev)
    dt = (time.perf_counter() - c0) / n_st
    os.environ["MF_EVENT_STRIDE"] = "1"
    sp = mf.filter_resident(ks, reads, thr, mode, n_st)
    os.environ["MF_EVENT_STRIDE"] = "8"
    return dt, st, sp


def superset_bait(base, size, seed):
    """the benchmark's bait plus random records up to `size` bases in all: the reads planted from the benchmark's bait stay bait reads"""
    from mitoflex_amd.utility.synth_bait import bait_records, random_bait
    have = sum(len(r) for r in bait_records(base))
    return base if size <= have else base + random_bait(size - have, seed=seed)


def bait_sweep_leg(mf, reads, base_bait, a, dev, alg_bytes, n_words):
    """extra.bait_sweep: the same resident reads against baits of 33 kbp .. 8.5 Mbp (the benchmark's bait plus random genomes; the last is
    the size of the reference's profile/MT_database read as nucleotides).  Which screen a bait takes follows its size (front_mode); every
    leg carries the roofline of what actually bounds it and a check of its first 1.5 M pass bits against the CPU oracle."""
    import numpy as np
    out = {}
    n_win = min(1_500_000, a.reads) // 32 * 32
    for size in (33_000, 100_000, 350_000, 1_000_000, 8_500_000):
        bait = superset_bait(base_bait, size, seed=size)
        t0 = time.perf_counter()
        ks = mf.KmerSet.from_text(bait, a.k, dev)
        t_build = time.perf_counter() - t0
        inf = ks.info
        n_st = axis_steps(a)
        dt, st, sp = timed_passes(mf, ks, reads, THRESHOLD, dev, n_st)
        scr_s = sp.ms_screen / 1e3
        whole = alg_bytes / dt / 1e9 / HBM_PEAK_GBPS
        hbm = {"achieved": alg_bytes / scr_s / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": alg_bytes / scr_s / 1e9 / HBM_PEAK_GBPS}
        if inf.front_mode in (2, 4):
            look = n_words / scr_s / 1e9              # one 16-base sample per 32-bit word of the stream, each looked up in front2 (mode 4: or answered by the one-bit LDS table)
            roof = {"bound": "l2_gather", "achieved": look, "peak": GATHER_PEAK_GLOOKUPS, "unit": "G samples/s against G lookups/s", "frac": look / GATHER_PEAK_GLOOKUPS,
                    "kernel": "screen2_kernel (front2 only)" if inf.front_mode == 2 else "screen2_kernel (one-bit LDS table, then front2: the LDS table answers 1/2 - 2/3 of the samples itself, so `frac` can pass the look-up rate's share)",
                    "lookups_per_launch": int(n_words), "hbm": hbm,
                    "whole_pass_frac": n_words / dt / 1e9 / GATHER_PEAK_GLOOKUPS,
                    "whole_pass_note": "samples a second of wall time over the gather roof: consecutive screens run on two streams in turn and overlap, so a launch lasts longer than a pass takes",
                    "peak_source": "tools/gather_roof.hip, profiles/r06/a_gather_roof.txt: random 16-byte gathers from <= 4 MiB, nothing else running (217 G/s beside a read stream)",
                    "note": "front3 look-ups of front2's survivors (baits of several Mbp: a table beyond L2, 55-80 G lookups/s) are not counted in `achieved`"}
        elif inf.front_mode == 1:
            v = committed_valu("screen2_kernel", "s2_100k")
            g = v / scr_s / 1e9 if v else None
            roof = {"bound": "valu", "achieved": g, "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s", "frac": g / VALU_PEAK_GINST if g else None,
                    "kernel": "screen2_kernel (LDS table + front2)", "hbm": hbm,
                    "valu_source": "profiles/r06/pmc_s2_100k.txt (SQ_INSTS_VALU per launch at 100 kbp; the count moves with the bait's size)"}
        else:
            roof = {"bound": "hbm", **hbm, "kernel": "screen3_kernel" if inf.front_mode == 3 else "screen_kernel"}
        roof.update({"avg_kernel_ms": sp.ms_screen, "kernel_launches_averaged": n_st, "whole_pass_frac_of_hbm_peak": whole})
        leg = {"bait_bases": size, "front_mode": int(inf.front_mode), "front2_MiB": (16 << inf.front2_log2_blocks) / 2**20 if inf.front2_log2_blocks else 0,
               "front3_MiB": (16 << inf.front3_log2_blocks) / 2**20 if inf.front3_log2_blocks else 0, "n_keys": int(inf.n_keys), "n_smers": int(inf.n_smers),
               "set_build_seconds": round(t_build, 3), "reads_per_s": a.reads / dt, "ms_per_step": dt * 1e3, "steps": n_st,
               "ms_screen_kernel": round(sp.ms_screen, 4), "work_items_per_read": st.n_candidates / a.reads, "passed": int(st.n_pass),
               "whole_pass_frac_of_hbm_peak": round(whole, 4), "roofline": roof}
        if reads.host_words is not None and n_win:
            gbits, _, _ = mf.filter_reads(ks, reads, THRESHOLD, mf.MODE_SCREENED)
            obits = oracle_window(bait, a.k, THRESHOLD, reads.host_words, reads.host_npos, n_win)
            leg["window_bits_match_oracle"] = bool(np.array_equal(gbits[:n_win // 32], obits[:n_win // 32]))
            leg["window_reads_checked"] = n_win
        out[str(size)] = leg
        ks.close()
    return out


def threshold_leg(mf, ks, reads, bait, a, dev, alg_bytes):
    """extra.threshold_sweep: the passes the headline does not take -- threshold 2 and 7 (screen + mark + exact on the candidates) and the
    hit-count form (every k-mer of every candidate verified) -- on the same resident reads; the exhaustive pass (the literal north-star
    kernel on every read) priced against the vector issue rate."""
    import numpy as np
    out = {}
    n_win = min(1_500_000, a.reads) // 32 * 32
    for thr in (1, 2, 7):
        dt, st, sp = timed_passes(mf, ks, reads, thr, dev, axis_steps(a))
        leg = {"ms_per_step": dt * 1e3, "steps": axis_steps(a), "reads_per_s": a.reads / dt, "passed": int(st.n_pass), "candidates_or_work_items": int(st.n_candidates),
               "ms_screen_kernel": round(sp.ms_screen, 4), "ms_mark_kernel": round(sp.ms_mark, 4), "ms_last_kernel": round(sp.ms_exact, 4),
               "whole_pass_frac_of_hbm_peak": round(alg_bytes / dt / 1e9 / HBM_PEAK_GBPS, 4)}
        _, hits, sh = mf.filter_reads(ks, reads, thr, mf.MODE_SCREENED, want_hits=True)          # hit counts: the count-all exact kernel
        _, _, sh = mf.filter_reads(ks, reads, thr, mf.MODE_SCREENED, want_hits=True)
        leg["with_hit_counts"] = {"ms_per_step": sh.ms_total, "reads_per_s": a.reads / (sh.ms_total / 1e3), "ms_exact_kernel": round(sh.ms_exact, 4)}
        if reads.host_words is not None and n_win and thr > 1:
            gbits, _, _ = mf.filter_reads(ks, reads, thr, mf.MODE_SCREENED)
            obits = oracle_window(bait, a.k, thr, reads.host_words, reads.host_npos, n_win)
            leg["window_bits_match_oracle"] = bool(np.array_equal(gbits[:n_win // 32], obits[:n_win // 32]))
        del hits
        out[str(thr)] = leg
    ex = mf.filter_resident(ks, reads, THRESHOLD, mf.MODE_EXHAUSTIVE, 1)
    os.environ["MF_EVENT_STRIDE"] = "1"
    ex = mf.filter_resident(ks, reads, THRESHOLD, mf.MODE_EXHAUSTIVE, 2)
    os.environ["MF_EVENT_STRIDE"] = "8"
    v = committed_valu("exact_kernel", "exhaustive")
    ex_s = ex.ms_exact / 1e3
    g = v / ex_s / 1e9 if (v and ex_s > 0) else None
    out["exhaustive"] = {"ms_per_step": ex.ms_total, "reads_per_s": a.reads / (ex.ms_total / 1e3), "ms_exact_kernel": round(ex.ms_exact, 4),
                         "roofline": {"bound": "valu", "achieved": g, "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s", "frac": g / VALU_PEAK_GINST if g else None,
                                      "kernel": "exact_kernel (every read: extract -> canonicalise -> LDS bit table -> open-address probe -> threshold)",
                                      "valu_wave_instructions_per_launch": v, "valu_source": "profiles/r06/pmc_exhaustive.txt (SQ_INSTS_VALU per launch, same reads, same bait)",
                                      "hbm_frac": alg_bytes / ex_s / 1e9 / HBM_PEAK_GBPS if ex_s > 0 else None}}
    return out


def ragged_leg(mf, ks, reads, bait, a, dev, uniform_ms):
    """extra.ragged: the same dense stream cut into reads of 60..150 bases (what the quality filter's cuts leave,
    filter/filter_bin/src/main.rs:239-268): the offsets path (a read is found by binary search instead of a multiplication)."""
    import numpy as np
    from mitoflex_amd.utility.synth_bait import ragged_offsets
    total = a.reads * READ_LEN
    off = ragged_offsets(total)
    n = len(off) - 1
    rr = mf.Reads.from_packed(reads.host_words, off, reads.host_npos, dev)
    try:
        dt, st, sp = timed_passes(mf, ks, rr, THRESHOLD, dev, axis_steps(a))
        alg = st.algorithmic_bytes
        leg = {"reads": n, "mean_length": total / n, "ms_per_step": dt * 1e3, "steps": axis_steps(a), "reads_per_s": n / dt, "bases_per_s": total / dt, "passed": int(st.n_pass),
               "ms_screen_kernel": round(sp.ms_screen, 4), "ms_last_kernel": round(sp.ms_exact, 4), "work_items": int(st.n_candidates),
               "whole_pass_frac_of_hbm_peak": round(alg / dt / 1e9 / HBM_PEAK_GBPS, 4), "ms_per_step_over_uniform": round(dt * 1e3 / uniform_ms, 3),
               "roofline": {"bound": "hbm", "achieved": alg / (sp.ms_screen / 1e3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": alg / (sp.ms_screen / 1e3) / 1e9 / HBM_PEAK_GBPS, "kernel": "screen_kernel", "avg_kernel_ms": sp.ms_screen}}
        n_win = min(1_500_000, n) // 32 * 32
        gbits, _, _ = mf.filter_reads(ks, rr, THRESHOLD, mf.MODE_SCREENED)
        obits = oracle_window(bait, a.k, THRESHOLD, reads.host_words, reads.host_npos, n_win, offsets=off)
        leg["window_bits_match_oracle"] = bool(np.array_equal(gbits[:n_win // 32], obits[:n_win // 32]))
        leg["window_reads_checked"] = n_win
        return leg
    finally:
        rr.close()


def realistic_leg(mf, a, dev, uniform_ms):
    """extra.realistic: an AT-rich bait with poly-T / poly-A runs and a (TA)n stretch, a background with 2 % microsatellite reads and
    0.1 % NUMT-like reads (bait fragments at 15 % divergence) beside the 0.5 % bait reads: what low-complexity s-mers shared between bait and
    background cost (every one of them is a true stage-1 positive that the finish kernels settle exactly)."""
    import numpy as np
    from mitoflex_amd.utility.synth_bait import realistic_bait
    bait = realistic_bait()
    ks = mf.KmerSet.from_text(bait, a.k, dev)
    rd = mf.Reads.synth(a.reads, READ_LEN, seed=20261004, bait_text=bait, mito_ppm=5000, sub_ppm=10000, n_read_ppm=10000, n_base_ppm=1000,
                        device=dev, keep_host=True, msat_ppm=20000, numt_ppm=1000, numt_div_ppm=150000)
    try:
        dt, st, sp = timed_passes(mf, ks, rd, THRESHOLD, dev, axis_steps(a))
        alg = st.algorithmic_bytes
        leg = {"bait": "16 569 bp, 68 % AT, poly-T(40) / poly-A(35) runs, (TA)60 (mitoflex_amd/utility/synth_bait.realistic_bait)",
               "background": "2 % microsatellite reads (motif of 1..6 bases), 0.1 % NUMT-like reads (15 % divergence), 0.5 % bait reads, N in 1 % of reads",
               "ms_per_step": dt * 1e3, "steps": axis_steps(a), "reads_per_s": a.reads / dt, "passed": int(st.n_pass), "work_items_per_read": st.n_candidates / a.reads,
               "ms_screen_kernel": round(sp.ms_screen, 4), "ms_last_kernel": round(sp.ms_exact, 4),
               "whole_pass_frac_of_hbm_peak": round(alg / dt / 1e9 / HBM_PEAK_GBPS, 4), "ms_per_step_over_iid": round(dt * 1e3 / uniform_ms, 3),
               "roofline": {"bound": "hbm", "achieved": alg / (sp.ms_screen / 1e3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": alg / (sp.ms_screen / 1e3) / 1e9 / HBM_PEAK_GBPS, "kernel": "screen_kernel", "avg_kernel_ms": sp.ms_screen}}
        n_win = min(1_500_000, a.reads) // 32 * 32
        gbits, _, _ = mf.filter_reads(ks, rd, THRESHOLD, mf.MODE_SCREENED)
        obits = oracle_window(bait, a.k, THRESHOLD, rd.host_words, rd.host_npos, n_win)
        leg["window_bits_match_oracle"] = bool(np.array_equal(gbits[:n_win // 32], obits[:n_win // 32]))
        leg["window_reads_checked"] = n_win
        return leg
    finally:
        rd.close()
        ks.close()



def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world
    if a.reads <= 0:
        a.reads = READS_50GBP_8 if world == 8 else READS_5GBP
    rdv = ShmRendezvous(rank, world) if world > 1 else None
    solo = rank == 0 and world == 1

    # ---- everything that needs child processes happens before this process touches the GPU
    import tempfile
    tmpdir = tempfile.TemporaryDirectory(prefix="mf_bench_", dir="/tmp") if rank == 0 else None
    if rank == 0 and not solo:
        a.e2e_full_reads, a.no_group_a = 0, True     # with several ranks only the small paired set is made (file level over all devices)
    files = prepare_inputs(a, tmpdir.name, solo) if rank == 0 else {}
    # N > 1: a .gz shard of its own for every rank (file level on its own GPU, beside the resident rate), made by child processes now
    rank_tmp, rank_gz = None, None
    if world > 1 and a.rank_file_reads > 0 and not being_profiled():
        try:
            rank_tmp = tempfile.TemporaryDirectory(prefix="mf_bench_r%d_" % rank, dir="/tmp")
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_fastq.py"), os.path.join(rank_tmp.name, "k"), "--pairs", str(a.rank_file_reads),
                                   "--mates", "1", "--seed", str(777 + rank)], env=child_env(), stdout=subprocess.DEVNULL)
            with open(os.path.join(rank_tmp.name, "k_1.fq.gz"), "wb") as g:
                subprocess.check_call(["gzip", "-1", "-c", os.path.join(rank_tmp.name, "k_1.fq")], stdout=g, env=child_env())
            rank_gz = os.path.join(rank_tmp.name, "k_1.fq.gz")
        except Exception:
            rank_gz = None
        os.environ.setdefault("MF_DEVPOOL_GB", "4")       # (ranks of one node: what a process keeps between calls stays small)
    live, valu = (None, None), (None, None)
    default_set = a.reads == READS_5GBP
    if solo and not a.no_live_traffic and default_set:
        live = live_traffic(a.k)
        if a.k < 28 and live[0] is not None:                  # the stride-8 screen is bound by vector issue, not by HBM: count its instructions too
            valu = live_counters(("SQ_INSTS_VALU",), a.k)
    # configs[2]: the stride-8 legs of the k sweep get their own live instruction count (one more short child run each)
    sweep_valu = {}
    if solo and not a.no_live_traffic and default_set and a.k_sweep and a.k_sweep != "none" and live[0] is not None:
        for kk in [int(x) for x in a.k_sweep.replace('"', "").split(",") if x.strip().isdigit()]:
            if kk < 28 and kk != a.k:
                sweep_valu[kk] = live_counters(("SQ_INSTS_VALU",), kk)

    # every rank packs its own synthetic shard on the host first: share the host's threads between the ranks of the node
    if local_world > 1 and "MF_HOST_THREADS" not in os.environ:
        os.environ["MF_HOST_THREADS"] = str(max(1, (os.cpu_count() or 1) // local_world))
    from mitoflex_amd import mitofilter as mf
    mf.load()
    from mitoflex_amd.utility.synth_bait import make_bait
    bait = make_bait()
    n_dev = mf.device_count()
    if local_world > max(1, n_dev) and os.environ.get("MF_BENCH_SHARE_GPU") != "1":
        sys.exit(f"bench.py: {local_world} ranks on this node but {n_dev} visible GPU(s): one rank per GPU "
                 "(MF_BENCH_SHARE_GPU=1 lets ranks share a device -- tests only, the figures mean nothing then)")
    dev = local_rank % max(1, n_dev)
    ks = mf.KmerSet.from_text(bait, a.k, dev)
    want_cpu = solo and a.cpu_sample > 0
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
    mine = time.perf_counter() - t0
    elapsed = mine
    per_rank = [mine]
    if rdv is not None:
        elapsed = rdv.allmax(mine)
        per_rank = rdv.gather(mine)

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
             # configs[3]: "per-GPU + aggregate reads/s" -- every rank's own rate over its own K steps (the aggregate uses the slowest)
             "per_gpu_reads_per_s": [a.reads * a.steps / t for t in per_rank],
             "per_rank_ms_per_step": [round(t / a.steps * 1e3, 4) for t in per_rank],
             "pipelined": "consecutive steps overlap: finish kernels of step i run under the screen kernel of step i+1 (second stream) and "
                          "consecutive screen kernels go to two streams in turn, so the duration of a screen launch (ms_screen_kernel, "
                          "roofline.achieved) includes time it shares the device with its neighbours; a step completes every ms_per_step"}
    alone_frac = None
    if solo:
        one = mf.filter_resident(ks, reads, THRESHOLD, mf.MODE_SCREENED, 1)    # a single step: nothing to overlap with
        extra["ms_single_pass_latency"] = round(one.ms_total, 4)
        # the same kernel with the device to itself (option pass=serial: one stream, no overlap between steps), every dispatch sampled
        mf.set_option("pass", "serial"); os.environ["MF_EVENT_STRIDE"] = "1"
        alone = mf.filter_resident(ks, reads, THRESHOLD, mf.MODE_SCREENED, min(a.steps, 20))
        os.environ["MF_EVENT_STRIDE"] = "8"
        mf.set_option("pass", os.environ.get("MF_PASS", "default") if os.environ.get("MF_ENV_KNOBS") == "1" else "default")
        if alone.ms_screen > 0:
            alone_frac = alg_bytes / (alone.ms_screen / 1e3) / 1e9 / HBM_PEAK_GBPS
            extra["screen_kernel_alone"] = {"ms": round(alone.ms_screen, 4), "frac_of_hbm_peak": round(alone_frac, 4),
                                            "ms_per_step_serial": round(alone.ms_total, 4)}
    whole_pass_frac = alg_bytes / (elapsed / a.steps) / 1e9 / HBM_PEAK_GBPS
    extra["whole_pass_frac_of_hbm_peak"] = round(whole_pass_frac, 4)
    if rank == 0 and not a.no_exhaustive:
        ex = mf.filter_resident(ks, reads, THRESHOLD, mf.MODE_EXHAUSTIVE, 1)
        extra["exhaustive_reads_per_s"] = a.reads / (ex.ms_total / 1e3)
        extra["exhaustive_passed"] = int(ex.n_pass)

    if solo and a.k_sweep and a.k_sweep != "none":
        # configs[2]: the other k on the same resident reads (a few steps each: set built on the device, warm-up, timed loop, sampled loop)
        sweep = {}
        for kk in [int(x) for x in a.k_sweep.replace('"', "").split(",") if x.strip().isdigit()]:
            if kk == a.k:
                continue
            try:
                ks2 = mf.KmerSet.from_text(bait, kk, dev)
                n_st = axis_steps(a)
                os.environ["MF_EVENT_STRIDE"] = "1000000000"
                mf.filter_resident(ks2, reads, THRESHOLD, mf.MODE_SCREENED, 10)
                mf.device_synchronize(dev)
                c0 = time.perf_counter()
                s2 = mf.filter_resident(ks2, reads, THRESHOLD, mf.MODE_SCREENED, n_st)
                mf.device_synchronize(dev)
                dt = time.perf_counter() - c0
                os.environ["MF_EVENT_STRIDE"] = "1"
                sp2 = mf.filter_resident(ks2, reads, THRESHOLD, mf.MODE_SCREENED, n_st)
                os.environ["MF_EVENT_STRIDE"] = "8"
                scr_s = sp2.ms_screen / 1e3
                hbm2 = {"achieved": alg_bytes / scr_s / 1e9 if scr_s > 0 else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": alg_bytes / scr_s / 1e9 / HBM_PEAK_GBPS if scr_s > 0 else None}
                if kk >= 28:
                    roof2 = {"bound": "hbm", **hbm2, "kernel": "screen_kernel", "avg_kernel_ms": sp2.ms_screen, "kernel_launches_averaged": n_st,
                             "whole_pass_frac": alg_bytes / (dt / n_st) / 1e9 / HBM_PEAK_GBPS}
                    if kk >= 33:
                        roof2["note"] = ("two-word keys rotate through three buffer sets: two or three screen launches are in flight at a time, so a launch lasts "
                                         "longer than a pass takes and `frac` (per launch) understates the rate -- `whole_pass_frac` is the pass")
                else:
                    v, why = sweep_valu.get(kk, (None, "not collected"))
                    ginst = v["SQ_INSTS_VALU"] / scr_s / 1e9 if (v and scr_s > 0) else None
                    roof2 = {"bound": "valu", "achieved": ginst, "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s", "frac": ginst / VALU_PEAK_GINST if ginst else None,
                             "kernel": "screen_kernel (stride-8 geometry)", "avg_kernel_ms": sp2.ms_screen, "kernel_launches_averaged": n_st, "hbm": hbm2,
                             "valu_wave_instructions_per_launch": v["SQ_INSTS_VALU"] if v else None,
                             "valu_source": "live: rocprofv3 --pmc SQ_INSTS_VALU child run of this script at this k on this box" if v else f"no live reading ({why})",
                             "whole_pass_frac_of_hbm_peak": alg_bytes / (dt / n_st) / 1e9 / HBM_PEAK_GBPS}
                sweep[str(kk)] = {"reads_per_s": a.reads * n_st / dt, "ms_per_step": dt / n_st * 1e3, "steps": n_st, "passed": int(s2.n_pass),
                                  "ms_screen_kernel": round(sp2.ms_screen, 4), "whole_pass_frac_of_hbm_peak": round(alg_bytes / (dt / n_st) / 1e9 / HBM_PEAK_GBPS, 4),
                                  "bound": "valu (stride-8 screen)" if kk < 28 else "hbm", "roofline": roof2}
                ks2.close()
            except Exception as e:
                sweep[str(kk)] = {"error": str(e)[:160]}
        extra["k_sweep"] = sweep

    # ---- the axes the headline does not move along (one bait size, threshold 1, uniform reads, iid background): a leg that raises ends the
    # run with a non-zero exit instead of hiding in `extra`
    def leg(name, fn):
        try:
            extra[name] = fn()
        except Exception as e:
            import traceback
            traceback.print_exc()
            extra[name] = {"error": str(e)[:200]}
    if solo and default_set and not a.no_axes:
        n_words_stream = (a.reads * READ_LEN + 15) // 16
        leg("bait_sweep", lambda: bait_sweep_leg(mf, reads, bait, a, dev, alg_bytes, n_words_stream))
        leg("threshold_sweep", lambda: threshold_leg(mf, ks, reads, bait, a, dev, alg_bytes))
        if reads.host_words is not None:
            # (against the uniform set timed the same way: ten pipelined passes behind three warm-up calls -- threshold_sweep's T = 1 leg)
            uni = extra.get("threshold_sweep", {}).get("1", {}).get("ms_per_step") or elapsed / a.steps * 1e3
            leg("ragged", lambda: ragged_leg(mf, ks, reads, bait, a, dev, uni))

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
    if solo:
        reads.close()                                    # (the file-level runs want the device memory)
        if default_set and not a.no_axes:
            leg("realistic", lambda: realistic_leg(mf, a, dev, extra.get("threshold_sweep", {}).get("1", {}).get("ms_per_step") or elapsed / a.steps * 1e3))
        if "skipped" in files or "error" in files:
            extra["e2e_files"] = {k: v for k, v in files.items() if k in ("skipped", "error")}
        else:
            try:
                extra["e2e_files"] = e2e_files(mf, ks, files, a)
            except Exception as e:           # the headline numbers do not depend on scratch space for files
                extra["e2e_files"] = {"error": str(e)[:200]}
            if "fv2" in files:
                try:
                    extra["filter_v2"] = filter_v2_leg(mf, files, a)
                except Exception as e:
                    extra["filter_v2"] = {"error": str(e)[:200]}
            elif "fv2_error" in files:
                extra["filter_v2"] = {"error": files["fv2_error"]}
            if "g20" in files:
                try:
                    extra["group_a"] = group_a(files)
                except Exception as e:
                    extra["group_a"] = {"error": str(e)[:200]}
    if world > 1:
        # The file path over all the node's devices in ONE process (mf_filter_fastq_files(..., n_devices = N): batches of whole
        # pairs dealt to one worker per device, no collective) next to the resident rate above -- it is the host that feeds it,
        # and the north star's ">= 6x at 8 GPUs" is a different question for each of the two.  Rank 0 runs it, the others wait.
        if rank == 0 and "small" in files and n_dev >= world:
            try:
                t = files["small"]
                best = 1e9
                for _ in range(3):
                    c0 = time.perf_counter()
                    mf.filter_fastq_files(ks, t + "_1.fq", t + "_2.fq", t + "_o1.fq", t + "_o2.fq", 1, 0, world)
                    best = min(best, time.perf_counter() - c0)
                extra["file_level_all_devices"] = {"n_devices": world, "pairs": a.e2e_pairs, "reads_per_s": 2 * a.e2e_pairs / best,
                                                   "note": "host-feed bound (parse, pack, PCIe, write): plain PE files, one process, one worker pair per device"}
            except Exception as e:
                extra["file_level_all_devices"] = {"error": str(e)[:200]}
        elif rank == 0:
            extra["file_level_all_devices"] = {"skipped": "fewer visible devices than ranks" if n_dev < world else "no input files"}
        rdv.barrier(timeout=1800.0)
        # every rank's own .gz shard, file to file, through the device ingest path on its own GPU (one process per GPU: the form in which
        # .gz input scales over a node -- `--devices N` in one process deals one stream's slabs to N devices instead)
        mine_rate, path = 0.0, -1
        if rank_gz:
            try:
                reads.close()
                best = 1e9
                for _ in range(3):
                    c0 = time.perf_counter()
                    mf.filter_fastq_files(ks, rank_gz, None, rank_gz + ".out", None, devices=[dev])
                    best = min(best, time.perf_counter() - c0)
                mine_rate = a.rank_file_reads / best
                path = mf.last_ingest_stats()["path"]
            except Exception:
                mine_rate = 0.0
        rates = rdv.gather(mine_rate, timeout=1800.0)
        paths = rdv.gather(float(path), timeout=1800.0)
        if rank == 0:
            extra["per_gpu_file_reads_per_s"] = rates
            extra["file_level_one_process_per_gpu"] = {"reads_per_rank": a.rank_file_reads, "aggregate_reads_per_s": sum(rates), "device_ingest_on_every_rank": all(p == 1.0 for p in paths),
                                                       "note": "each rank: its own gzip -1 shard -> mf_filter_fastq_files_on(devices=[its GPU]); small shards, a call's fixed latencies dominate; MF_DEVPOOL_GB=4 per rank"}
    if rank_tmp is not None:
        rank_tmp.cleanup()
    if tmpdir is not None:
        tmpdir.cleanup()

    traffic, traffic_src = committed_traffic()
    if not default_set or a.k != K:
        traffic, traffic_src = None, None              # the committed profile was taken on the default workload
    if live[0] is not None:
        traffic, traffic_src = live
    elif live[1] and traffic_src:
        traffic_src += f" (live collection unavailable: {live[1]})"
    if rank == 0:
        hbm = {"achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS}
        roofline = {"bound": "hbm", **hbm, "traffic": traffic, "traffic_source": traffic_src,
                    "kernel": "screen_kernel", "algorithmic_bytes_per_launch": int(alg_bytes),
                    "avg_kernel_ms": sp.ms_screen, "kernel_launches_averaged": a.steps,
                    # `frac` is the kernel's launch duration inside the pipelined loop (it shares the device with its neighbours there):
                    # beside it the same kernel with the device to itself, and the whole pass (all kernels, alg. bytes / time per step)
                    "kernel_alone_frac": alone_frac, "whole_pass_frac": whole_pass_frac}
        if a.k < 28:
            # the stride-8 screen (k < 28) tests twice the samples and is bound by vector-instruction issue, not by HBM: price it
            # against the issue rate (SQ_INSTS_VALU wave-instructions per launch, one per 4 cycles per wave stream, 1024 SIMDs)
            roofline["hbm"] = hbm
            roofline["bound"] = "valu"
            if valu[0] is not None and screen_s > 0:
                ginst = valu[0]["SQ_INSTS_VALU"] / screen_s / 1e9
                roofline.update({"achieved": ginst, "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s", "frac": ginst / VALU_PEAK_GINST,
                                 "valu_wave_instructions_per_launch": valu[0]["SQ_INSTS_VALU"],
                                 "valu_source": "live: rocprofv3 --pmc SQ_INSTS_VALU pass of this script on this box; peak = 1024 SIMDs x 2.4 GHz / 4 cycles"})
            else:
                roofline.update({"achieved": None, "peak": VALU_PEAK_GINST, "unit": "G wave-instructions/s", "frac": None,
                                 "valu_source": f"no live SQ_INSTS_VALU reading ({valu[1] or live[1] or 'not collected'}); see profiles/ for the committed one"})
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
            "roofline": roofline,
            "cpu_baseline": cpu,
            "extra": extra,
        }
        print(json.dumps(out))
    if rdv is not None:
        rdv.close()
    # a leg that raised is in `extra` as {"error": ...} (the line is still printed: the other legs' figures stand) -- and the run fails
    def find_errors(d, path):
        found = []
        if isinstance(d, dict):
            if "error" in d:
                found.append(path)
            for k2, v2 in d.items():
                found += find_errors(v2, path + "." + str(k2) if path else str(k2))
        return found
    bad = find_errors(extra, "extra") if rank == 0 else []
    if bad:
        sys.exit("bench.py: legs that raised: " + ", ".join(bad))


if __name__ == "__main__":
    main()
