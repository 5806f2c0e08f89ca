"""The file-level legs of bench.py run once a round, on the GPU box, after minutes of set-up: a slip in the code that puts their
dictionaries together would cost the round its figures.  Here they run on the CPU with a stand-in for the library (files are copied,
statistics made up): every key the legs read exists, every dictionary they build is JSON."""
import importlib.util
import json
import os
import shutil
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return b


class FakeLib:
    """what the legs call of mitoflex_amd.mitofilter"""
    def __init__(self):
        self.calls = []

    class KmerSet:
        @staticmethod
        def from_fasta(path, k):
            return ("ks", path, k)

    def filter_fastq_files(self, ks, f1, f2, o1, o2, *a, **kw):
        self.calls.append(("bait", f1, f2))
        open(o1, "wb").write(b"@r\nACGT\n+\nIIII\n")
        if o2:
            open(o2, "wb").write(b"@r\nACGT\n+\nIIII\n")
        return 1, 4

    def qualfilter_files(self, f1, f2, o1, o2, **kw):
        self.calls.append(("qual", f1, f2, os.environ.get("MF_QUAL_INGEST")))
        open(o1, "wb").write(b"@r\nACGT\n+\nIIII\n")
        open(o2, "wb").write(b"@r\nACGT\n+\nIIII\n")
        return 1, 1, False

    def last_ingest_stats(self):
        return {"path": 1, "n_devices": 1, "consumers": 3, "input_bytes": 10, "text_bytes": 40, "records": 4, "seconds": 0.1, "decode_busy_seconds": 0.05,
                "pool_bytes_peak": 1 << 30, "device_bytes_peak": 2 << 30, "chunks": 7, "chunks_linked": 7, "gaps": 0, "gap_bytes": 0}

    def h2d_bandwidth(self, *a):
        return 57.0


def touch(path, data=b"@r\nACGT\n+\nIIII\n"):
    open(path, "wb").write(data)
    return path


def test_file_level_legs_build_their_dictionaries(bench, tmp_path):
    t = str(tmp_path)
    for n in ("s_1.fq", "s_2.fq", "s_1.fq.gz", "f_1.fq", "f_1.fq.gz", "r_1.fq", "r_1.gzip-6.fq.gz", "r.bait.fa"):
        touch(os.path.join(t, n))
    files = {"small": os.path.join(t, "s"), "full": os.path.join(t, "f"), "full_prep_seconds": {"generate": 1.0, "compress": 2.0},
             "real": {"plain": os.path.join(t, "r_1.fq"), "bait": os.path.join(t, "r.bait.fa"), "gz": {"gzip-6": os.path.join(t, "r_1.gzip-6.fq.gz")}}}
    a = types.SimpleNamespace(e2e_pairs=1, e2e_full_reads=4, real_gz_reads=4, k=31)
    lib = FakeLib()
    out = bench.e2e_files(lib, ("ks",), files, a)
    json.dumps(out)
    c4 = out["configs4_se_gz"]
    assert c4["roofline"]["bound"] == "pcie_h2d" and c4["roofline"]["peak"] == 57.0 and c4["output_equals_host_pipeline_on_plain_text"] is True
    assert "bound" in c4["inflate_kernels"] and c4["ingest_path"] == "device"
    assert out["real_compressors"]["files"]["gzip-6"]["output_equals_host_pipeline_on_plain_text"] is True
    assert os.environ.get("MF_INGEST") is None          # (the legs put the environment back)


def test_filter_v2_leg_builds_its_dictionary(bench, tmp_path):
    q = os.path.join(str(tmp_path), "q")
    for n in ("_1.fq", "_2.fq", "_1.fq.gz", "_2.fq.gz", "device_c1.fq", "device_c2.fq", "host_c1.fq", "host_c2.fq"):
        touch(q + n)
    files = {"fv2": {"prefix": q, "cli_seconds": {"device": 0.5, "host": 0.6}}}
    a = types.SimpleNamespace(fv2_pairs=1)
    lib = FakeLib()
    out = bench.filter_v2_leg(lib, files, a)
    json.dumps(out)
    assert out["cli_outputs_equal"] and out["library_outputs_equal"]
    assert out["library_call"]["device"]["ingest_path"] == "device" and set(out["cli_process_start_to_exit"]) == {"device", "host"}
    assert [c[3] for c in lib.calls if c[0] == "qual"] == [None, None, None, "host"] and os.environ.get("MF_QUAL_INGEST") is None


class FakeResident:
    """stand-in for the resident-path API the axis legs of bench.py use: the 'GPU' bits come from the oracle itself, the timings are made up"""
    MODE_SCREENED, MODE_EXHAUSTIVE = 0, 1

    def __init__(self, ol):
        self.ol = ol
        outer = self

        class KS:
            def __init__(self, bait, k):
                from mitoflex_amd.utility.synth_bait import bait_records
                self.bait, self.k = bait, k
                size = sum(len(r) for r in bait_records(bait))
                mode = 0 if size < 20_000 else 3 if size < 60_000 else 1 if size < 105_000 else 2
                self.info = types.SimpleNamespace(front_mode=mode, front2_log2_blocks=0 if mode == 0 else 15, front3_log2_blocks=20 if size > 2_000_000 else 0,
                                                  n_keys=size, n_smers=2 * size)

            @classmethod
            def from_text(cls, bait, k, dev=0):
                return cls(bait, k)

            def close(self):
                pass

        class Reads:
            def __init__(self, words, off, npos):
                self.host_words, self.off, self.host_npos = words, off, npos
                self.info = types.SimpleNamespace(n_reads=len(off) - 1)

            @classmethod
            def from_packed(cls, words, off, npos, dev=0):
                return cls(words, off, npos)

            @classmethod
            def synth(cls, n, L, seed, bait_text, **kw):
                import numpy as np
                rng = np.random.default_rng(seed)
                nw = (n * L + 15) // 16
                return cls(rng.integers(0, 2**32, size=nw + 8, dtype=np.uint32), np.arange(n + 1, dtype=np.uint64) * L, np.zeros(0, np.uint64))

            def close(self):
                pass
        self.KmerSet, self.Reads = KS, Reads

    def _stats(self, reads):
        n = reads.info.n_reads
        return types.SimpleNamespace(n_reads=n, n_pass=3, n_candidates=n // 10, ms_total=0.25, ms_screen=0.24, ms_mark=0.0, ms_exact=0.05,
                                     algorithmic_bytes=int(reads.off[-1]) // 4 + n // 8)

    def filter_resident(self, ks, reads, thr, mode, steps):
        return self._stats(reads)

    def filter_reads(self, ks, reads, thr, mode, want_hits=False):
        import numpy as np
        n = reads.info.n_reads
        R = self.ol.OracleReads.from_arrays(reads.host_words, reads.off, reads.host_npos)
        bits, hits = self.ol.filter_reads(self.ol.OracleTable(ks.bait, ks.k), R, thr, threads=2)
        return bits, (hits if want_hits else None), self._stats(reads)

    def device_synchronize(self, dev):
        pass


def test_resident_axis_legs_build_their_dictionaries(bench):
    """extra.bait_sweep / threshold_sweep / ragged / realistic: every key the legs read exists, the window checks run against the real
    oracle, the dictionaries are JSON -- with a stand-in for the device"""
    import numpy as np
    from oracle import oracle_lib as ol
    ol.lib()
    from mitoflex_amd.utility.synth_bait import make_bait
    lib = FakeResident(ol)
    bait = make_bait()
    a = types.SimpleNamespace(reads=3200, k=31, steps=7)
    reads = lib.Reads.synth(a.reads, 150, 5, bait)
    ks = lib.KmerSet.from_text(bait, 31)
    alg = 3200 * 150 // 4 + 400
    out = bench.bait_sweep_leg(lib, reads, bait, a, 0, alg, (3200 * 150 + 15) // 16)
    json.dumps(out)
    assert set(out) == {"33000", "100000", "350000", "1000000", "8500000"}
    assert [out[k]["roofline"]["bound"] for k in ("33000", "100000", "350000", "8500000")] == ["hbm", "valu", "l2_gather", "l2_gather"]
    assert all(v["window_bits_match_oracle"] and v["window_reads_checked"] == 3200 for v in out.values())
    assert out["8500000"]["front3_MiB"] == 16.0 and out["350000"]["roofline"]["peak"] == bench.GATHER_PEAK_GLOOKUPS
    t = bench.threshold_leg(lib, ks, reads, bait, a, 0, alg)
    json.dumps(t)
    assert set(t) == {"1", "2", "7", "exhaustive"} and t["2"]["window_bits_match_oracle"] and t["exhaustive"]["roofline"]["bound"] == "valu"
    r = bench.ragged_leg(lib, ks, reads, bait, a, 0, 0.22)
    json.dumps(r)
    assert 3200 * 150 / 150 <= r["reads"] <= 3200 * 150 / 60 and r["window_bits_match_oracle"]
    z = bench.realistic_leg(lib, a, 0, 0.22)
    json.dumps(z)
    assert z["window_bits_match_oracle"] and z["roofline"]["bound"] == "hbm"


def test_a_leg_that_raised_is_found(bench):
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "find_errors(extra" in src and "sys.exit(\"bench.py: legs that raised" in src
