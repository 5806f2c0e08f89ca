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
