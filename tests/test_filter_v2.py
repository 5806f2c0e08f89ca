"""`filter_v2` (SURVEY.md 8f "next" #2): the reference's FASTQ quality filter.
CPU part: the Python restatement (oracle/filter_v2_ref.py) against golden vectors captured from the
reference's ELF, and the wrapper mirror's command strings against the reference's own.
GPU part (-m gpu): the drop-in CLI (GPU counting) against the same vectors, bulk md5s and the oracle
on randomized inputs -- bit-exact output bytes and exit codes."""
import gzip
import hashlib
import importlib.util
import json
import os
import random
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "filter_v2_golden.json")))
CALLS = json.load(open(os.path.join(HERE, "golden", "filter_callsite_golden.json")))


def _mk():
    spec = importlib.util.spec_from_file_location("mkf", os.path.join(HERE, "golden", "make_filter_v2_golden.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def _expect(case):
    e1 = None if case["out1"] is None else case["out1"].encode("latin-1")
    e2 = None if case["out2"] is None else case["out2"].encode("latin-1")
    if case["name"] == "pe_out2_stdout":
        e2 = case["stdout"].encode("latin-1")
    return e1, e2


@pytest.mark.parametrize("case", GOLD["cases"], ids=[c["name"] for c in GOLD["cases"]])
def test_oracle_matches_elf(case):
    from oracle import filter_v2_ref as ref
    t1 = None if case["in1"] is None else case["in1"].encode("latin-1")
    t2 = None if case["in2"] is None else case["in2"].encode("latin-1")
    rc, o1, o2 = ref.run(list(case["argv"]), lambda p: t1 if (p is None or "in1" in p) else t2)
    e1, e2 = _expect(case)
    assert rc == case["rc"]
    assert (o1 or b"") == (e1 or b"") and (o2 or b"") == (e2 or b"")


def test_siphash_known_value():
    from oracle.filter_v2_ref import siphash13
    assert siphash13(b"") == 15130871412783076140          # DefaultHasher::new().finish()


@pytest.mark.parametrize("c", CALLS, ids=lambda c: c["kind"])
def test_wrapper_command_strings(c, tmp_path, monkeypatch):
    from mitoflex_amd.filter import filter as w
    from mitoflex_amd.utility import helper
    calls = []
    d = str(tmp_path)
    for f in ("a.fq", "b.fq"):
        open(os.path.join(d, f), "w").write("@r\nACGT\n+\nIIII\n")

    def fake(cmd):
        calls.append(cmd)
        open(os.path.join(d, "o1.fq"), "w").close()          # (the call site sizes its first output afterwards)
        return ""
    monkeypatch.setattr(helper, "direct_call", fake)
    if c["kind"] == "se":
        w.filter_se(fqiabs=f"{d}/a.fq", fqoabs=f"{d}/o1.fq", **c["kwargs"])
    else:
        w.filter_pe(fq1=f"{d}/a.fq", fq2=f"{d}/b.fq", o1=f"{d}/o1.fq", o2=f"{d}/o2.fq", **c["kwargs"])
    assert calls[0].replace(d, "{dir}").replace(os.path.dirname(os.path.abspath(w.__file__)), "{bin}") == c["command"]


def test_wrapper_errors_are_the_call_sites(tmp_path, monkeypatch):
    """What the reference's call site does when things go wrong (filter/filter.py:39, 51, 62, 84): a missing input is the
    FileNotFoundError of its `path.getsize`, a failing command ends the process with its message."""
    from mitoflex_amd.filter import filter as w
    from mitoflex_amd.utility import helper
    d = str(tmp_path)
    with pytest.raises(FileNotFoundError):
        w.filter_se(fqiabs=f"{d}/nothing.fq", fqoabs=f"{d}/o1.fq")
    open(f"{d}/a.fq", "w").write("@r\nACGT\n+\nIIII\n")
    with pytest.raises(FileNotFoundError):
        w.filter_pe(fq1=f"{d}/a.fq", fq2=f"{d}/nothing.fq", o1=f"{d}/o1.fq", o2=f"{d}/o2.fq")

    def boom(cmd):
        raise RuntimeError("exit status 101")
    monkeypatch.setattr(helper, "direct_call", boom)
    with pytest.raises(SystemExit) as e:
        w.filter_se(fqiabs=f"{d}/a.fq", fqoabs=f"{d}/o1.fq")
    assert str(e.value) == "Error occured when running filter!"


# ------------------------------------------------------------------------------------------- GPU
def _run_cli(cli, tmp, t1, t2, argv):
    paths = {k: os.path.join(tmp, v) for k, v in dict(in1="a_1.fq", in2="a_2.fq", out1="o_1.fq", out2="o_2.fq", in1gz="a_1.fq.gz",
                                                       in2gz="a_2.fq.gz", out1gz="o_1.fq.gz", out2gz="o_2.fq.gz").items()}
    for p in paths.values():
        if os.path.exists(p):
            os.remove(p)
    for key, t in (("in1", t1), ("in2", t2)):
        if t is not None:
            raw = t if isinstance(t, bytes) else t.encode("latin-1")
            open(paths[key], "wb").write(raw)
            with gzip.open(paths[key + "gz"], "wb") as f:
                f.write(raw)
    args = [a.format(**paths) for a in argv]
    use_stdin = "-1" not in argv and not any(a.startswith("--fastq1") for a in argv) and t1 is not None
    p = subprocess.run([cli] + args, input=((t1 if isinstance(t1, bytes) else t1.encode("latin-1")) if use_stdin else None),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    outs = []
    for k in ("out1", "out2"):
        ob = None
        for suffix in ("", "gz"):
            pth = paths[k + suffix]
            if os.path.exists(pth):
                ob = open(pth, "rb").read()
                if suffix == "gz":
                    ob = gzip.decompress(ob) if ob else b""
        outs.append(ob)
    return p.returncode, p.stdout, outs


@pytest.fixture(scope="module")
def cli(built_lib):
    from mitoflex_amd import mitofilter
    if mitofilter.device_count() < 1:
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    p = os.path.join(os.path.dirname(HERE), "mitoflex_amd", "filter", "filter_v2")
    assert os.path.exists(p)
    return p


@pytest.fixture(scope="module")
def cli_host_only(built_lib, tmp_path_factory):
    """The real drop-in CLI and the real host pipeline with tests/native/qualfilter_stub.cpp standing in for the
    GPU library (test infrastructure: obvious loops instead of the counting / hashing kernels) -- the host logic
    against the ELF's golden vectors without a GPU."""
    root = os.path.dirname(HERE)
    csrc = os.path.join(root, "mitoflex_amd", "csrc")
    so = str(tmp_path_factory.mktemp("stub") / "libqualfilter_stub.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", csrc, os.path.join(HERE, "native", "qualfilter_stub.cpp"),
                           *[os.path.join(csrc, f) for f in ("mf_pipeline.cpp", "mf_host.cpp", "mf_inflate.cpp", "mf_pinflate.cpp")],
                           "-lz", "-lpthread", "-o", so])
    p = os.path.join(root, "mitoflex_amd", "filter", "filter_v2")
    assert os.path.exists(p)
    return p, so


def _check_case(cli, tmp_path, case):
    rc, so, outs = _run_cli(cli, str(tmp_path), case["in1"], case["in2"], case["argv"])
    e1, e2 = _expect(case)
    assert rc == case["rc"]
    if case["name"] == "pe_out2_stdout":
        assert so == e2
    elif case["argv"][0] not in ("-h", "-V"):
        assert so == case["stdout"].encode("latin-1")
    assert outs[0] == e1
    if case["name"] != "pe_out2_stdout":
        assert outs[1] == (None if case["out2"] is None else case["out2"].encode("latin-1"))


@pytest.mark.parametrize("case", GOLD["cases"], ids=[c["name"] for c in GOLD["cases"]])
def test_cli_host_logic_matches_elf(cli_host_only, tmp_path, case, monkeypatch):
    monkeypatch.setenv("MITOFILTER_LIB", cli_host_only[1])
    _check_case(cli_host_only[0], tmp_path, case)


@pytest.mark.parametrize("case", GOLD["bulk"], ids=lambda b: b["name"])
def test_cli_host_logic_bulk_md5(cli_host_only, tmp_path, case, monkeypatch):
    monkeypatch.setenv("MITOFILTER_LIB", cli_host_only[1])
    _check_bulk(cli_host_only[0], tmp_path, case, monkeypatch)


def test_cli_host_logic_random_against_oracle(cli_host_only, tmp_path, monkeypatch):
    monkeypatch.setenv("MITOFILTER_LIB", cli_host_only[1])
    _check_random(cli_host_only[0], tmp_path, monkeypatch)


# The drop-in has two ways to the same bytes: regular files go to the GPU as they lie (inflate, line index, counting, hashing, the
# de-duplication set, decisions and the formatting of the kept records on the device: mf_devingest.cpp -- the default for .gz input,
# MF_QUAL_INGEST=device for plain files too), everything else (standard input, pipes, BGZF, .gz outputs) and MF_QUAL_INGEST=host take
# the host pipeline with GPU counting.  Both are held to the ELF's vectors.
# "device-seams": chunks of 4 KiB of compressed input, three to a slab, text pieces of a few hundred bytes to 20 kB, two text buffers
# a mate -- records, mates' pieces and decisions meet at every possible kind of border.
INGEST = {"device": {"MF_QUAL_INGEST": "device"}, "host": {"MF_QUAL_INGEST": "host"},
          "device-seams": {"MF_QUAL_INGEST": "device", "MF_GZDEV_CHUNK_BYTES": "4096", "MF_GZDEV_SLAB_CHUNKS": "3", "MF_GZDEV_TEXT_PIECE": "20000", "MF_INGEST_SLAB_BYTES": "333",
                           "MF_INGEST_TEXT_BUFS": "2", "MF_QUAL_OUT_CHUNK": "4096", "MF_INGEST_CONSUMERS": "3"}}


def _set_ingest(monkeypatch, ingest):
    monkeypatch.setenv("MF_PIPE_TIMING", "1")
    for k, v in INGEST[ingest].items():
        monkeypatch.setenv(k, v)


@pytest.mark.gpu
@pytest.mark.parametrize("ingest", ["device", "host", "device-seams"])
@pytest.mark.parametrize("case", GOLD["cases"], ids=[c["name"] for c in GOLD["cases"]])
def test_cli_matches_elf(cli, tmp_path, case, ingest, monkeypatch):
    _set_ingest(monkeypatch, ingest)
    _check_case(cli, tmp_path, case)


@pytest.mark.gpu
@pytest.mark.parametrize("ingest", ["device", "host", "device-seams"])
@pytest.mark.parametrize("case", GOLD["bulk"], ids=lambda b: b["name"])
def test_cli_bulk_md5(cli, tmp_path, case, ingest, monkeypatch):
    _set_ingest(monkeypatch, ingest)
    if ingest == "device-seams":
        monkeypatch.setenv("MF_INGEST_SLAB_BYTES", "7001")
    _check_bulk(cli, tmp_path, case, monkeypatch)
    if ingest == "device-seams":
        monkeypatch.setenv("MF_INGEST_SLAB_BYTES", "50001")
        _check_bulk(cli, tmp_path, case, monkeypatch, gz=True)


@pytest.mark.gpu
@pytest.mark.parametrize("ingest", ["device", "device-seams"])
def test_cli_bulk_second_output_to_stdout(cli, tmp_path, ingest, monkeypatch):
    """Without -4 the second mate's records go to standard output (helper.rs:46-49): a sink that takes its chunks in order -- from
    consumers that finish their pieces in any order, through a pool of two chunks."""
    _set_ingest(monkeypatch, ingest)
    monkeypatch.setenv("MF_QUAL_OUT_CHUNKS", "2")
    monkeypatch.setenv("MF_QUAL_OUT_CHUNK", "4096" if ingest == "device-seams" else "65536")
    if ingest == "device-seams":
        monkeypatch.setenv("MF_INGEST_SLAB_BYTES", "7001")
    case = [c for c in GOLD["bulk"] if c["name"] == "bulk_pe_dedup_q"][0]
    mk = _mk()
    s1, s2, q1, q2 = mk.rand_pair(case["n"], case["seed"], L=case["L"])
    t1, t2 = mk.fq(s1, q1, "a"), mk.fq(s2, q2, "b")
    argv = [a for a in case["argv"] if a not in ("-4", "{out2}")]
    rc, so, outs = _run_cli(cli, str(tmp_path), t1, t2, argv)
    assert rc == case["rc"]
    assert hashlib.md5(outs[0]).hexdigest() == case["out1_md5"]
    assert hashlib.md5(so).hexdigest() == case["out2_md5"]


@pytest.mark.gpu
def test_device_path_is_the_one_that_ran(cli, tmp_path):
    """A .gz pair through the CLI: the timing line of the device path's quality filter must appear (and not the host pipeline's)."""
    mk = _mk()
    s1, s2, q1, q2 = mk.rand_pair(2000, 5, L=100)
    for name, t in (("a_1.fq.gz", mk.fq(s1, q1, "a")), ("a_2.fq.gz", mk.fq(s2, q2, "b"))):
        with gzip.open(str(tmp_path / name), "wb") as f:
            f.write(t.encode("latin-1"))
    env = dict(os.environ, MF_PIPE_TIMING="1")
    env.pop("MF_QUAL_INGEST", None)
    p = subprocess.run([cli, "-1", str(tmp_path / "a_1.fq.gz"), "-2", str(tmp_path / "a_2.fq.gz"), "-3", str(tmp_path / "o_1.fq"), "-4", str(tmp_path / "o_2.fq"), "-d"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0, p.stderr
    assert b"[mf device ingest] quality filter: wall" in p.stderr and b"[mf qualfilter]" not in p.stderr, p.stderr
    from oracle import filter_v2_ref as ref
    t1, t2 = mk.fq(s1, q1, "a").encode("latin-1"), mk.fq(s2, q2, "b").encode("latin-1")
    erc, e1, e2 = ref.run(["-1", "a_1", "-2", "a_2", "-3", "o1", "-4", "o2", "-d"], lambda pth: t1 if "a_1" in pth else t2)
    assert erc == 0 and open(str(tmp_path / "o_1.fq"), "rb").read() == e1 and open(str(tmp_path / "o_2.fq"), "rb").read() == e2


def _check_bulk(cli, tmp_path, case, monkeypatch, gz=False):
    mk = _mk()
    monkeypatch.setenv("MF_BATCH_READS", "3001")           # several batches, dedup/trim state carried across
    monkeypatch.setenv("MF_PARSE_SEG", "40000")
    monkeypatch.setenv("MF_DEDUP_LOG2_SLOTS", "6")         # the device's dedup set starts with 64 slots: it is doubled and rehashed again and again
    s1, s2, q1, q2 = mk.rand_pair(case["n"], case["seed"], L=case["L"])
    t1, t2 = mk.fq(s1, q1, "a"), mk.fq(s2, q2, "b")
    argv = [a.replace("{in1}", "{in1gz}").replace("{in2}", "{in2gz}") for a in case["argv"]] if gz else case["argv"]
    rc, so, outs = _run_cli(cli, str(tmp_path), t1, t2 if "{in2}" in case["argv"] else None, argv)
    assert rc == case["rc"]
    assert hashlib.md5(outs[0]).hexdigest() == case["out1_md5"] and outs[0].count(b"\n") == case["out1_lines"]
    if case["out2_md5"]:
        assert hashlib.md5(outs[1]).hexdigest() == case["out2_md5"]


@pytest.mark.gpu
@pytest.mark.parametrize("ingest", ["device", "host", "device-seams"])
def test_cli_random_against_oracle(cli, tmp_path, ingest, monkeypatch):
    _set_ingest(monkeypatch, ingest)
    _check_random(cli, tmp_path, monkeypatch, seams=ingest == "device-seams")


def _check_random(cli, tmp_path, monkeypatch, seams=False):
    from oracle import filter_v2_ref as ref
    mk = _mk()
    rng = random.Random(11)
    for it in range(40):
        if seams:
            monkeypatch.setenv("MF_INGEST_SLAB_BYTES", str(rng.choice([97, 333, 5000, 1 << 20])))
            monkeypatch.setenv("MF_INGEST_CONSUMERS", str(rng.choice([1, 2, 5])))
        n = rng.randint(0, 300)
        s1, s2, q1, q2 = mk.rand_pair(n, 1000 + it, L=rng.choice([12, 40, 90]), lowq=rng.choice([0.02, 0.2]), dup=rng.choice([0, 0.4]))
        t1, t2 = mk.fq(s1, q1, "a", eol=rng.choice(["\n", "\r\n"])), mk.fq(s2, q2, "b")
        pe = rng.random() < 0.6
        argv = ["-1", "{in1}", "-3", "{out1}"] + (["-2", "{in2}", "-4", "{out2}"] if pe else [])
        if rng.random() < 0.5:
            a = rng.randint(0, 8); argv += ["-s", str(a), "-e", str(a + rng.randint(0, 60))]
        if rng.random() < 0.5:
            argv += ["-q", str(rng.choice([35, 42, 55, 70])), "-l", rng.choice(["0.1", "0.25", "0.5", "0.99"])]
        if rng.random() < 0.4:
            argv += ["-n", str(rng.choice([0, 1, 5]))]
        if rng.random() < 0.3:
            argv += ["-t", str(rng.randint(1, 4000))]
        if pe and rng.random() < 0.5:
            argv += ["-d"]
        if rng.random() < 0.15:
            argv += ["--truncate_only"]
        monkeypatch.setenv("MF_BATCH_READS", str(rng.choice([1, 7, 64, 2000000])))
        monkeypatch.setenv("MF_PARSE_SEG", str(rng.choice([64, 1000, 1 << 25])))
        monkeypatch.setenv("MF_DEDUP_LOG2_SLOTS", str(rng.choice([4, 24])))
        run_argv = [a.replace("{in1}", "{in1gz}").replace("{in2}", "{in2gz}") for a in argv] if seams and it % 2 else argv
        rc, so, outs = _run_cli(cli, str(tmp_path), t1, t2 if pe else None, run_argv)
        b1, b2 = t1.encode("latin-1"), t2.encode("latin-1")
        erc, e1, e2 = ref.run(list(argv), lambda p: b1 if "in1" in p else b2)
        assert rc == erc, (it, argv)
        assert (outs[0] or b"") == (e1 or b""), (it, argv)
        assert (outs[1] or b"") == (e2 or b""), (it, argv)


def test_trim_budget_exits_without_waiting_for_a_slow_producer(cli_host_only, tmp_path, monkeypatch):
    """The reference leaves its loop at the `break` of the -t budget (main.rs:254-259) and exits; the drop-in's reader
    thread must not sit in a read for the rest of a batch while a slow producer still holds stdin open."""
    import time
    monkeypatch.setenv("MITOFILTER_LIB", cli_host_only[1])
    out = str(tmp_path / "o.fq")
    recs = "".join("@r%d d\nACGTACGTAC\n+\nIIIIIIIIII\n" % i for i in range(20)).encode()
    p = subprocess.Popen([cli_host_only[0], "-3", out, "-t", "25"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    t0 = time.time()
    p.stdin.write(recs)
    p.stdin.flush()                                   # ... and the pipe stays open: the producer is "slow"
    try:
        rc = p.wait(timeout=20)
    finally:
        p.stdin.close()
        p.kill()
    assert rc == 0 and time.time() - t0 < 15
    assert open(out).read() == "".join("@r%d d\nACGTACGTAC\n+\nIIIIIIIIII\n" % i for i in range(2))
