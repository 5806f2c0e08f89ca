"""Rows A1-A6: the drop-in `fastfilter` CLI against golden vectors captured from the reference's
prebuilt ELF (tests/golden/fastfilter_golden.json) and against the Python restatement
(oracle/fastfilter_ref.py) on randomized inputs.  Bit-exact: stdout, exit code, output payload."""
import gzip
import hashlib
import json
import os
import random
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "fastfilter_golden.json")))


@pytest.fixture(scope="module")
def cli(built_lib):
    p = os.path.join(os.path.dirname(HERE), "mitoflex_amd", "assemble", "fastfilter")
    assert os.path.exists(p)
    return p


def _run(cli, tmp, case_input, enc, in_name, out_name, argv):
    inp, outp = os.path.join(tmp, in_name), os.path.join(tmp, out_name)
    for p in (inp, outp):
        if os.path.exists(p):
            os.remove(p)
    if case_input is not None:
        raw = case_input if isinstance(case_input, bytes) else case_input.encode(enc)
        if in_name.endswith(".gz"):
            with gzip.open(inp, "wb") as f:
                f.write(raw)
        else:
            open(inp, "wb").write(raw)
    args = [a.replace("{in}", inp).replace("{out}", outp) for a in argv]
    p = subprocess.run([cli] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    ob = None
    if os.path.exists(outp):
        ob = open(outp, "rb").read()
        if out_name.endswith(".gz"):
            ob = gzip.decompress(ob) if ob else b""
    return p.returncode, p.stdout, ob


@pytest.mark.parametrize("case", GOLD["cases"], ids=[c["name"] for c in GOLD["cases"]])
def test_golden(cli, tmp_path, case):
    rc, so, ob = _run(cli, str(tmp_path), case["input"], case["input_encoding"], case["in_name"], case["out_name"], case["argv"])
    assert rc == case["rc"]
    assert so == case["stdout"].encode("latin-1")
    exp = None if case["output"] is None else case["output"].encode("latin-1")
    assert ob == exp


def test_oracle_matches_goldens(tmp_path):
    """The Python restatement itself is pinned by the same vectors."""
    from oracle import fastfilter_ref as ref
    for case in GOLD["cases"]:
        if case["argv"][0] in ("-V", "--help"):
            continue
        raw = None if case["input"] is None else case["input"].encode(case["input_encoding"])
        rc, so, ob = ref.run([a.replace("{in}", "IN").replace("{out}", "OUT") for a in case["argv"]], lambda p: raw)
        assert rc == case["rc"], case["name"]
        assert so == case["stdout"].encode("latin-1"), case["name"]
        exp = None if case["output"] is None else case["output"].encode("latin-1")
        assert ob == exp, case["name"]


def _bulk(n, seed):
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(HERE, "golden", "make_fastfilter_golden.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    return mk.bulk_fasta(n, seed)


@pytest.mark.parametrize("case", [b for b in GOLD["bulk"] if b["n_records"] <= 20000 or os.environ.get("MF_SLOW_TESTS")],
                         ids=lambda b: b["name"])
def test_bulk_golden(cli, tmp_path, case):
    text = _bulk(case["n_records"], case["seed"])
    assert hashlib.md5(text.encode()).hexdigest() == case["input_md5"]
    rc, so, ob = _run(cli, str(tmp_path), text, "latin-1", "bulk.fa", "bulk.out.fa", case["argv"])
    assert rc == case["rc"] and so.decode() == case["stdout"]
    assert hashlib.md5(ob).hexdigest() == case["output_md5"] and ob.count(b"\n") == case["output_lines"]


def test_random_against_oracle(cli, tmp_path):
    from oracle import fastfilter_ref as ref
    rng = random.Random(5)
    depths = ["0.0000", "1.0000", "2.9999", "3.0000", "3.0001", "10.5000", "16777216.0000", "1e2", "NaN", "inf", "x"]
    for it in range(150):
        recs = []
        for i in range(rng.randint(0, 12)):
            style = rng.random()
            L = rng.choice([0, 1, 2, 5, 6, 7, 20, 50])
            seq = "".join(rng.choices("ACGT", k=L))
            if style < 0.75:
                h = f">k{i} flag={rng.randint(0, 2)} multi={rng.choice(depths[:7])} len={L}"
            elif style < 0.85:
                h = f">k{i} multi={rng.choice(depths)}"
            elif style < 0.9:
                h = f"k{i} flag=1 multi=5.0 len={L}"
            else:
                h = f">k{i}\tflag=1  multi={rng.choice(depths)}   len={L}"
            eol = "\r\n" if rng.random() < 0.1 else "\n"
            recs.append(h + eol + seq + eol)
        text = "".join(recs)
        if rng.random() < 0.2:
            text = text.rstrip("\n")
        if rng.random() < 0.1:
            text += ">odd line"
        l = f"{rng.choice([0, 1, 5, 6])},{rng.choice([5, 6, 7, 50, 20000])}"
        mode = ["-d", str(rng.choice([0, 1, 3, 10, 11]))] if rng.random() < 0.6 else ["-m", str(rng.choice([0, 1, 3, 100]))]
        gz_in, gz_out = rng.random() < 0.15, rng.random() < 0.15
        in_name, out_name = "r.fa" + (".gz" if gz_in else ""), "o.fa" + (".gz" if gz_out else "")
        argv = ["-i", "{in}", "-o", "{out}", "-l", l] + mode
        rc, so, ob = _run(cli, str(tmp_path), text, "latin-1", in_name, out_name, argv)
        erc, eso, eob = ref.run(["-i", "I", "-o", "O", "-l", l] + mode, lambda p: text.encode("latin-1"))
        assert (rc, so) == (erc, eso), (it, argv, text)
        assert ob == eob, (it, argv, text)
