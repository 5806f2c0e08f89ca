"""The C-ABI library loads without a GPU and exports every symbol include/mitofilter.h declares;
compute entry points fail loudly (no CPU fallback) when no device is visible."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exports_match_header(built_lib):
    from mitoflex_amd import mitofilter
    hdr = open(os.path.join(ROOT, "include", "mitofilter.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mf_[a-z0-9_]+)\s*\(", hdr))
    declared -= {n for n in declared if n.endswith("_t")}
    assert declared == set(mitofilter.EXPORTS)
    for name in declared:
        assert hasattr(built_lib, name), name
    assert built_lib.mf_abi_version() == 5


def test_no_cpu_fallback(built_lib):
    from mitoflex_amd import mitofilter as mf
    if mf.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(mf.MitoFilterError, match="no CPU fallback|no HIP device"):
        mf.KmerSet.from_text(">a\n" + "ACGT" * 20 + "\n", 31)
    with pytest.raises(mf.MitoFilterError, match="no CPU fallback|no HIP device"):
        mf.KmerSet.protein_from_text(">p\nMLSFIVGATMPYNWKEDHQRC\n", 7, 5)


def test_product_does_not_touch_oracle():
    """Nothing under mitoflex_amd/ (or include/) may import, link or call the oracle."""
    bad = []
    for base in ("mitoflex_amd", "include"):
        for d, _, files in os.walk(os.path.join(ROOT, base)):
            if os.sep + "build" in d:
                continue
            for f in files:
                if f.endswith((".py", ".cpp", ".h", ".hip", "Makefile")):
                    txt = open(os.path.join(d, f), errors="replace").read()
                    if re.search(r"oracle_lib|kmer_bait_oracle|libmf_oracle|from oracle|import oracle|mfo_", txt):
                        bad.append(os.path.join(d, f))
    assert not bad, bad


def test_cli_bait_mode_fails_loudly_without_gpu(built_lib, tmp_path):
    import subprocess
    from mitoflex_amd import mitofilter as mf
    if mf.device_count() > 0:
        pytest.skip("a GPU is visible")
    cli = os.path.join(ROOT, "mitoflex_amd", "assemble", "fastfilter")
    (tmp_path / "b.fa").write_text(">b\n" + "ACGT" * 30 + "\n")
    (tmp_path / "r.fq").write_text("@r\nACGT\n+\nIIII\n")
    p = subprocess.run([cli, "bait", "--bait", str(tmp_path / "b.fa"), "--fq1", str(tmp_path / "r.fq"),
                        "--out1", str(tmp_path / "o.fq")], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode != 0 and p.stdout == b""
