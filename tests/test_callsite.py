"""Rows A7, A8, A10: the host-side mirror of the reference call site.

Golden data (tests/golden/callsite_golden.json) was produced by importing the reference's own
`utility/helper.py` and `assemble/assemble_wrapper.py` in the build container
(tests/golden/make_callsite_golden.py): the command strings and the exact shell-command sequence
`MEGAHIT.filter()` issues.  The mirror must reproduce them character for character."""
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "callsite_golden.json")))


@pytest.mark.parametrize("case", GOLD["concat"], ids=lambda c: c["command"][:40])
def test_concat_command(case):
    from mitoflex_amd.utility.helper import concat_command
    assert concat_command(*case["args"], **case["kwargs"]) == case["command"]


def test_direct_call_contract():
    from mitoflex_amd.utility import helper
    assert helper.direct_call("printf '42\\n'") == "42\n"
    assert int(helper.shell_call("echo", 7)) == 7
    with pytest.raises(RuntimeError) as e:
        helper.direct_call("exit 3")
    assert str(e.value) == "Error when running command 'exit 3'. Exiting."


@pytest.mark.parametrize("sc", GOLD["filter"], ids=lambda s: s["name"])
def test_filter_command_sequence(sc, tmp_path, monkeypatch):
    from mitoflex_amd.assemble import assemble_wrapper as w
    from mitoflex_amd.utility import helper
    for suffix in sc["present"]:
        (tmp_path / f"k{sc['kwargs']['kmer']}{suffix}").write_text(">x\nACGT\n")
    for k, v in sc["conf"].items():
        monkeypatch.setattr(w.a_conf, k, v)
    calls, replies = [], list(sc["canned"])
    bin_dir = os.path.dirname(os.path.abspath(w.__file__))

    def fake(command):
        calls.append(command.replace(str(tmp_path), "{dir}").replace(bin_dir, "{bin}"))
        return replies.pop(0) if "/fastfilter " in command else ""
    monkeypatch.setattr(helper, "direct_call", fake)
    m = w.MEGAHIT(contig_dir=str(tmp_path))
    assert list(m.filter(**sc["kwargs"])) == sc["result"]
    assert calls == sc["commands"]


def test_filter_end_to_end_with_built_cli(built_lib, tmp_path):
    """The real thing: MEGAHIT.filter() -> shell_call -> built `fastfilter` -> int(stdout) -> mv."""
    from mitoflex_amd.assemble import assemble_wrapper as w
    gold = json.load(open(os.path.join(HERE, "golden", "fastfilter_golden.json")))
    g1 = next(c for c in gold["cases"] if c["name"] == "G1_depth_len")
    src = tmp_path / "k31.contigs.fa"
    src.write_text(g1["input"])
    m = w.MEGAHIT(contig_dir=str(tmp_path))
    assert os.path.exists(m.FAST_FILTER)
    assert m.filter(31, min_depth=3, min_length=5, max_length=15, deny_number=0) == (2, 0, 0)
    assert src.read_text() == g1["output"]                       # rewritten in place (mv)
    assert not (tmp_path / "k31.filtered.contigs.fa").exists()
    # deny_number fallback: depth filter keeps nothing -> "-m N" keeps the last N in reverse order
    src.write_text(g1["input"])
    assert m.filter(31, min_depth=1000, min_length=0, max_length=100, deny_number=2)[0] == 2
    g3 = next(c for c in gold["cases"] if c["name"] == "G3_m2")
    assert src.read_text() == g3["output"]


def test_build_lib_writes_reference_layout(tmp_path, monkeypatch):
    from mitoflex_amd.assemble import assemble_wrapper as w
    from mitoflex_amd.utility import helper
    calls = []
    monkeypatch.setattr(helper, "direct_call", lambda c: calls.append(c) or "")
    m = w.MEGAHIT(fq1="/d/a_1.fq", fq2="/d/a_2.fq", temp_dir=str(tmp_path), read_lib=str(tmp_path / "reads.lib"))
    m.build_lib()
    assert (tmp_path / "reads.lib").read_text() == "/d/a_1.fq,/d/a_2.fq\npe /d/a_1.fq /d/a_2.fq\n"
    assert calls == [f"megahit_core buildlib {tmp_path}/reads.lib {tmp_path}/reads.lib"]
    m = w.MEGAHIT(fq1="/d/s.fq", fq2=None, temp_dir=str(tmp_path), read_lib=str(tmp_path / "se.lib"))
    m.build_lib()
    assert (tmp_path / "se.lib").read_text() == "/d/s.fq\nse /d/s.fq\n"
