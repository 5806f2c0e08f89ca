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


FAKE_CORE = """#!/bin/bash
# stands in for `megahit_core buildlib LIB LIB`: reads the mates named by the lib's second line
lib=$2; line=$(sed -n 2p "$lib"); set -- $line
if [ "$1" = pe ]; then cat "$2" > "$lib.m1" & cat "$3" > "$lib.m2"; wait; else cat "$2" > "$lib.m1"; fi
"""


def test_build_lib_fifo_handoff_plumbing(tmp_path, monkeypatch):
    """bait_fifo: survivors reach `buildlib` through named pipes (ref :176-184 does the same for gunzipped
    reads).  The filter is a stub here (no GPU on this box); tests/test_gpu_parity.py runs the real one."""
    from mitoflex_amd.assemble import assemble_wrapper as w
    stub = tmp_path / "fastfilter"
    stub.write_text("#!/bin/bash\n# bait --bait B --kmer K --threshold T --fq1 A --fq2 B --out1 X --out2 Y ...\n"
                    "while [ $# -gt 0 ]; do case $1 in --fq1) a=$2;; --fq2) b=$2;; --out1) x=$2;; --out2) y=$2;; esac; shift; done\n"
                    "cat $a > $x & [ -n \"$b\" ] && cat $b > $y; wait; echo 7\n")
    stub.chmod(0o755)
    core = tmp_path / "megahit_core"
    core.write_text(FAKE_CORE)
    core.chmod(0o755)
    (tmp_path / "a_1.fq").write_text("@r\nACGT\n+\nIIII\n")
    (tmp_path / "a_2.fq").write_text("@r\nTTTT\n+\nIIII\n")
    monkeypatch.setattr(w.MEGAHIT, "FAST_FILTER", property(lambda self: str(stub)))
    monkeypatch.setattr(w.MEGAHIT, "MEGAHIT_CORE", str(core))
    monkeypatch.setattr(w.a_conf, "bait_fasta", str(tmp_path / "bait.fa"))
    monkeypatch.setattr(w.a_conf, "bait_fifo", True)
    t = tmp_path / "t"; t.mkdir()
    m = w.MEGAHIT(fq1=str(tmp_path / "a_1.fq"), fq2=str(tmp_path / "a_2.fq"), temp_dir=str(t), read_lib=str(t / "reads.lib"))
    m.build_lib()
    assert (t / "reads.lib").read_text() == f"{tmp_path}/a_1.fq,{tmp_path}/a_2.fq\npe {t}/pipe.bait1 {t}/pipe.bait2\n"
    assert (t / "reads.lib.m1").read_text() == "@r\nACGT\n+\nIIII\n" and (t / "reads.lib.m2").read_text() == "@r\nTTTT\n+\nIIII\n"
    assert m.bait_kept == 7
    t2 = tmp_path / "t2"; t2.mkdir()
    m = w.MEGAHIT(fq1=str(tmp_path / "a_1.fq"), fq2=None, temp_dir=str(t2), read_lib=str(t2 / "reads.lib"))
    m.build_lib()
    assert (t2 / "reads.lib").read_text() == f"{tmp_path}/a_1.fq\nse {t2}/pipe.bait1\n"
    assert (t2 / "reads.lib.m1").read_text() == "@r\nACGT\n+\nIIII\n"


def test_build_lib_fifo_cleans_up_when_buildlib_fails(tmp_path, monkeypatch):
    """`megahit_core buildlib` dying must not leave the bait filter blocked on its pipes, nor the pipes in temp_dir: the
    same temp_dir is usable for a retry."""
    import pytest
    from mitoflex_amd.assemble import assemble_wrapper as w
    stub = tmp_path / "fastfilter"
    stub.write_text("#!/bin/bash\nwhile [ $# -gt 0 ]; do case $1 in --fq1) a=$2;; --out1) x=$2;; esac; shift; done\n"
                    "echo $$ > $(dirname $x)/filter.pid\ncat $a > $x; echo 1\n")     # blocks opening the pipe: nobody reads it
    stub.chmod(0o755)
    bad_core = tmp_path / "megahit_core"
    bad_core.write_text("#!/bin/bash\nfor i in $(seq 200); do [ -s %s/t/filter.pid ] && break; sleep 0.05; done\nsleep 0.2\nexit 3\n" % tmp_path)       # (dies once the filter is up and blocked on its pipe)
    bad_core.chmod(0o755)
    (tmp_path / "a_1.fq").write_text("@r\nACGT\n+\nIIII\n")
    monkeypatch.setattr(w.MEGAHIT, "FAST_FILTER", property(lambda self: str(stub)))
    monkeypatch.setattr(w.MEGAHIT, "MEGAHIT_CORE", str(bad_core))
    monkeypatch.setattr(w.a_conf, "bait_fasta", str(tmp_path / "bait.fa"))
    monkeypatch.setattr(w.a_conf, "bait_fifo", True)
    t = tmp_path / "t"; t.mkdir()
    m = w.MEGAHIT(fq1=str(tmp_path / "a_1.fq"), fq2=None, temp_dir=str(t), read_lib=str(t / "reads.lib"))
    with pytest.raises(RuntimeError):
        m.build_lib()
    assert not (t / "pipe.bait1").exists()
    pid = int((t / "filter.pid").read_text())
    import time
    for _ in range(100):                                # the filter's process group is gone
        try:
            os.kill(pid, 0)
            time.sleep(0.02)
        except ProcessLookupError:
            break
    else:
        raise AssertionError("bait filter still running")
    good_core = tmp_path / "megahit_core"
    good_core.write_text(FAKE_CORE)
    m = w.MEGAHIT(fq1=str(tmp_path / "a_1.fq"), fq2=None, temp_dir=str(t), read_lib=str(t / "reads.lib"))
    m.build_lib()                                       # the retry in the same temp_dir works
    assert (t / "reads.lib.m1").read_text() == "@r\nACGT\n+\nIIII\n" and not (t / "pipe.bait1").exists()
