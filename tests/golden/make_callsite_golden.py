#!/usr/bin/env python3
"""Capture the reference's own behaviour for the call-site rows (A7, A8, A10).

Imports /root/reference/utility/helper.py and /root/reference/assemble/assemble_wrapper.py IN
THE BUILD CONTAINER ONLY and records (a) the command strings `concat_command` produces for a
list of calls and (b) the exact sequence of shell commands `MEGAHIT.filter()` issues for a set
of scenarios (with `direct_call` replaced by a recorder that plays back canned outputs).  The
JSON it writes is data; nothing of the reference travels.
"""
import json
import os
import sys
import tempfile

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

CONCAT_CASES = [
    {"args": ["python", "fun.py"], "kwargs": {"foo": "bar"}},
    {"args": ["x"], "kwargs": {"i": "a.fa", "o": "b.fa", "l": "0,20000", "d": 10}},
    {"args": ["x"], "kwargs": {"i": "a.fa", "o": "b.fa", "l": "200,20000", "m": 0}},
    {"args": ["tool", "sub"], "kwargs": {"wow_fun": "method", "t": 8}},
    {"args": ["tool"], "kwargs": {"wow_fun": "method", "useconv": False}},
    {"args": ["tool"], "kwargs": {"_in": "x", "_l": 3, "flag": True, "noflag": False, "none": None}},
    {"args": ["tool"], "kwargs": {"many": ["a", "b", "c"], "k": ["1", "2"]}},
    {"args": ["tool"], "kwargs": {"appending": [">", "out.txt"], "q": 30}},
    {"args": ["tool", 1, 2.5], "kwargs": {"useconv": True, "a_b_c": 1}},
    {"args": [], "kwargs": {"x": 1}},
    {"args": ["tool"], "kwargs": {"appending": "notalist"}},
    {"args": ["fastfilter", "bait"], "kwargs": {"bait": "b.fa", "kmer": 31, "threshold": 1, "fq1": "r1.fq", "fq2": None,
                                                "out1": "o1.fq", "out2": None, "pair": "either", "devices": 1}},
]

FILTER_SCENARIOS = [
    # name, files present, conf overrides, filter kwargs, canned fastfilter outputs (in call order)
    ("all_three", [".contigs.fa", ".addi.fa", ".bubble_seq.fa"], {}, dict(kmer=31, min_depth=10, min_length=0, max_length=20000, deny_number=0), ["12\n", "3\n", "0\n"]),
    ("only_contigs", [".contigs.fa"], {}, dict(kmer=59, min_depth=20, min_length=200, max_length=20000, force_filter=True, deny_number=0), ["7\n"]),
    ("deny_fallback", [".contigs.fa", ".addi.fa"], {}, dict(kmer=31, min_depth=50, min_length=0, max_length=20000, deny_number=5), ["2\n", "5\n", "9\n"]),
    ("deny_zero_count", [".contigs.fa"], {}, dict(kmer=31, min_depth=50, min_length=0, max_length=20000, deny_number=0), ["0\n", "0\n"]),
    ("no_filter_conf", [".contigs.fa"], {"no_filter": True}, dict(kmer=31, min_depth=3), []),
    ("no_filter_forced", [".contigs.fa"], {"no_filter": True}, dict(kmer=141, min_depth=3, force_filter=True), ["1\n"]),
    ("defaults", [".contigs.fa", ".bubble_seq.fa"], {}, dict(kmer=99), ["4\n", "1\n"]),
    ("none_exist", [], {}, dict(kmer=31, min_depth=3), []),
]


def main():
    from utility import helper as ref_helper
    out = {"concat": [], "filter": []}
    for case in CONCAT_CASES:
        out["concat"].append({**case, "command": ref_helper.concat_command(*case["args"], **case["kwargs"])})

    import assemble.assemble_wrapper as ref_wrap
    for name, present, conf, kwargs, canned in FILTER_SCENARIOS:
        with tempfile.TemporaryDirectory() as tmp:
            for suffix in present:
                open(os.path.join(tmp, f"k{kwargs['kmer']}{suffix}"), "w").write(">x\nACGT\n")
            saved = {k: getattr(ref_wrap.a_conf, k) for k in conf}
            for k, v in conf.items():
                setattr(ref_wrap.a_conf, k, v)
            calls, replies = [], list(canned)

            def fake_direct_call(command):
                calls.append(command.replace(tmp, "{dir}").replace(os.path.dirname(ref_wrap.__file__), "{bin}"))
                if "/fastfilter " in command:
                    return replies.pop(0)
                return ""
            ref_helper.direct_call = fake_direct_call
            try:
                m = ref_wrap.MEGAHIT(contig_dir=tmp)
                kw = dict(kwargs)
                if "deny_number" not in kw:
                    pass
                result = m.filter(**kw)
            finally:
                for k, v in saved.items():
                    setattr(ref_wrap.a_conf, k, v)
            out["filter"].append({"name": name, "present": present, "conf": conf, "kwargs": kwargs, "canned": canned,
                                  "commands": calls, "result": list(result)})
            print(name, result, len(calls), "commands")
    with open(os.path.join(HERE, "callsite_golden.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
