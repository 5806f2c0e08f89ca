#!/usr/bin/env python3
"""Command strings the reference's filter/filter.py issues for filter_v2 (build container only):
imports /root/reference/filter/filter.py with shell_call replaced by a recorder."""
import json
import os
import sys
import tempfile

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [
    ("se", dict(Ns=10, quality=55, limit=0.2, start=None, end=None, trim=0, trunc=False)),
    ("se", dict(Ns=3, quality=60, limit=0.3, start=2, end=100, trim=5000000000, trunc=True)),
    ("pe", dict(dedup=False, start=None, end=None, n=10, q=55, l=0.2, trim=0, trunc=False)),
    ("pe", dict(dedup=True, start=5, end=140, n=2, q=50, l=0.1, trim=5000000000, trunc=False)),
]


def main():
    import filter.filter as ref
    from utility import logger
    logger.log = lambda *a, **k: None
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        a, b = os.path.join(tmp, "a.fq"), os.path.join(tmp, "b.fq")
        for p in (a, b, os.path.join(tmp, "o1.fq"), os.path.join(tmp, "o2.fq")):
            open(p, "w").write("@r\nA\n+\nI\n")
        for kind, kw in CASES:
            calls = []
            ref.shell_call = lambda *args, **kwargs: calls.append(ref_helper.concat_command(*args, **kwargs)) or ""
            if kind == "se":
                ref.filter_se(fqiabs=a, fqoabs=os.path.join(tmp, "o1.fq"), **kw)
            else:
                ref.filter_pe(fq1=a, fq2=b, o1=os.path.join(tmp, "o1.fq"), o2=os.path.join(tmp, "o2.fq"), **kw)
            out.append({"kind": kind, "kwargs": kw,
                        "command": calls[0].replace(tmp, "{dir}").replace(os.path.dirname(ref.__file__), "{bin}")})
            print(out[-1]["command"])
    json.dump(out, open(os.path.join(HERE, "filter_callsite_golden.json"), "w"), indent=1)


if __name__ == "__main__":
    from utility import helper as ref_helper
    main()
