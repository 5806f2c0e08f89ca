"""The call site of the quality filter (`filter/filter.py:38-90` of the reference): `filter_se` / `filter_pe` put the same
`filter_v2` command line together -- option for option, in the same order -- and hand it to `shell_call`, against this package's
drop-in binary (mitoflex_amd/filter/filter_v2: same output bytes).  What a caller can observe besides the files is kept: the
inputs are sized first (a missing one is the `FileNotFoundError` of `path.getsize`, filter.py:39, 62), a command that fails ends
the process with the reference's message (:51, 84), the first output is sized afterwards (:53, 86).  The log lines are not part
of the path and are left out."""
from __future__ import annotations

import os
import sys

from mitoflex_amd.utility import helper

filter_dir = os.path.dirname(os.path.abspath(__file__))
_FAILED = "Error occured when running filter!"


def _quoted(p):
    return '"%s"' % p


def _filter(inputs, first_output, options):
    """options: (name, value) in command-line order -- `concat_command` writes them as it finds them, `_1` as `-1`, one letter as a
    short option, a word as a long one, None and False not at all."""
    for f in inputs:
        os.path.getsize(f)
    try:
        helper.shell_call(os.path.join(filter_dir, "filter_v2"), **dict(options))
    except Exception:
        sys.exit(_FAILED)
    os.path.getsize(first_output)


def filter_se(fqiabs=None, fqoabs=None, Ns=10, quality=55, limit=0.2, start=None, end=None, trim=0, trunc=False):
    _filter([fqiabs], fqoabs,
            [("cleanq1", _quoted(fqoabs)), ("fastq1", _quoted(fqiabs)), ("n", Ns), ("q", quality), ("l", limit), ("s", start), ("e", end),
             ("t", trim), ("truncate_only", trunc)])
    return fqoabs


def filter_pe(fq1=None, fq2=None, o1=None, o2=None, dedup=False, start=None, end=None, n=10, q=55, l=0.2, trim=0, trunc=False):
    _filter([fq1, fq2], o1,
            [("_1", _quoted(fq1)), ("_2", _quoted(fq2)), ("_3", _quoted(o1)), ("_4", _quoted(o2)), ("d", dedup), ("s", start), ("e", end),
             ("n", n), ("q", q), ("l", l), ("t", trim), ("truncate_only", trunc)])
    return o1, o2
