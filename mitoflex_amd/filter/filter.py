"""Host-side mirror of the reference's `filter/filter.py:38-90`: builds the `filter_v2` command lines
exactly as MitoFlex does and runs them through `shell_call`, but against this package's drop-in
binary (mitoflex_amd/filter/filter_v2: GPU counting, same output bytes).  Logging and the
size report of the reference are not part of the path and are left out."""
from __future__ import annotations

import os
from os import path

from mitoflex_amd.utility import helper

filter_dir = os.path.dirname(os.path.abspath(__file__))


def filter_se(fqiabs=None, fqoabs=None, Ns=10, quality=55, limit=0.2, start=None, end=None, trim=0, trunc=False):
    helper.shell_call(path.join(filter_dir, 'filter_v2'), cleanq1=f'"{fqoabs}"', fastq1=f'"{fqiabs}"',
                      n=Ns, q=quality, l=limit, s=start, e=end, t=trim, truncate_only=trunc)
    return fqoabs


def filter_pe(fq1=None, fq2=None, o1=None, o2=None, dedup=False, start=None, end=None, n=10, q=55, l=0.2, trim=0, trunc=False):
    helper.shell_call(path.join(filter_dir, 'filter_v2'),
                      _1=f'"{fq1}"', _2=f'"{fq2}"', _3=f'"{o1}"', _4=f'"{o2}"', d=dedup, s=start,
                      e=end, n=n, q=q, l=l, t=trim, truncate_only=trunc)
    return o1, o2
