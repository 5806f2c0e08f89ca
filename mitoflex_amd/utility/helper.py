"""Command-line glue with the behaviour of the reference's `utility/helper.py:35-86`
(`shell_call`, `concat_command`, `direct_call`), so a MitoFlex stage can call this
package's tools exactly the way it calls its own.  Written from the behaviour, pinned by
tests/golden/callsite_golden.json (command strings produced by the reference itself).

Rules reproduced (reference lines in brackets):
  * positional arguments come first, joined by single spaces                     [59]
  * keyword arguments whose value is None are dropped                            [52-53]
  * one leading underscore of a keyword is stripped (`_in` -> `in`)              [52]
  * `useconv` (default on) turns '_' into '-' in keywords; it is consumed        [55-57]
  * one-letter keywords get '-', longer ones '--'                                [64,67,73]
  * list value -> `--key a b c`; True -> bare switch; False -> nothing           [63-69]
  * `appending=[...]` is appended verbatim                                       [70-71]
  * the command runs through `sh -c`; stdout is returned decoded as UTF-8; a
    non-zero exit prints the error and raises RuntimeError with the reference's
    message                                                                      [78-86]
"""
from __future__ import annotations

import subprocess
from typing import Any, Dict


def _dash(key: str) -> str:
    return "-" if len(key) == 1 else "--"


def concat_command(*args: Any, **kwargs: Any) -> str:
    named: Dict[str, Any] = {}
    for key, value in kwargs.items():
        if value is None:
            continue
        named[key[1:] if key.startswith("_") else key] = value
    useconv = named.pop("useconv") if "useconv" in named else True
    if useconv:
        named = {str(key).replace("_", "-"): value for key, value in named.items()}

    pieces = [" ".join(str(a) for a in args)]
    for key, value in named.items():
        if key == "appending" and isinstance(value, list):
            pieces.append(" ".join(value))
        elif isinstance(value, list):
            pieces.append(f"{_dash(key)}{key} {' '.join(value)}")
        elif isinstance(value, bool):
            if value:
                pieces.append(f"{_dash(key)}{key}")
        else:
            pieces.append(f"{_dash(key)}{key} {value}")
    return " ".join(pieces)


def direct_call(command: str) -> str:
    try:
        return subprocess.check_output(command, shell=True).decode("utf-8")
    except subprocess.CalledProcessError as err:
        print(err)
        raise RuntimeError(f"Error when running command '{command}'. Exiting.")


def shell_call(*args: Any, **kwargs: Any) -> str:
    return direct_call(concat_command(*args, **kwargs))
