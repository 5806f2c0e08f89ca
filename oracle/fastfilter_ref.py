"""TEST INFRASTRUCTURE ONLY -- Python restatement of the reference contig filter
(`assemble/fastfilter`, Rust: assemble/fastfilter_src/src/main.rs:9-134, helper.rs:12-41).

Pinned against golden vectors captured from the reference's prebuilt ELF
(tests/golden/fastfilter_golden.json, made by tests/golden/make_fastfilter_golden.py);
used to check the product CLI (mitoflex_amd/assemble/fastfilter) on randomized
inputs beyond those vectors.  The product never imports this.

run(argv, read_file, ...) -> (rc, stdout_bytes, output_bytes | None)
  rc 0 ok, 101 = Rust panic, 1 = clap usage error.
"""
from __future__ import annotations

import math
import re
import struct
from typing import Callable, List, Optional, Tuple


class Panic(Exception):
    pass


class ClapError(Exception):
    pass


_USIZE_MAX = (1 << 64) - 1


def parse_usize(s: str) -> int:
    # str::parse::<usize>  (main.rs:60)
    if not re.fullmatch(r"\+?[0-9]+", s):
        raise Panic("ParseIntError")
    v = int(s)
    if v > _USIZE_MAX:
        raise Panic("ParseIntError overflow")
    return v


def parse_i32(s: str) -> int:
    # str::parse::<i32>  (main.rs:75)
    if not re.fullmatch(r"[+-]?[0-9]+", s):
        raise Panic("ParseIntError")
    v = int(s)
    if not -(1 << 31) <= v < (1 << 31):
        raise Panic("ParseIntError overflow")
    return v


def parse_f32(s: str) -> float:
    # str::parse::<f32> as the shipped binary does it (goldens X14-X17): optional sign, then
    # exactly "inf" / "NaN", or a decimal with optional exponent; result rounded to binary32
    m = re.fullmatch(r"([+-]?)(inf|NaN|(?:[0-9]+\.?[0-9]*|\.[0-9]+)(?:[eE][+-]?[0-9]+)?)", s)
    if not m:
        raise Panic("ParseFloatError")
    body = m.group(2)
    if body == "inf":
        v = math.inf
    elif body == "NaN":
        v = math.nan
    else:
        try:
            v = struct.unpack("f", struct.pack("f", float(body)))[0]
        except OverflowError:
            v = math.inf
    return -v if m.group(1) == "-" else v


def f32(x: float) -> float:
    return struct.unpack("f", struct.pack("f", float(x)))[0]


def header_depth(title: str) -> float:
    # title.split_whitespace()[2].split('=')[1].parse::<f32>().unwrap()   (main.rs:86-91)
    toks = title.split()           # str.split() uses the same Unicode White_Space set for these inputs
    if len(toks) < 3:
        raise Panic("index out of bounds")
    parts = toks[2].split("=")
    if len(parts) < 2:
        raise Panic("index out of bounds")
    return parse_f32(parts[1])


def lines_of(data: bytes) -> List[bytes]:
    # BufRead::lines(): split on '\n', strip one trailing '\r'; a trailing empty piece is not a line
    if not data:
        return []
    parts = data.split(b"\n")
    if parts[-1] == b"":
        parts.pop()
    return [p[:-1] if p.endswith(b"\r") else p for p in parts]


def _utf8(b: bytes) -> str:
    try:
        return b.decode("utf-8")
    except UnicodeDecodeError:
        raise Panic("stream did not contain valid UTF-8")


def parse_args(argv: List[str]):
    opts = {}
    i = 0
    while i < len(argv):
        a = argv[i]
        if a in ("-h", "--help"):
            return {"help": True}
        if a in ("-V", "--version"):
            return {"version": True}
        if len(a) < 2 or a[0] != "-" or a[1] == "-" or a[1] not in "ldiom":
            raise ClapError(f"unexpected argument {a}")
        o = a[1]
        if len(a) > 2:
            val = a[2:]
            if val.startswith("="):
                val = val[1:]
        else:
            if i + 1 >= len(argv):
                raise ClapError(f"-{o} requires a value")
            i += 1
            val = argv[i]
            if len(val) > 1 and val.startswith("-"):
                raise ClapError(f"unexpected argument {val}")
        if o in opts:
            raise ClapError(f"-{o} provided more than once")
        opts[o] = val
        i += 1
    if "d" in opts and "m" in opts:
        raise ClapError("-d cannot be used with -m")
    for req in "ilo":
        if req not in opts:
            raise ClapError(f"missing -{req}")
    return opts


def run(argv: List[str], read_file: Callable[[str], Optional[bytes]]) -> Tuple[int, bytes, Optional[bytes]]:
    """read_file(path) -> decompressed bytes or None when it cannot be opened.
    Returns (rc, stdout, output payload or None if the output file was never created)."""
    stdout = b""
    out: Optional[List[bytes]] = None
    try:
        try:
            o = parse_args(argv)
        except ClapError:
            return 1, b"", None
        if o.get("help") or o.get("version"):
            return 0, b"", None            # text checked through the goldens only
        lengths = [parse_usize(p) for p in o["l"].split(",") if p != ""]     # clap drops empty pieces (golden X37)
        if len(lengths) != 2:
            stdout += b"Input length string not valid, please input INT,INT.\n"
        if len(lengths) < 2:
            raise Panic("index out of bounds")
        mn, mx = lengths[0], lengths[1]
        data = read_file(o["i"])
        if data is None:
            raise Panic("Cannot open file")
        out = []
        lines = lines_of(data)
        count = 0
        if "m" not in o:
            if "d" not in o:
                raise Panic("unwrap on None")
            depth = parse_i32(o["d"])
            for i in range(0, len(lines) - 1, 2):
                title, seq = _utf8(lines[i]), _utf8(lines[i + 1])
                if not title.startswith(">"):
                    continue
                if depth != 0:
                    if f32(depth) > header_depth(title):
                        continue
                length = (len(lines[i + 1]) - 1) & _USIZE_MAX        # seq.len() - 1, wrapping in release
                if length < mn or length > mx:
                    continue
                out += [lines[i], lines[i + 1]]
                count += 1
        else:
            max_count = parse_usize(o["m"])
            strs = [_utf8(l) for l in lines]
            seqs = [(lines[i], lines[i + 1]) for i in range(0, len(lines) - 1, 2) if mn <= len(lines[i + 1]) <= mx]
            if len(seqs) >= 2:                      # sort_by_cached_key evaluates every key (and may panic)
                for t, _ in seqs:
                    header_depth(t.decode("utf-8"))
            for t, s in reversed(seqs):
                if count >= max_count:
                    break
                out += [t, s]
                count += 1
            del strs
        stdout += str(count).encode() + b"\n"
        return 0, stdout, b"".join(l + b"\n" for l in out)
    except Panic:
        return 101, stdout, (None if out is None else b"".join(l + b"\n" for l in out))
