#!/usr/bin/env python3
"""Capture golden vectors from the reference's prebuilt `assemble/fastfilter` ELF.

Runs ONLY in the build container (needs /root/reference); the GPU box and the
test-suite read the committed JSON, never the reference.  The Rust source
(assemble/fastfilter_src/src/main.rs) cannot be rebuilt here (no cargo/rustc),
so the shipped binary -- whose strings/symbols match that source, SURVEY.md 0 --
is the authority for rows A1-A6.

Each case: input files (text, latin-1 so raw bytes survive), argv after the
binary name ("{in}"/"{out}" are substituted), and what the ELF did: exit code,
stdout, and the output file (decompressed if .gz; bulk cases keep count + md5).

    python tests/golden/make_fastfilter_golden.py   # rewrites fastfilter_golden.json
"""
import gzip
import hashlib
import json
import os
import random
import subprocess
import sys
import tempfile

ELF = "/root/reference/assemble/fastfilter"
HERE = os.path.dirname(os.path.abspath(__file__))

IN4 = (">k31_0 flag=1 multi=12.5000 len=10\nACGTACGTAC\n"
       ">k31_1 flag=0 multi=2.0000 len=4\nACGT\n"
       ">k31_2 flag=1 multi=3.0000 len=20\nACGTACGTACGTACGTACGT\n"
       ">k31_3 flag=1 multi=100.2500 len=6\nAAAAAA\n")


def bulk_fasta(n, seed=20261003):
    """SURVEY.md 8c G20/G21 generator (call order matters)."""
    rnd = random.Random(seed)
    parts = []
    for i in range(n):
        L = rnd.randint(60, 600)
        multi = rnd.choice([1.0, 2.0, 3.5, 9.9999, 10.0, 25.25, 300.0]) * rnd.random() * 2
        parts.append(f">k31_{i} flag={rnd.randint(0, 2)} multi={multi:.4f} len={L}\n")
        parts.append("".join(rnd.choices("ACGT", k=L)) + "\n")
    return "".join(parts)


def hdr(depth, name="a"):
    return f">{name} flag=1 multi={depth} len=4\nACGTA\n"


CASES = [
    # name, input text (or None), in-name, out-name, argv
    ("G1_depth_len", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "5,15", "-d", "3"]),
    ("G2_depth0", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "0"]),
    ("G3_m2", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-m", "2"]),
    ("G4_len_minus1", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "4,4", "-d", "1"]),
    ("G5_m_len_exact", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "6,6", "-m", "5"]),
    ("G6_m0", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-m", "0"]),
    ("G7_m99", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-m", "99"]),
    ("G8_gz_both", IN4, "in.fa.gz", "out.fa.gz", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "3"]),
    ("G9_strict_gt_no_trailing_nl",
     ">a f=1 multi=3.0 len=4\nACGTA\n>b f=1 multi=2.9999 len=4\nACGTA\n>c f=1 multi=3.0001 len=4\nACGTA",
     "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "3"]),
    ("G10_short_header_panics", ">a x\nACGT\n", "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "1"]),
    ("G11_short_header_depth0", ">a x\nACGT\n", "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "0"]),
    ("G12_no_d_no_m", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100"]),
    ("G13_one_length", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "5", "-d", "0"]),
    ("G14_float_depth", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "2.5"]),
    ("G15_d_and_m", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "1", "-m", "2"]),
    ("G16_empty_seq", ">e f=1 multi=5.0 len=0\n\n>n f=1 multi=5.0 len=4\nACGT\n", "in.fa", "out.fa",
     ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "1"]),
    ("G17_multiline", ">a f=1 multi=5.0 l=8\nACGT\nACGT\n>b f=1 multi=5.0 l=4\nACGT\n", "in.fa", "out.fa",
     ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "1"]),
    ("G18_crlf", ">a f=1 multi=5.0 l=4\r\nACGT\r\n", "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "4,4", "-d", "1"]),
    ("G18b_crlf_kept", ">a f=1 multi=5.0 l=4\r\nACGT\r\n>b f=1 multi=5.0 l=4\r\nACGTAC\r\n", "in.fa", "out.fa",
     ["-i", "{in}", "-o", "{out}", "-l", "3,5", "-d", "1"]),
    ("G19_missing_input", None, "nonexist.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "1"]),
    # ---- extra probes beyond the survey's list -------------------------------------------------
    ("X1_three_lengths", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "5,15,99", "-d", "3"]),
    ("X2_attached_value", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l5,15", "-d3"]),
    ("X3_equals_value", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l=5,15", "-d=3"]),
    ("X4_negative_depth", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "-5"]),
    ("X5_dup_d", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "1", "-d", "2"]),
    ("X6_unknown_flag", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "1", "-z"]),
    ("X7_missing_o", IN4, "in.fa", "out.fa", ["-i", "{in}", "-l", "0,100", "-d", "1"]),
    ("X8_positional", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "1", "extra"]),
    ("X9_version", None, "in.fa", "out.fa", ["-V"]),
    ("X10_depth_plus", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "+3"]),
    ("X11_len_plus", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "+5,15", "-d", "3"]),
    ("X12_len_bad", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "a,15", "-d", "3"]),
    ("X13_m_bad", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-m", "x"]),
    ("X14_f32_forms",
     hdr("inf", "a") + hdr("1e1", "b") + hdr("+3.5", "c") + hdr(".5", "d") + hdr("5.", "e") + hdr("3", "f") + hdr("NaN", "g")
     + hdr("2.9999999", "h") + hdr("16777217", "i"),
     "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "3"]),
    ("X15_f32_infinity_word", hdr("infinity"), "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "3"]),
    ("X16_f32_nan_lower", hdr("nan"), "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "3"]),
    ("X17_f32_bad", hdr("3.0x"), "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "3"]),
    ("X18_no_equals", ">a b c d\nACGTA\n", "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "3"]),
    ("X19_two_equals", ">a b multi=4.0=9 d\nACGTA\n", "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "3"]),
    ("X20_tabs_in_header", ">a\tb\tmulti=4.0\td\nACGTA\n", "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "3"]),
    ("X21_m_bad_header_two_recs", ">a x\nACGT\n>b y\nACGT\n", "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-m", "5"]),
    ("X22_m_bad_header_one_rec", ">a x\nACGT\n", "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-m", "5"]),
    ("X23_m_no_gt_check", "a f=1 multi=1.0 l=4\nACGT\n>b f=1 multi=2.0 l=4\nACGT\n", "in.fa", "out.fa",
     ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-m", "5"]),
    ("X24_d_skips_non_gt", "a f=1 multi=1.0 l=4\nACGTA\n>b f=1 multi=2.0 l=4\nACGTA\n", "in.fa", "out.fa",
     ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "1"]),
    ("X25_odd_trailing_line", IN4 + ">tail f=1 multi=9.0 l=1\n", "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "1"]),
    ("X26_invalid_utf8_d", ">a f=1 multi=5.0 l=4\nAC\xffGT\n", "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "1"]),
    ("X27_invalid_utf8_after_good", IN4 + ">a f=1 multi=5.0 l=4\nAC\xffGT\n", "in.fa", "out.fa",
     ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "1"]),
    ("X28_invalid_utf8_m", IN4 + ">a f=1 multi=5.0 l=4\nAC\xffGT\n", "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-m", "9"]),
    ("X29_utf8_multibyte_len", ">a f=1 multi=5.0 l=4\nACéGT\n", "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "5,5", "-d", "1"]),
    ("X30_blank_lines", "\n\n" + IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "1"]),
    ("X31_lone_cr", ">a f=1 multi=5.0 l=4\rACGT\n>b f=1 multi=5.0 l=4\nACGT\n", "in.fa", "out.fa",
     ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "1"]),
    ("X32_empty_file", "", "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "1"]),
    ("X33_len_overflow", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,99999999999999999999", "-d", "1"]),
    ("X34_depth_overflow", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "2147483648"]),
    ("X35_depth_max", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "2147483647"]),
    ("X36_gz_in_plain_out", IN4, "in.fa.gz", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-m", "3"]),
    ("X37_empty_len_piece", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "5,", "-d", "3"]),
    ("X38_help", None, "in.fa", "out.fa", ["--help"]),
    ("X39_missing_value", IN4, "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d"]),
    ("X40_unicode_space_header", ">a b f=1 multi=5.0 l=4\nACGTA\n", "in.fa", "out.fa", ["-i", "{in}", "-o", "{out}", "-l", "0,100", "-d", "1"]),
]

BULK = [
    ("B1_bulk20k_d", 20000, ["-i", "{in}", "-o", "{out}", "-l", "0,20000", "-d", "10"]),
    ("B2_bulk20k_m", 20000, ["-i", "{in}", "-o", "{out}", "-l", "200,20000", "-m", "1000"]),
    ("G20_bulk1M_d", 1000000, ["-i", "{in}", "-o", "{out}", "-l", "0,20000", "-d", "10"]),
    ("G21_bulk1M_m", 1000000, ["-i", "{in}", "-o", "{out}", "-l", "200,20000", "-m", "1000"]),
]


def run_case(tmp, text, in_name, out_name, argv):
    inp, outp = os.path.join(tmp, in_name), os.path.join(tmp, out_name)
    for p in (inp, outp):
        if os.path.exists(p):
            os.remove(p)
    if text is not None:
        raw = text.encode("latin-1") if all(ord(c) < 256 for c in text) and "é" not in text and " " not in text else text.encode("utf-8")
        if in_name.endswith(".gz"):
            with gzip.open(inp, "wb") as f:
                f.write(raw)
        else:
            with open(inp, "wb") as f:
                f.write(raw)
    args = [a.replace("{in}", inp).replace("{out}", outp) for a in argv]
    p = subprocess.run([ELF] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    out_bytes = None
    if os.path.exists(outp):
        out_bytes = open(outp, "rb").read()
        if out_name.endswith(".gz"):
            out_bytes = gzip.decompress(out_bytes) if out_bytes else b""
    return p.returncode, p.stdout, p.stderr, out_bytes


def main():
    if not os.path.exists(ELF):
        sys.exit("reference ELF not found: this script only runs in the build container")
    golden = {"elf": "assemble/fastfilter (prebuilt, MitoFlex 0.2.9)", "cases": [], "bulk": []}
    with tempfile.TemporaryDirectory() as tmp:
        for name, text, in_name, out_name, argv in CASES:
            rc, so, se, ob = run_case(tmp, text, in_name, out_name, argv)
            utf8 = text is not None and ("é" in text or " " in text)
            golden["cases"].append({
                "name": name, "input": text, "input_encoding": "utf-8" if utf8 else "latin-1",
                "in_name": in_name, "out_name": out_name, "argv": argv,
                "rc": rc, "stdout": so.decode("latin-1"),
                "stderr_head": se.decode("latin-1", "replace")[:160],
                "output": None if ob is None else ob.decode("latin-1"),
            })
            print(f"{name:34s} rc={rc:3d} stdout={so!r:14} out={'-' if ob is None else len(ob)}")
        for name, n, argv in BULK:
            text = bulk_fasta(n)
            rc, so, se, ob = run_case(tmp, text, "bulk.fa", "bulk.out.fa", argv)
            golden["bulk"].append({"name": name, "n_records": n, "seed": 20261003, "argv": argv, "rc": rc,
                                   "stdout": so.decode(), "input_bytes": len(text),
                                   "input_md5": hashlib.md5(text.encode()).hexdigest(),
                                   "output_md5": hashlib.md5(ob).hexdigest(), "output_lines": ob.count(b"\n")})
            print(f"{name:34s} rc={rc:3d} stdout={so!r} md5={hashlib.md5(ob).hexdigest()}")
    with open(os.path.join(HERE, "fastfilter_golden.json"), "w") as f:
        json.dump(golden, f, indent=1, ensure_ascii=True)


if __name__ == "__main__":
    main()
