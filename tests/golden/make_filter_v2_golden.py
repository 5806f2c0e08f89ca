#!/usr/bin/env python3
"""Capture golden vectors from the reference's prebuilt `filter/filter_v2` ELF (build container only).
Source: filter/filter_bin/src/main.rs, helper.rs (Rust; cannot be rebuilt here: no cargo).
Each case: input FASTQ texts (latin-1), argv ("{in1}" "{in2}" "{out1}" "{out2}" substituted; stdin used when
"stdin" is set), and what the ELF did: exit code, stdout, output payloads (gunzipped for .gz names)."""
import gzip
import hashlib
import json
import os
import random
import subprocess
import sys
import tempfile

ELF = "/root/reference/filter/filter_v2"
HERE = os.path.dirname(os.path.abspath(__file__))


def fq(seqs, quals=None, prefix="r", eol="\n", plus="+"):
    out = []
    for i, s in enumerate(seqs):
        q = quals[i] if quals else "I" * len(s)
        out.append(f"@{prefix}{i} d{eol}{s}{eol}{plus}{eol}{q}{eol}")
    return "".join(out)


def rand_pair(n, seed, L=40, lowq=0.04, nfrac=0.1, dup=0.2):
    rng = random.Random(seed)
    s1, s2, q1, q2 = [], [], [], []
    for i in range(n):
        if s1 and rng.random() < dup:
            j = rng.randrange(len(s1)); a = s1[j]
        else:
            a = "".join(rng.choices("ACGT", k=rng.choice([L, L, L - 7, L + 9])))
        b = "".join(rng.choices("ACGT", k=len(a) if rng.random() < 0.8 else rng.choice([L, L + 3])))
        if rng.random() < nfrac:
            a = list(a)
            for _ in range(rng.randint(1, 14)):
                a[rng.randrange(len(a))] = rng.choice("Nn")
            a = "".join(a)
        if rng.random() < nfrac:
            b = list(b)
            for _ in range(rng.randint(1, 14)):
                b[rng.randrange(len(b))] = "N"
            b = "".join(b)
        mk = lambda s: "".join(rng.choice("#*57") if rng.random() < (lowq if rng.random() < 0.8 else 0.4) else rng.choice("8FIJ") for _ in s)
        s1.append(a); s2.append(b); q1.append(mk(a)); q2.append(mk(b))
    return s1, s2, q1, q2


S1, S2, Q1, Q2 = rand_pair(60, 1)
F1, F2 = fq(S1, Q1, "a"), fq(S2, Q2, "b")
PE = ["-1", "{in1}", "-2", "{in2}", "-3", "{out1}", "-4", "{out2}"]
SE = ["-1", "{in1}", "-3", "{out1}"]

CASES = [
    ("pe_default", F1, F2, PE),
    ("se_default", F1, None, SE),
    ("pe_dedup", F1, F2, PE + ["-d"]),
    ("pe_q_l_n", F1, F2, PE + ["-q", "60", "-l", "0.1", "-n", "2"]),
    ("se_q_l_n", F1, None, SE + ["-q", "42", "-l", "0.3", "-n", "0"]),
    ("pe_cut", F1, F2, PE + ["-s", "3", "-e", "30"]),
    ("se_cut", F1, None, SE + ["-s", "5", "-e", "25", "-n", "1"]),
    ("pe_cut_dedup", F1, F2, PE + ["-s", "2", "-e", "20", "-d"]),
    ("pe_end_only", F1, F2, PE + ["-e", "10"]),
    ("pe_start_only_panics", F1, F2, PE + ["-s", "3"]),
    ("pe_start_gt_len_panics", F1, F2, PE + ["-s", "45", "-e", "60"]),
    ("se_start_eq_len", fq(["ACGTACGT", "ACGTAC"]), None, SE + ["-s", "6", "-e", "9"]),
    ("pe_trim", F1, F2, PE + ["-t", "400"]),
    ("se_trim", F1, None, SE + ["-t", "333"]),
    ("pe_trim_exact", fq(["ACGTA"] * 4), fq(["TTTTT"] * 4), PE + ["-t", "10"]),
    ("pe_truncate_only", F1, F2, PE + ["--truncate_only", "-t", "500"]),
    ("se_truncate_only_cut", F1, None, SE + ["--truncate_only", "-s", "1", "-e", "12"]),
    ("pe_long_options", F1, F2, ["--fastq1", "{in1}", "--fastq2", "{in2}", "--cleanq1", "{out1}", "--cleanq2", "{out2}",
                                 "--quality", "60", "--limit", "0.25", "--nvalues", "3", "--trim", "1000", "--deduplication"]),
    ("pe_equals_options", F1, F2, ["--fastq1={in1}", "--fastq2={in2}", "--cleanq1={out1}", "--cleanq2={out2}", "-q=60", "-l0.3"]),
    ("pe_unequal_files", fq(S1[:20], Q1[:20], "a"), fq(S2[:13], Q2[:13], "b"), PE),
    ("pe_crlf_partial", fq(S1[:10], Q1[:10], "a", eol="\r\n") + "@x\r\nACGT\r\n", fq(S2[:10], Q2[:10], "b"), PE),
    ("se_no_trailing_newline", fq(S1[:5], Q1[:5], "a").rstrip("\n"), None, SE),
    ("se_plus_line_ignored", fq(S1[:5], Q1[:5], "a", plus="+something"), None, SE),
    ("pe_gz", F1, F2, ["-1", "{in1gz}", "-2", "{in2gz}", "-3", "{out1gz}", "-4", "{out2gz}"]),
    ("se_stdin", F1, None, ["-3", "{out1}"]),
    ("pe_out2_stdout", F1, F2, ["-1", "{in1}", "-2", "{in2}", "-3", "{out1}"]),
    ("pe_lower_n_not_counted", fq(["nnnnnnnnnnnnACGT", "NNNNNNNNNNNNACGT"]), fq(["ACGTACGTACGTACGT"] * 2), PE),
    ("pe_qual_len_mismatch", fq(["ACGTACGTAC"] * 2, ["II", "#########I"]), fq(["ACGTACGTAC"] * 2, ["IIIIIIIIII", "I"]), PE + ["-l", "0.5"]),
    ("pe_cutoff_zero", fq(["ACG"] * 3, ["III", "#II", "II#"]), fq(["ACG"] * 3), PE + ["-l", "0.3"]),
    ("se_cutoff_float", fq(["ACGTA"] * 3, ["#IIII", "##III", "IIIII"]), None, SE + ["-l", "0.39"]),
    ("bad_quality_0", F1, None, SE + ["-q", "0"]),
    ("bad_quality_101", F1, None, SE + ["-q", "101"]),
    ("bad_quality_300", F1, None, SE + ["-q", "300"]),
    ("bad_quality_text", F1, None, SE + ["-q", "x"]),
    ("bad_limit_0", F1, None, SE + ["-l", "0"]),
    ("bad_limit_1", F1, None, SE + ["-l", "1.0"]),
    ("bad_limit_text", F1, None, SE + ["-l", "abc"]),
    ("limit_nan", F1, None, SE + ["-l", "NaN"]),
    ("bad_start", F1, None, SE + ["-s", "x"]),
    ("bad_end", F1, None, SE + ["-e", "-1"]),
    ("bad_n", F1, None, SE + ["-n", "x"]),
    ("bad_trim", F1, None, SE + ["-t", "1.5"]),
    ("missing_cleanq1", F1, None, ["-1", "{in1}"]),
    ("cleanq2_without_fastq2", F1, None, SE + ["-4", "{out2}"]),
    ("dedup_without_fastq2", F1, None, SE + ["-d"]),
    ("unknown_option", F1, None, SE + ["-z"]),
    ("missing_input", None, None, ["-1", "{in1}", "-3", "{out1}"]),
    ("invalid_utf8", fq(S1[:3], Q1[:3], "a") + "@b\nAC\xffGT\n+\nIIIII\n", None, SE),
    ("invalid_utf8_plus_line", fq(S1[:2], Q1[:2], "a") + "@b\nACGTACGA\n+\xff\nIIIIIIII\n" + fq(S1[2:4], Q1[2:4], "c"), None, SE),
    ("pe_invalid_utf8_plus_line", fq(S1[:3], Q1[:3], "a", plus="+\xfe\xff"), fq(S2[:3], Q2[:3], "b", plus="+\xc3"), PE),
    ("invalid_utf8_quality", fq(S1[:2], Q1[:2], "a") + "@b\nACGT\n+\nII\xffI\n", None, SE),
    ("empty_input", "", None, SE),
    ("pe_empty_seq", fq(["", "ACGT"], ["", "IIII"]), fq(["ACGT", ""], ["IIII", ""]), PE),
]

BULK = [
    ("bulk_pe_default", 20000, PE),
    ("bulk_pe_dedup_q", 20000, PE + ["-d", "-q", "52", "-l", "0.3", "-n", "3"]),
    ("bulk_pe_cut_trim", 20000, PE + ["-s", "4", "-e", "36", "-t", "300000"]),
    ("bulk_se", 20000, SE + ["-q", "52", "-l", "0.25"]),
]


def run_case(tmp, t1, t2, argv):
    paths = {k: os.path.join(tmp, v) for k, v in dict(in1="a_1.fq", in2="a_2.fq", out1="o_1.fq", out2="o_2.fq", in1gz="a_1.fq.gz",
                                                       in2gz="a_2.fq.gz", out1gz="o_1.fq.gz", out2gz="o_2.fq.gz").items()}
    for p in paths.values():
        if os.path.exists(p):
            os.remove(p)
    for key, t in (("in1", t1), ("in2", t2)):
        if t is not None:
            raw = t.encode("latin-1")
            open(paths[key], "wb").write(raw)
            with gzip.open(paths[key + "gz"], "wb") as f:
                f.write(raw)
    args = [a.format(**paths) for a in argv]
    use_stdin = "-1" not in argv and not any(a.startswith("--fastq1") for a in argv) and t1 is not None
    p = subprocess.run([ELF] + args, input=(t1.encode("latin-1") if use_stdin else None), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    outs = []
    for k in ("out1", "out2"):
        ob = None
        for suffix in ("", "gz"):
            pth = paths[k + suffix]
            if os.path.exists(pth):
                ob = open(pth, "rb").read()
                if suffix == "gz":
                    ob = gzip.decompress(ob) if ob else b""
        outs.append(ob)
    return p.returncode, p.stdout, p.stderr, outs


def main():
    if not os.path.exists(ELF):
        sys.exit("reference ELF not found: this script only runs in the build container")
    golden = {"elf": "filter/filter_v2 (prebuilt, MitoFlex 0.2.9)", "cases": [], "bulk": []}
    with tempfile.TemporaryDirectory() as tmp:
        for name, t1, t2, argv in CASES:
            rc, so, se, outs = run_case(tmp, t1, t2, argv)
            golden["cases"].append({"name": name, "in1": t1, "in2": t2, "argv": argv, "rc": rc, "stdout": so.decode("latin-1"),
                                    "stderr_head": se.decode("latin-1", "replace")[:120],
                                    "out1": None if outs[0] is None else outs[0].decode("latin-1"),
                                    "out2": None if outs[1] is None else outs[1].decode("latin-1")})
            print(f"{name:28s} rc={rc:3d} stdout={len(so):5d} out1={'-' if outs[0] is None else len(outs[0])} out2={'-' if outs[1] is None else len(outs[1])}")
        for name, n, argv in BULK:
            s1, s2, q1, q2 = rand_pair(n, 77, L=44)
            t1, t2 = fq(s1, q1, "a"), fq(s2, q2, "b")
            rc, so, se, outs = run_case(tmp, t1, t2 if "{in2}" in argv else None, argv)
            golden["bulk"].append({"name": name, "n": n, "seed": 77, "L": 44, "argv": argv, "rc": rc,
                                   "out1_md5": hashlib.md5(outs[0]).hexdigest(), "out1_lines": outs[0].count(b"\n"),
                                   "out2_md5": None if outs[1] is None else hashlib.md5(outs[1]).hexdigest()})
            print(f"{name:28s} rc={rc} lines={outs[0].count(10)}")
    json.dump(golden, open(os.path.join(HERE, "filter_v2_golden.json"), "w"), indent=1, ensure_ascii=True)


if __name__ == "__main__":
    main()
