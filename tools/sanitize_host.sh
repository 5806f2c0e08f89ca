#!/bin/bash
# Host-side code under ASan+UBSan and TSan (CPU build only; GPU sanitizers are not available on the pool).
# Runs the packer harness and the gzip decoders (serial, parallel, BGZF, multi-member, damaged) on generated data.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/mitoflex_amd/csrc; T=${TMPDIR:-/tmp}/mf_sanitize; mkdir -p $T
for san in "address,undefined" "thread"; do
  tag=${san%%,*}
  g++ -O1 -g -fsanitize=$san -fno-omit-frame-pointer -std=c++17 -I $C $R/tests/native/pack_check.cpp $C/mf_host.cpp -lz -lpthread -o $T/pack_$tag
  g++ -O1 -g -fsanitize=$san -fno-omit-frame-pointer -std=c++17 -I $C $R/tests/native/inflate_check.cpp $C/mf_inflate.cpp $C/mf_pinflate.cpp -lz -lpthread -o $T/inflate_$tag
  g++ -O1 -g -fsanitize=$san -fno-omit-frame-pointer -std=c++17 -I $C $R/tests/native/simd_check.cpp $C/mf_inflate.cpp $C/mf_pinflate.cpp -lz -lpthread -o $T/simd_$tag
  $T/pack_$tag; MF_NO_SIMD=1 $T/pack_$tag
  $T/simd_$tag 1; MF_NO_SIMD=1 $T/simd_$tag 2
  MF_REPO=$R python3 - "$T" "$tag" <<'PY'
import sys, random, subprocess, zlib
T, tag = sys.argv[1], sys.argv[2]
sys.path.insert(0, __import__("os").environ["MF_REPO"])
from tests.test_inflate import fastq_like, gz_member, bgzf
rng = random.Random(11); text = fastq_like(rng, 6000)
cases = {"l6": (gz_member(text, 6), text), "stored": (gz_member(text[:200000], 0), text[:200000]), "fixed": (gz_member(text[:100000], 6, zlib.Z_FIXED), text[:100000]),
         "multi": (gz_member(text[:300000], 6) + gz_member(b"") + gz_member(text[300000:], 9), text), "bgzf": (bgzf(text), text),
         "truncated": (gz_member(text, 6)[:50000], None)}
bad = 0
for name, (blob, raw) in cases.items():
    open(f"{T}/a.gz", "wb").write(blob)
    if raw is not None: open(f"{T}/a.raw", "wb").write(raw)
    for args in (["65536"], ["4099"], ["65536", "--parallel", "4", "16384"], ["1048576", "--parallel", "8", "65536"]):
        p = subprocess.run([f"{T}/inflate_{tag}", f"{T}/a.gz", f"{T}/a.raw" if raw is not None else "-"] + args, capture_output=True)
        out, err = p.stdout.decode().strip(), p.stderr.decode()
        good = (out.startswith("ok") if raw is not None else out.startswith("error")) and "Sanitizer" not in err and "runtime error" not in err
        if not good: bad += 1; print("FAIL", name, args, out, err[:400])
print(tag, "decoders:", "clean" if not bad else f"{bad} problems")
sys.exit(1 if bad else 0)
PY
done
rm -rf $T
echo "sanitizers clean"
