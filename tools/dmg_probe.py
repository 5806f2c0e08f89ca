import sys, os, zlib, random
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from mitoflex_amd import mitofilter as mf
from tests.util_data import make_bait, make_reads
from tests.test_gpu_devingest import fastq_text, gz_bytes
bait_text = make_bait()
open("/tmp/dmg_bait.fa", "w").write(bait_text)
ks = mf.KmerSet.from_fasta("/tmp/dmg_bait.fa", 31)
s = make_reads(bait_text, 3000, seed=71)
good = gz_bytes(fastq_text(s, "d"), 6)
name = sys.argv[1]
data = bytearray(good)
if name == "flip": data[len(good) // 2] ^= 0x10
elif name == "crc": data[-8] ^= 1
elif name == "len": data[-1] ^= 1
elif name == "cut": data = bytearray(good[:len(good) * 2 // 3])
open("/tmp/dmg.fq.gz", "wb").write(bytes(data))
try:
    print(name, mf.filter_fastq_files(ks, "/tmp/dmg.fq.gz", None, "/tmp/dmg_o.fq", None), flush=True)
except mf.MitoFilterError as e:
    print(name, "error:", e, flush=True)
