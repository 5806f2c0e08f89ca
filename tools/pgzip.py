#!/usr/bin/env python3
"""One gzip member, compressed on many cores (what pigz does): the input is cut into slices, every slice becomes a run of
deflate blocks from a reset state that ends on a byte boundary (Z_SYNC_FLUSH), the runs are written in order behind one
gzip header, and an empty final block, the CRC-32 of the whole input and its length close the member.  Any gzip reader
takes the result for an ordinary single-member file (the reference's flate2 GzDecoder included); it is how bench.py gets a
10 GB test input compressed in seconds instead of minutes.  Matches cannot reach back across a slice border, so the file is
about a per cent larger than gzip's.

    tools/pgzip.py IN OUT.gz [--level 6] [--procs N] [--slice-mb 8]
"""
import argparse
import mmap
import os
import struct
import sys
import zlib
from concurrent.futures import ProcessPoolExecutor


def _slice(args):
    path, a, b, level = args
    with open(path, "rb") as f:
        m = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
        data = m[a:b]
        m.close()
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    return c.compress(data) + c.flush(zlib.Z_SYNC_FLUSH), zlib.crc32(data), b - a


def _crc_shift(crc, nbytes):
    """crc32(A) -> the contribution of A to crc32(A || B) with len(B) = nbytes: multiplication by x^(8 nbytes) mod P"""
    POLY = 0xEDB88320

    def mul(a, b):
        p, m = 0, 0x80000000
        while m:
            if a & m:
                p ^= b
            b = (b >> 1) ^ POLY if b & 1 else b >> 1
            m >>= 1
        return p
    r, sq, n = 0x80000000, 0x00800000, nbytes
    while n:
        if n & 1:
            r = mul(r, sq)
        sq = mul(sq, sq)
        n >>= 1
    return mul(crc, r)


def pgzip(src, dst, level=6, procs=0, slice_mb=8):
    size = os.path.getsize(src)
    step = slice_mb << 20
    jobs = [(src, a, min(size, a + step), level) for a in range(0, size, step)]
    procs = procs or min(len(jobs), os.cpu_count() or 1) or 1
    crc = 0
    with open(dst, "wb") as out:
        out.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03")
        if jobs:
            with ProcessPoolExecutor(procs) as ex:
                for blob, c, n in ex.map(_slice, jobs, chunksize=1):
                    out.write(blob)
                    crc = _crc_shift(crc, n) ^ c          # crc32(A || B) from crc32(A), crc32(B), len(B)
        out.write(b"\x03\x00")                            # an empty final block (fixed Huffman, end of block)
        out.write(struct.pack("<II", crc & 0xFFFFFFFF, size & 0xFFFFFFFF))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("src"); ap.add_argument("dst")
    ap.add_argument("--level", type=int, default=6); ap.add_argument("--procs", type=int, default=0); ap.add_argument("--slice-mb", type=int, default=8)
    a = ap.parse_args()
    pgzip(a.src, a.dst, a.level, a.procs, a.slice_mb)
