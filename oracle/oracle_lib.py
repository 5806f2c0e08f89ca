"""TEST INFRASTRUCTURE ONLY -- ctypes loader for oracle/libmf_oracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this.  The product (mitoflex_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmf_oracle.so")


class Reads(C.Structure):
    _fields_ = [("words", C.POINTER(C.c_uint32)), ("n_words", C.c_uint64),
                ("offsets", C.POINTER(C.c_uint64)), ("n_reads", C.c_uint64),
                ("npos", C.POINTER(C.c_uint64)), ("n_npos", C.c_uint64)]


class Table(C.Structure):
    _fields_ = [("k", C.c_int), ("kw", C.c_int), ("slots", C.c_uint64),
                ("n_keys", C.c_uint64), ("keys", C.POINTER(C.c_uint64))]


def build() -> str:
    src = os.path.join(_HERE, "kmer_bait_oracle.c")
    if (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.mfo_pack_seqs.argtypes = [C.c_char_p, C.POINTER(C.c_uint64), C.c_uint64, C.POINTER(Reads)]
        L.mfo_pack_fastq.argtypes = [C.c_char_p, C.POINTER(Reads)]
        L.mfo_reads_free.argtypes = [C.POINTER(Reads)]
        L.mfo_table_build.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(Table)]
        L.mfo_table_build_file.argtypes = [C.c_char_p, C.c_int, C.POINTER(Table)]
        L.mfo_table_contains.argtypes = [C.POINTER(Table), C.c_uint64, C.c_uint64]
        L.mfo_table_free.argtypes = [C.POINTER(Table)]
        L.mfo_filter.argtypes = [C.POINTER(Table), C.POINTER(Reads), C.c_uint64, C.c_uint64, C.c_uint32,
                                 C.c_void_p, C.c_void_p, C.c_int]
        L.mfo_filter_fastq_files.argtypes = [C.c_char_p, C.c_int, C.c_uint32, C.c_int, C.c_char_p, C.c_char_p,
                                             C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                             C.c_int]
        L.mfo_ptable_build.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(Table)]
        L.mfo_pfilter.argtypes = [C.POINTER(Table), C.POINTER(Reads), C.c_uint64, C.c_uint64, C.c_uint32, C.c_int,
                                  C.c_void_p, C.c_void_p, C.c_int]
        L.mfo_pfilter_fastq_files.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_char_p, C.c_char_p,
                                              C.c_char_p, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                              C.c_int]
        L.mfo_hash64.argtypes = [C.c_uint64, C.c_uint64, C.c_int]
        L.mfo_hash64.restype = C.c_uint64
        _lib = L
    return _lib


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed rc={rc}")


class OracleReads:
    """Packed reads held by the C oracle; numpy views are copies."""

    def __init__(self, r: Reads):
        self._r = r

    @classmethod
    def from_seqs(cls, seqs):
        bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
        concat = b"".join(bs)
        off = np.zeros(len(bs) + 1, dtype=np.uint64)
        if bs:
            off[1:] = np.cumsum([len(b) for b in bs], dtype=np.uint64)
        r = Reads()
        _chk(lib().mfo_pack_seqs(concat, off.ctypes.data_as(C.POINTER(C.c_uint64)), len(bs), C.byref(r)), "pack_seqs")
        return cls(r)

    @classmethod
    def from_fastq(cls, path):
        r = Reads()
        _chk(lib().mfo_pack_fastq(os.fsencode(path), C.byref(r)), "pack_fastq")
        return cls(r)

    @classmethod
    def from_arrays(cls, words: np.ndarray, offsets: np.ndarray, npos: np.ndarray):
        """Borrow numpy arrays (kept alive on the object); not freed by C."""
        self = cls.__new__(cls)
        self._keep = (np.ascontiguousarray(words, dtype=np.uint32),
                      np.ascontiguousarray(offsets, dtype=np.uint64),
                      np.ascontiguousarray(npos, dtype=np.uint64))
        w, o, p = self._keep
        r = Reads()
        r.words = w.ctypes.data_as(C.POINTER(C.c_uint32)); r.n_words = int((int(o[-1]) + 15) // 16)
        r.offsets = o.ctypes.data_as(C.POINTER(C.c_uint64)); r.n_reads = len(o) - 1
        r.npos = p.ctypes.data_as(C.POINTER(C.c_uint64)); r.n_npos = len(p)
        self._r = r
        self._borrowed = True
        return self

    @property
    def n_reads(self): return int(self._r.n_reads)
    @property
    def words(self): return np.ctypeslib.as_array(self._r.words, shape=(int(self._r.n_words) + 8,)).copy() \
        if not getattr(self, "_borrowed", False) else self._keep[0]
    @property
    def offsets(self): return np.ctypeslib.as_array(self._r.offsets, shape=(self.n_reads + 1,)).copy()
    @property
    def npos(self):
        n = int(self._r.n_npos)
        return np.ctypeslib.as_array(self._r.npos, shape=(n,)).copy() if n else np.zeros(0, np.uint64)

    def __del__(self):
        if not getattr(self, "_borrowed", False) and self._r.words:
            lib().mfo_reads_free(C.byref(self._r))


class OracleTable:
    def __init__(self, fasta_text: str | bytes, k: int, protein: bool = False):
        """protein=True: `fasta_text` is a protein FASTA and k the peptide k-mer length (Spec P)."""
        if isinstance(fasta_text, str):
            fasta_text = fasta_text.encode()
        self._t = Table()
        build = lib().mfo_ptable_build if protein else lib().mfo_table_build
        _chk(build(fasta_text, len(fasta_text), k, C.byref(self._t)), "table_build")

    k = property(lambda s: s._t.k)
    kw = property(lambda s: s._t.kw)
    slots = property(lambda s: int(s._t.slots))
    n_keys = property(lambda s: int(s._t.n_keys))

    @property
    def keys(self) -> np.ndarray:
        return np.ctypeslib.as_array(self._t.keys, shape=(self.slots * self.kw,)).copy()

    def contains(self, code: int) -> bool:
        return bool(lib().mfo_table_contains(C.byref(self._t), code & ((1 << 64) - 1), code >> 64))

    def __del__(self):
        if self._t.keys:
            lib().mfo_table_free(C.byref(self._t))


def filter_reads(table: OracleTable, reads: OracleReads, threshold=1, first=0, count=None, threads=1):
    """-> (bits u32[ceil(n/32)], hits u32[n])"""
    n = reads.n_reads - first if count is None else count
    bits = np.zeros((n + 31) // 32, dtype=np.uint32)
    hits = np.zeros(max(n, 1), dtype=np.uint32)
    _chk(lib().mfo_filter(C.byref(table._t), C.byref(reads._r), first, n, threshold,
                          bits.ctypes.data, hits.ctypes.data, threads), "filter")
    return bits, hits[:n]


def filter_fastq_files(bait, k, threshold, pair_mode, fq1, fq2, out1, out2, threads=1):
    kept, total = C.c_uint64(0), C.c_uint64(0)
    enc = lambda p: None if p is None else os.fsencode(p)
    _chk(lib().mfo_filter_fastq_files(enc(bait), k, threshold, pair_mode, enc(fq1), enc(fq2), enc(out1), enc(out2),
                                      C.byref(kept), C.byref(total), threads), "filter_fastq_files")
    return kept.value, total.value


def pfilter_reads(table: OracleTable, reads: OracleReads, genetic_code: int, threshold=1, first=0, count=None, threads=1):
    """Protein-space baiting (Spec P): -> (bits u32[ceil(n/32)], hits u32[n])"""
    n = reads.n_reads - first if count is None else count
    bits = np.zeros((n + 31) // 32, dtype=np.uint32)
    hits = np.zeros(max(n, 1), dtype=np.uint32)
    _chk(lib().mfo_pfilter(C.byref(table._t), C.byref(reads._r), first, n, threshold, genetic_code,
                           bits.ctypes.data, hits.ctypes.data, threads), "pfilter")
    return bits, hits[:n]


def pfilter_fastq_files(bait, kp, genetic_code, threshold, pair_mode, fq1, fq2, out1, out2, threads=1):
    kept, total = C.c_uint64(0), C.c_uint64(0)
    enc = lambda p: None if p is None else os.fsencode(p)
    _chk(lib().mfo_pfilter_fastq_files(enc(bait), kp, genetic_code, threshold, pair_mode, enc(fq1), enc(fq2), enc(out1),
                                       enc(out2), C.byref(kept), C.byref(total), threads), "pfilter_fastq_files")
    return kept.value, total.value
