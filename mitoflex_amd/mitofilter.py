"""ctypes binding of libmitofilter_hip.so (include/mitofilter.h).

Host code stays Python behind this thin shim, as BASELINE.json's north_star
asks; there is no PyTorch and no CPU fallback here: if the shared library is
missing, or no gfx950 device is visible, calls raise `MitoFilterError`.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmitofilter_hip.so")

KIND_NUCLEOTIDE, KIND_PROTEIN = 0, 1
MODE_SCREENED = 0
MODE_EXHAUSTIVE = 1
PAIR_EITHER = 0
PAIR_BOTH = 1

# every symbol include/mitofilter.h declares (checked by tests/test_abi.py)
EXPORTS = (
    "mf_abi_version", "mf_last_error", "mf_device_count", "mf_device_name", "mf_device_synchronize",
    "mf_kmerset_build_from_fasta", "mf_kmerset_build_from_text", "mf_kmerset_build_protein_from_fasta",
    "mf_kmerset_build_protein_from_text", "mf_kmerset_info", "mf_kmerset_export",
    "mf_kmerset_free", "mf_reads_from_packed", "mf_reads_from_fastq", "mf_reads_synth", "mf_reads_synth_ex", "mf_free_host",
    "mf_reads_info", "mf_reads_free", "mf_filter", "mf_filter_resident", "mf_filter_resident_passes", "mf_filter_packed",
    "mf_filter_fastq_files", "mf_filter_fastq_files_on", "mf_last_ingest_stats", "mf_h2d_bandwidth", "mf_set_option", "mf_qualfilter_files", "mf_release_cached",
)


class MitoFilterError(RuntimeError):
    pass


class KmerSetInfo(C.Structure):
    _fields_ = [("k", C.c_int32), ("key_words", C.c_int32), ("slots", C.c_uint64), ("n_keys", C.c_uint64),
                ("n_windows", C.c_uint64), ("screen_s", C.c_int32), ("screen_stride", C.c_int32),
                ("bloom_words", C.c_uint32), ("smer_slots", C.c_uint32), ("n_smers", C.c_uint64),
                ("kind", C.c_int32), ("genetic_code", C.c_int32),
                ("front_mode", C.c_uint32), ("front2_log2_blocks", C.c_uint32), ("front3_log2_blocks", C.c_uint32), ("canonical_screen", C.c_uint32)]


class ReadsInfo(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("total_bases", C.c_uint64), ("n_invalid", C.c_uint64),
                ("uniform_len", C.c_uint32), ("device", C.c_int32)]


class FilterStats(C.Structure):
    _fields_ = [("n_reads", C.c_uint64), ("n_pass", C.c_uint64), ("n_candidates", C.c_uint64),
                ("ms_total", C.c_float), ("ms_screen", C.c_float), ("ms_mark", C.c_float), ("ms_exact", C.c_float),
                ("algorithmic_bytes", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class IngestStats(C.Structure):
    _fields_ = [("path", C.c_int32), ("n_devices", C.c_int32), ("consumers", C.c_int32), ("reserved", C.c_int32),
                ("input_bytes", C.c_uint64), ("text_bytes", C.c_uint64), ("records", C.c_uint64),
                ("seconds", C.c_double), ("decode_busy_seconds", C.c_double),
                ("pool_bytes_peak", C.c_uint64), ("device_bytes_peak", C.c_uint64),
                ("chunks", C.c_uint64), ("chunks_linked", C.c_uint64), ("gaps", C.c_uint64), ("gap_bytes", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "reserved"}


_lib = None


def load(path: Optional[str] = None):
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("MITOFILTER_LIB", LIB_PATH)
    if not os.path.exists(path):
        raise MitoFilterError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(there is no CPU fallback)")
    L = C.CDLL(path)
    vp, u32p, u64p = C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    L.mf_abi_version.restype = C.c_int
    L.mf_last_error.restype = C.c_char_p
    L.mf_device_count.restype = C.c_int
    L.mf_device_name.argtypes = [C.c_int, C.c_char_p, C.c_size_t]
    L.mf_device_synchronize.argtypes = [C.c_int]
    L.mf_kmerset_build_from_fasta.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(vp)]
    L.mf_kmerset_build_from_text.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(vp)]
    L.mf_kmerset_build_protein_from_fasta.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.mf_kmerset_build_protein_from_text.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.mf_kmerset_info.argtypes = [vp, C.POINTER(KmerSetInfo)]
    L.mf_kmerset_export.argtypes = [vp, C.c_int, vp, C.c_size_t]
    L.mf_kmerset_free.argtypes = [vp]
    L.mf_reads_from_packed.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, C.c_int, C.POINTER(vp)]
    L.mf_reads_from_fastq.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
    L.mf_reads_synth.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_char_p, C.c_size_t,
                                 C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(vp),
                                 C.POINTER(u32p), u64p, C.POINTER(u64p), u64p]
    L.mf_reads_synth_ex.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_char_p, C.c_size_t,
                                    C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(vp),
                                    C.POINTER(u32p), u64p, C.POINTER(u64p), u64p]
    L.mf_free_host.argtypes = [vp]
    L.mf_free_host.restype = None
    L.mf_reads_info.argtypes = [vp, C.POINTER(ReadsInfo)]
    L.mf_reads_free.argtypes = [vp]
    L.mf_filter.argtypes = [vp, vp, C.c_uint32, C.c_int, vp, vp, C.POINTER(FilterStats)]
    L.mf_filter_resident.argtypes = [vp, vp, C.c_uint32, C.c_int, C.c_int, C.POINTER(FilterStats)]
    L.mf_filter_resident_passes.argtypes = [vp, vp, C.c_uint32, C.c_int, C.c_int, vp, C.POINTER(FilterStats)]
    L.mf_filter_packed.argtypes = [vp, C.c_int, vp, vp, C.c_uint64, vp, C.c_uint64, C.c_uint32, vp]
    L.mf_filter_fastq_files.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_int,
                                        C.c_int, u64p, u64p]
    L.mf_filter_fastq_files_on.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_int,
                                           C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.mf_set_option.argtypes = [C.c_char_p, C.c_char_p]
    L.mf_last_ingest_stats.argtypes = [C.POINTER(IngestStats)]
    L.mf_h2d_bandwidth.argtypes = [C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
    L.mf_release_cached.argtypes = [u64p]
    L.mf_qualfilter_files.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64,
                                      C.c_uint32, C.c_float, C.c_int, C.c_uint64, C.c_int, C.c_int, u64p, u64p,
                                      C.POINTER(C.c_int)]
    if L.mf_abi_version() != 5:
        raise MitoFilterError("libmitofilter_hip ABI version mismatch")
    _lib = L
    return L


def _chk(rc: int):
    if rc != 0:
        raise MitoFilterError(f"libmitofilter_hip error {rc}: {load().mf_last_error().decode(errors='replace')}")


def device_count() -> int:
    n = load().mf_device_count()
    if n < 0:
        _chk(n)
    return n


def device_name(device: int = 0) -> str:
    buf = C.create_string_buffer(256)
    _chk(load().mf_device_name(device, buf, 256))
    return buf.value.decode()


def device_synchronize(device: int = 0):
    _chk(load().mf_device_synchronize(device))


def _enc(p):
    return None if p is None else os.fsencode(p)


class KmerSet:
    """Bait canonical-k-mer set resident on the GPU (rows B3/B5)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def from_fasta(cls, path: str, k: int = 31, device: int = 0) -> "KmerSet":
        h = C.c_void_p()
        _chk(load().mf_kmerset_build_from_fasta(_enc(path), k, device, C.byref(h)))
        return cls(h)

    @classmethod
    def from_text(cls, fasta_text, k: int = 31, device: int = 0) -> "KmerSet":
        if isinstance(fasta_text, str):
            fasta_text = fasta_text.encode()
        h = C.c_void_p()
        _chk(load().mf_kmerset_build_from_text(fasta_text, len(fasta_text), k, device, C.byref(h)))
        return cls(h)

    @classmethod
    def protein_from_fasta(cls, path: str, kp: int = 9, genetic_code: int = 5, device: int = 0) -> "KmerSet":
        """Peptide k-mer set of a protein FASTA (e.g. the reference's profile/MT_database/<clade>.fa); reads
        filtered against it are translated in six frames with NCBI table `genetic_code`."""
        h = C.c_void_p()
        _chk(load().mf_kmerset_build_protein_from_fasta(_enc(path), kp, genetic_code, device, C.byref(h)))
        return cls(h)

    @classmethod
    def protein_from_text(cls, fasta_text, kp: int = 9, genetic_code: int = 5, device: int = 0) -> "KmerSet":
        if isinstance(fasta_text, str):
            fasta_text = fasta_text.encode()
        h = C.c_void_p()
        _chk(load().mf_kmerset_build_protein_from_text(fasta_text, len(fasta_text), kp, genetic_code, device, C.byref(h)))
        return cls(h)

    @property
    def info(self) -> KmerSetInfo:
        i = KmerSetInfo()
        _chk(load().mf_kmerset_info(self._h, C.byref(i)))
        return i

    def export_table(self, device: int = 0) -> np.ndarray:
        i = self.info
        out = np.empty(i.slots * i.key_words, dtype=np.uint64)
        _chk(load().mf_kmerset_export(self._h, device, out.ctypes.data, out.size))
        return out

    def close(self):
        if self._h:
            load().mf_kmerset_free(self._h)
            self._h = None

    __del__ = close


class Reads:
    """One packed read set resident in one GPU's HBM (row B1)."""

    def __init__(self, handle):
        self._h = handle
        self.host_words: Optional[np.ndarray] = None
        self.host_npos: Optional[np.ndarray] = None

    @classmethod
    def from_packed(cls, words: np.ndarray, offsets: np.ndarray, npos: np.ndarray, device: int = 0) -> "Reads":
        words = np.ascontiguousarray(words, dtype=np.uint32)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        npos = np.ascontiguousarray(npos, dtype=np.uint64)
        h = C.c_void_p()
        _chk(load().mf_reads_from_packed(words.ctypes.data, offsets.ctypes.data, len(offsets) - 1,
                                         npos.ctypes.data if npos.size else None, npos.size, device, C.byref(h)))
        return cls(h)

    @classmethod
    def from_fastq(cls, path: str, device: int = 0) -> "Reads":
        h = C.c_void_p()
        _chk(load().mf_reads_from_fastq(_enc(path), device, C.byref(h)))
        return cls(h)

    @classmethod
    def synth(cls, n_reads: int, read_len: int, seed: int, bait_text, mito_ppm=5000, sub_ppm=10000,
              n_read_ppm=10000, n_base_ppm=1000, device: int = 0, keep_host: bool = False,
              msat_ppm: int = 0, numt_ppm: int = 0, numt_div_ppm: int = 150000) -> "Reads":
        if isinstance(bait_text, str):
            bait_text = bait_text.encode()
        h = C.c_void_p()
        L = load()
        if keep_host:
            wp, np_ = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint64)()
            nw, nn = C.c_uint64(), C.c_uint64()
            _chk(L.mf_reads_synth_ex(n_reads, read_len, seed, bait_text, len(bait_text), mito_ppm, sub_ppm, n_read_ppm,
                                     n_base_ppm, msat_ppm, numt_ppm, numt_div_ppm, device, C.byref(h), C.byref(wp), C.byref(nw), C.byref(np_), C.byref(nn)))
            self = cls(h)
            self.host_words = np.ctypeslib.as_array(wp, shape=(nw.value + 8,)).copy()
            self.host_npos = (np.ctypeslib.as_array(np_, shape=(nn.value,)).copy() if nn.value
                              else np.zeros(0, np.uint64))
            L.mf_free_host(wp)
            L.mf_free_host(np_)
            return self
        _chk(L.mf_reads_synth_ex(n_reads, read_len, seed, bait_text, len(bait_text), mito_ppm, sub_ppm, n_read_ppm,
                                 n_base_ppm, msat_ppm, numt_ppm, numt_div_ppm, device, C.byref(h), None, None, None, None))
        return cls(h)

    @property
    def info(self) -> ReadsInfo:
        i = ReadsInfo()
        _chk(load().mf_reads_info(self._h, C.byref(i)))
        return i

    def close(self):
        if self._h:
            load().mf_reads_free(self._h)
            self._h = None

    __del__ = close


def filter_reads(ks: KmerSet, reads: Reads, threshold: int = 1, mode: int = MODE_SCREENED,
                 want_hits: bool = False) -> Tuple[np.ndarray, Optional[np.ndarray], FilterStats]:
    """Run the hot path once.  -> (bits u32[ceil(n/32)], hits u32[n] | None, stats)."""
    n = reads.info.n_reads
    bits = np.zeros((n + 31) // 32, dtype=np.uint32)
    hits = np.zeros(max(n, 1), dtype=np.uint32) if want_hits else None
    st = FilterStats()
    _chk(load().mf_filter(ks._h, reads._h, threshold, mode, bits.ctypes.data,
                          hits.ctypes.data if want_hits else None, C.byref(st)))
    return bits, (hits[:n] if want_hits else None), st


def filter_resident(ks: KmerSet, reads: Reads, threshold: int = 1, mode: int = MODE_SCREENED, steps: int = 1) -> FilterStats:
    st = FilterStats()
    _chk(load().mf_filter_resident(ks._h, reads._h, threshold, mode, steps, C.byref(st)))
    return st


def filter_resident_passes(ks: KmerSet, reads: Reads, threshold: int = 1, mode: int = MODE_SCREENED, steps: int = 1):
    """`steps` passes back to back; -> (passing reads of every pass, stats)"""
    st = FilterStats()
    per = np.zeros(steps, dtype=np.uint64)
    _chk(load().mf_filter_resident_passes(ks._h, reads._h, threshold, mode, steps, per.ctypes.data, C.byref(st)))
    return per, st


def filter_packed(ks: KmerSet, words, offsets, npos, threshold: int = 1, device: int = 0) -> np.ndarray:
    words = np.ascontiguousarray(words, dtype=np.uint32)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    npos = np.ascontiguousarray(npos, dtype=np.uint64)
    n = len(offsets) - 1
    bits = np.zeros((n + 31) // 32, dtype=np.uint32)
    _chk(load().mf_filter_packed(ks._h, device, words.ctypes.data, offsets.ctypes.data, n,
                                 npos.ctypes.data if npos.size else None, npos.size, threshold, bits.ctypes.data))
    return bits


def filter_fastq_files(ks: KmerSet, fq1: str, fq2: Optional[str], out1: str, out2: Optional[str],
                       threshold: int = 1, pair_mode: int = PAIR_EITHER, n_devices: int = 1,
                       devices: Optional[Sequence[int]] = None) -> Tuple[int, int]:
    """-> (kept, total) reads (SE) or pairs (PE).  devices: an explicit list of device indices (instead of 0 .. n_devices - 1)."""
    kept, total = C.c_uint64(), C.c_uint64()
    if devices is not None:
        arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        _chk(load().mf_filter_fastq_files_on(ks._h, _enc(fq1), _enc(fq2), _enc(out1), _enc(out2), threshold, pair_mode,
                                             arr, len(devices), C.byref(kept), C.byref(total)))
    else:
        _chk(load().mf_filter_fastq_files(ks._h, _enc(fq1), _enc(fq2), _enc(out1), _enc(out2), threshold, pair_mode,
                                          n_devices, C.byref(kept), C.byref(total)))
    return kept.value, total.value


def set_option(name: str, value) -> None:
    """Process-wide switch of how a filter pass is run (pass=default|split|serial, adapt, finish_streams, screen_streams, split_pipe,
    exact_co): the library does not read these from the environment unless MF_ENV_KNOBS=1."""
    _chk(load().mf_set_option(_enc(name), _enc(str(value))))


def last_ingest_stats() -> dict:
    """What this thread's last filter_fastq_files call did: which ingest path, bytes, seconds, device memory, decoder counters."""
    st = IngestStats()
    _chk(load().mf_last_ingest_stats(C.byref(st)))
    return st.as_dict()


def release_cached() -> int:
    """Give the device buffers, pinned staging and read sets the file-level calls keep between calls back to the runtime; returns
    the device bytes released (the library does the same by itself when one of its allocations finds a device full)."""
    v = C.c_uint64()
    _chk(load().mf_release_cached(C.byref(v)))
    return v.value


def h2d_bandwidth(device: int = 0, nbytes: int = 1 << 30, reps: int = 3) -> float:
    """GB/s of a pinned host-to-device copy on this box (the roof of the device ingest path)."""
    v = C.c_double()
    _chk(load().mf_h2d_bandwidth(device, nbytes, reps, C.byref(v)))
    return v.value


def qualfilter_files(fq1: Optional[str], fq2: Optional[str], out1: str, out2: Optional[str], start: int = 0, end: int = 0,
                     ns: int = 10, quality: int = 55, limit: float = 0.2, dedup: bool = False, trim: int = 0,
                     truncate_only: bool = False, device: int = 0) -> Tuple[int, int, bool]:
    """The reference's `filter_v2` rules (filter/filter_bin/src/main.rs) with GPU counting.
    -> (kept, total, panicked)."""
    kept, total, pan = C.c_uint64(), C.c_uint64(), C.c_int()
    _chk(load().mf_qualfilter_files(_enc(fq1), _enc(fq2), _enc(out1), _enc(out2), start, end, ns, quality, limit, int(dedup),
                                    trim, int(truncate_only), device, C.byref(kept), C.byref(total), C.byref(pan)))
    return kept.value, total.value, bool(pan.value)


def unpack_bits(bits: np.ndarray, n: int) -> np.ndarray:
    """u32 bitmap -> bool[n] (bit r of word r>>5)."""
    return np.unpackbits(bits.view(np.uint8), bitorder="little")[:n].astype(bool)
