"""Host-side mirror of the reference call site for the filter path
(`assemble/assemble_wrapper.py`): only what sits on the path is here.

  MEGAHIT.FAST_FILTER   -> path of this package's drop-in `fastfilter`        [ref :105-108]
  MEGAHIT.filter()      -> the contig filter protocol, command for command    [ref :317-345]
  MEGAHIT.prefilter()   -> NEW: the k-mer bait read pre-filter (HIP), placed where the north
                           star puts it: on the reads, before `megahit_core buildlib`
  MEGAHIT.build_lib()   -> read-library hand-off as in the reference, with the pre-filter hooked
                           in front of it and fq1/fq2 swapped for the survivors [ref :162-200];
                           with `bait_fifo` the survivors go to `buildlib` through named pipes,
                           the way the reference feeds it gunzipped reads      [ref :176-184]

Everything else of the reference class (graph/assemble/local/iterate/finalize) drives
`megahit_core` and is out of scope.

`AssembleConf` carries the knobs the reference reads from `configurations.assemble`
(configurations.py:83,91,98,101) with the same names and defaults.
"""
from __future__ import annotations

import os
import signal
import subprocess
from os import path
from typing import Optional, Tuple

from mitoflex_amd.utility import helper
from mitoflex_amd.utility.helper import shell_call  # noqa: F401  (tests patch helper.direct_call)


class AssembleConf:
    no_filter = False       # configurations.py:83
    filter_keep = 0         # configurations.py:91
    min_length = 200        # configurations.py:98
    max_length = 20000      # configurations.py:101
    # --- additions for the read pre-filter (no reference counterpart) ---
    bait_fasta: Optional[str] = None   # nucleotide bait; None = pre-filter off
    bait_kmer = 31
    bait_threshold = 1
    bait_pair_mode = "either"
    bait_devices = 1
    prefilter_in_process = True        # ctypes (libmitofilter_hip.so) instead of the CLI
    bait_fifo = False                  # stream survivors to `buildlib` through named pipes (no FASTQ round trip)


a_conf = AssembleConf()


class MEGAHIT:
    basedir = None
    fq1 = None
    fq2 = None
    prefix = None
    threads = None

    def __init__(self, **kwargs):
        for key, value in kwargs.items():
            self.__dict__[key] = value

    # ----------------------------------------------------------------- paths
    @property
    def FAST_FILTER(self) -> str:
        return path.join(path.dirname(path.abspath(__file__)), "fastfilter")

    def _contig_prefix(self, kmer) -> str:
        return path.join(self.contig_dir, f"k{kmer}")

    # ------------------------------------------------- contig filter (ref :317-345)
    def filter(self, kmer=None, min_depth=3, min_length=0, max_length=20000,
               force_filter=False, deny_number=None) -> Tuple[int, int, int]:
        if deny_number is None:
            deny_number = a_conf.filter_keep
        kept = [0, 0, 0]
        if a_conf.no_filter and not force_filter:
            return tuple(kept)
        for slot, suffix in enumerate((".contigs.fa", ".addi.fa", ".bubble_seq.fa")):
            source = self._contig_prefix(kmer) + suffix
            if not path.exists(source):
                continue
            filtered = self._contig_prefix(kmer) + ".filtered" + suffix
            span = f"{min_length},{max_length}"
            kept[slot] = int(helper.shell_call(self.FAST_FILTER, i=source, o=filtered, l=span, d=min_depth))
            if slot == 0 and kept[slot] <= deny_number:
                # too few contigs survive the depth filter: keep a fixed number instead
                kept[slot] = int(helper.shell_call(self.FAST_FILTER, i=source, o=filtered, l=span, m=deny_number))
            helper.shell_call("mv", filtered, source)
        return tuple(kept)

    # ---------------------------------------------- read pre-filter (north star)
    def prefilter(self, bait_fasta: Optional[str] = None) -> Tuple[int, int]:
        """Screen self.fq1 / self.fq2 against the bait's canonical k-mers on the GPU and point
        fq1/fq2 at the survivors.  Returns (kept, total) reads (SE) or pairs (PE)."""
        bait = bait_fasta or a_conf.bait_fasta
        if not bait:
            raise ValueError("no bait FASTA configured")
        out1 = path.join(self.temp_dir, "baited.1.fq")
        out2 = path.join(self.temp_dir, "baited.2.fq") if self.fq2 else None
        if a_conf.prefilter_in_process:
            from mitoflex_amd import mitofilter as mf
            ks = mf.KmerSet.from_fasta(bait, a_conf.bait_kmer, 0)
            try:
                kept, total = mf.filter_fastq_files(
                    ks, self.fq1, self.fq2, out1, out2, a_conf.bait_threshold,
                    mf.PAIR_BOTH if a_conf.bait_pair_mode == "both" else mf.PAIR_EITHER, a_conf.bait_devices)
            finally:
                ks.close()
        else:
            # same stdout contract as the contig filter: one integer
            kept = int(helper.shell_call(self.FAST_FILTER, "bait", bait=bait, kmer=a_conf.bait_kmer,
                                         threshold=a_conf.bait_threshold, fq1=self.fq1, fq2=self.fq2,
                                         out1=out1, out2=out2, pair=a_conf.bait_pair_mode,
                                         devices=a_conf.bait_devices))
            total = -1
        self.fq1, self.fq2 = out1, out2
        return kept, total

    # --------------------------------------------------- read library (ref :162-200)
    def prefilter_fifo(self, bait_fasta: Optional[str] = None):
        """Start the bait filter as a background writer into two (one) named pipes and point fq1/fq2 at
        them -- the hand-off the reference uses for `gzip -dc` (ref :176-184), so survivors never touch
        the disk.  Returns the Popen; its stdout carries the kept count once the reader has drained."""
        bait = bait_fasta or a_conf.bait_fasta
        if not bait:
            raise ValueError("no bait FASTA configured")
        pipes = [path.join(self.temp_dir, "pipe.bait1")]
        if self.fq2:
            pipes.append(path.join(self.temp_dir, "pipe.bait2"))
        for p in pipes:
            os.mkfifo(p)
        opts = dict(bait=bait, kmer=a_conf.bait_kmer, threshold=a_conf.bait_threshold, fq1=self.fq1, fq2=self.fq2,
                    out1=pipes[0], out2=pipes[1] if self.fq2 else None, pair=a_conf.bait_pair_mode,
                    devices=a_conf.bait_devices)
        cmd = helper.concat_command(self.FAST_FILTER, "bait", **opts)
        proc = subprocess.Popen(cmd, shell=True, stdout=subprocess.PIPE, preexec_fn=os.setsid)
        self.fq1 = pipes[0]
        self.fq2 = pipes[1] if len(pipes) > 1 else None
        return proc

    # --------------------------------------------------- read library (ref :162-200)
    def build_lib(self):
        fifos = []
        bait_proc = None
        orig = (self.fq1, self.fq2)
        if a_conf.bait_fasta:
            if a_conf.bait_fifo:
                bait_proc = self.prefilter_fifo()
            else:
                self.prefilter()
        with open(self.read_lib, "w") as lib:
            if self.fq1 and self.fq2:
                print(*(orig if bait_proc else (self.fq1, self.fq2)), sep=",", file=lib)
                names = []
                for fq, pipe in ((self.fq1, "pipe.pe1"), (self.fq2, "pipe.pe2")):
                    if fq.endswith("gz"):
                        fifo = path.join(self.temp_dir, pipe)
                        os.mkfifo(fifo)
                        fifos.append(subprocess.Popen(f"gzip -dc {fq} > {fifo}", shell=True, preexec_fn=os.setsid))
                        names.append(fifo)
                    else:
                        names.append(fq)
                print("pe", names[0], names[1], file=lib)
            else:
                print(orig[0] if bait_proc else self.fq1, file=lib)
                name = self.fq1 if not self.fq1.endswith("gz") else path.join(self.temp_dir, "pipe.se")
                print("se", name, file=lib)
        try:
            helper.shell_call(self.MEGAHIT_CORE, "buildlib", self.read_lib, self.read_lib)
        except BaseException:
            # buildlib died: the bait filter is blocked writing into pipes nobody reads.  End its process group, reap it and
            # take the pipes away, so that a retry in the same temp_dir does not stumble over them.
            if bait_proc is not None:
                try:
                    os.killpg(bait_proc.pid, signal.SIGTERM)
                except ProcessLookupError:
                    pass
                bait_proc.wait()
            self._unlink_bait_pipes()
            raise
        if any(p.wait() != 0 for p in fifos):
            raise RuntimeError("Error occured in reading input fifos")
        if bait_proc is not None:
            out = bait_proc.communicate()[0]
            self._unlink_bait_pipes()
            if bait_proc.returncode != 0:
                raise RuntimeError("Error occured in the bait filter feeding buildlib")
            self.bait_kept = int(out.decode().strip() or 0)

    def _unlink_bait_pipes(self):
        for name in ("pipe.bait1", "pipe.bait2"):
            try:
                os.unlink(path.join(self.temp_dir, name))
            except FileNotFoundError:
                pass

    MEGAHIT_CORE = "megahit_core"
