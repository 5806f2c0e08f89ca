/* libmitofilter_hip.so -- C ABI of the MI355X-native MitoFlex read pre-filter.
 *
 * What this boundary replaces.  The reference has NO FFI on this path: its
 * only interface is the subprocess CLI `assemble/fastfilter` invoked through
 * `shell_call` (assemble/assemble_wrapper.py:317-345, utility/helper.py:35-86).
 * That CLI contract is kept by mitoflex_amd/assemble/fastfilter (built from
 * mitoflex_amd/csrc/fastfilter_main.cpp).  The functions below are the
 * additional in-process boundary BASELINE.json's north_star asks for
 * ("libmitofilter_hip.so + ctypes wrapper"), following the export set
 * recommended in SURVEY.md section 8b.  Each entry point cites the reference
 * site whose role it takes over; where the reference has none it says so.
 *
 * Conventions: plain C, caller-allocated buffers, opaque handles created and
 * destroyed by the library, return 0 on success / negative MF_E_* on failure,
 * never throws; mf_last_error() returns a thread-local message.  Safe for
 * concurrent calls on different devices; one host thread per device.
 *
 * There is no CPU fallback in this library: every compute entry point runs
 * HIP kernels on a gfx950 device and fails with MF_E_NO_DEVICE without one.
 */
#ifndef MITOFILTER_H
#define MITOFILTER_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MF_ABI_VERSION 5

enum {
    MF_OK = 0,
    MF_E_ARG = -1,        /* bad argument */
    MF_E_IO = -2,         /* cannot open / read / write a file */
    MF_E_NOMEM = -3,
    MF_E_HIP = -4,        /* a HIP call failed; see mf_last_error() */
    MF_E_NO_DEVICE = -5,  /* no usable gfx950 device */
    MF_E_FORMAT = -6      /* malformed input */
};

/* mf_filter modes */
enum {
    MF_MODE_SCREENED = 0,   /* s-mer screen kernel + exact kernel on candidates (default) */
    MF_MODE_EXHAUSTIVE = 1  /* exact kernel on every read (no screen) */
};

/* what a mf_kmerset holds */
enum {
    MF_KIND_NUCLEOTIDE = 0, /* canonical nucleotide k-mers (the north-star path) */
    MF_KIND_PROTEIN = 1     /* peptide k-mers of a protein database; reads are translated in six frames */
};

/* pair rule for mf_filter_fastq_files */
enum { MF_PAIR_EITHER = 0, MF_PAIR_BOTH = 1 };

typedef struct mf_kmerset mf_kmerset; /* bait k-mer table (+ screen structures), replicated per device */
typedef struct mf_reads mf_reads;     /* one packed read set resident in one device's HBM */

typedef struct {
    int32_t  k;            /* k-mer length, 11..63 */
    int32_t  key_words;    /* 1 (k<=32) or 2 u64 per key */
    uint64_t slots;        /* open-address table slots (pow2) */
    uint64_t n_keys;       /* distinct canonical k-mers */
    uint64_t n_windows;    /* sum over records of max(0, len-k+1) */
    int32_t  screen_s;     /* s-mer length used by the screen, 0 = screen disabled */
    int32_t  screen_stride;/* sample stride in bases */
    uint32_t bloom_words;  /* LDS bit-table size in u32 words */
    uint32_t smer_slots;   /* exact s-mer table slots */
    uint64_t n_smers;      /* distinct s-mers (both strands) */
    int32_t  kind;         /* MF_KIND_*; for MF_KIND_PROTEIN k is the peptide k-mer length (ABI 2) */
    int32_t  genetic_code; /* NCBI translation table of a protein set, else 0 (ABI 2) */
    /* ABI 5: which screen a set of this size takes (the bait-size axis) */
    uint32_t front_mode;         /* 0 LDS table only | 1 LDS table, positives looked up in front2 turn by turn | 2 every sample through
                                    front2 (+ front3), no LDS table | 3 LDS table, lone positives through front2 sixty-four at a time |
                                    4 a one-bit LDS table in front of mode 2's look-ups */
    uint32_t front2_log2_blocks; /* log2 of front2's 128-bit blocks (0: none) */
    uint32_t front3_log2_blocks; /* log2 of front3's 128-bit blocks (0: none) */
    uint32_t canonical_screen;      /* 1: the screen's tables hold one canonical key per bait s-mer (larger baits) instead of one per strand */
} mf_kmerset_info_t;

typedef struct {
    uint64_t n_reads;
    uint64_t total_bases;
    uint64_t n_invalid;    /* invalid (non-ACGT) bases */
    uint32_t uniform_len;  /* >0 when every read has this length */
    int32_t  device;
} mf_reads_info_t;

typedef struct {
    uint64_t n_reads;
    uint64_t n_pass;
    uint64_t n_candidates;   /* screened mode: reads handed to the exact kernel -- or, for threshold 1 without hit counts (the
                                finish-kernel pass), stage-1 positives that were looked up (work items, not reads) */
    float    ms_total;       /* hipEvent time per pass, first launch -> last kernel end, averaged over the passes of the call
                                (consecutive threshold-1 passes overlap: finish kernels run under the next screen kernel) */
    float    ms_screen;      /* screen kernel only (streams every packed byte once); the kernel times come from
                                events attached to the dispatches themselves (hipExtLaunchKernelGGL start/stop) */
    float    ms_mark;        /* mark kernel: finishes the screen's positives, sets candidate bits (0 in the finish-kernel pass) */
    float    ms_exact;       /* exact kernel -- or the first (run) phase of the finish kernel */
    uint64_t algorithmic_bytes; /* ceil(2*bases/8) + ceil(n_reads/8): SURVEY.md 8d byte model */
} mf_filter_stats_t;

/* ---- library ---------------------------------------------------------- */
int         mf_abi_version(void);
const char *mf_last_error(void);
/* number of visible gfx950 devices (>=0) or MF_E_HIP */
int         mf_device_count(void);
int         mf_device_name(int device, char *buf, size_t buflen);
/* hipDeviceSynchronize on `device` (bench.py's barrier bracket) */
int         mf_device_synchronize(int device);

/* ---- bait k-mer set (no reference counterpart:
 * profile/MT_database is protein, findmitoscaf/findmitoscaf.py:57) --------
 * Parses a NUCLEOTIDE FASTA on the host, then builds the canonical-k-mer
 * open-address table and the screen structures with device kernels.  The
 * table is byte-identical to the CPU oracle's (history-independent layout). */
int mf_kmerset_build_from_fasta(const char *fasta_path, int k, int device, mf_kmerset **out);
int mf_kmerset_build_from_text(const char *fasta_text, size_t len, int k, int device, mf_kmerset **out);
/* Protein-space bait set (SURVEY.md 8f "next" #4): the peptide k-mers (kp residues, 4..12; 5 bits per
 * residue, first residue least significant) of a PROTEIN FASTA such as the reference's
 * profile/MT_database/<clade>.fa, which the reference itself only hands to tblastn
 * (findmitoscaf/findmitoscaf.py:57, annotation/annotation_tookit.py:55-97 `-db_gencode`).
 * genetic_code is the NCBI translation table the reads are translated with -- the reference's
 * --genetic-code / profile/codes.json value (arguments.py:449-453): 1, 2, 3, 4, 5, 9, 11, 13, 14, 21.
 * The handle goes to the same mf_filter* entry points: every read is translated in six frames,
 * hits = number of (frame, window) pairs whose kp codons are all sense codons free of invalid
 * bases and whose peptide k-mer is in the set; `mode` is ignored (there is no screen). */
int mf_kmerset_build_protein_from_fasta(const char *protein_fasta_path, int kp, int genetic_code, int device, mf_kmerset **out);
int mf_kmerset_build_protein_from_text(const char *protein_fasta_text, size_t len, int kp, int genetic_code, int device,
                                       mf_kmerset **out);
int mf_kmerset_info(const mf_kmerset *ks, mf_kmerset_info_t *info);
/* copy the device table back: keys_out must hold slots*key_words u64 */
int mf_kmerset_export(const mf_kmerset *ks, int device, uint64_t *keys_out, size_t n_u64);
int mf_kmerset_free(mf_kmerset *ks);

/* ---- packed reads (row B1).  Layout: dense little-endian 2-bit stream,
 * base i in words[i>>4] bits [2*(i&15), +1]; invalid bases stored as 0 and
 * listed ascending in npos (global base indices); offsets has n_reads+1 base
 * offsets.  The FASTQ conventions (4-line records, CR stripped, partial tail
 * dropped, .gz by extension) are the reference's: filter/filter_bin/src/
 * main.rs:287-321, filter/filter_bin/src/helper.rs:14-31. ------------------ */
int mf_reads_from_packed(const uint32_t *words, const uint64_t *offsets, uint64_t n_reads,
                         const uint64_t *npos, uint64_t n_npos, int device, mf_reads **out);
int mf_reads_from_fastq(const char *fastq_path, int device, mf_reads **out);
/* Deterministic synthetic PE150-shaped read set generated straight into the
 * packed layout (bench / large parity properties): n_reads
 * reads of read_len bases, iid uniform background; a fraction mito_ppm/1e6 of
 * reads is sampled from bait_text (random strand, sub_ppm/1e6 substitution
 * error); n_read_ppm/1e6 of reads get invalid bases at rate n_base_ppm/1e6.
 * host_words_out/host_npos_out (optional) receive malloc'd host copies that
 * the caller frees with mf_free_host(). */
int mf_reads_synth(uint64_t n_reads, uint32_t read_len, uint64_t seed,
                   const char *bait_fasta_text, size_t bait_len,
                   uint32_t mito_ppm, uint32_t sub_ppm, uint32_t n_read_ppm, uint32_t n_base_ppm,
                   int device, mf_reads **out,
                   uint32_t **host_words_out, uint64_t *host_n_words_out,
                   uint64_t **host_npos_out, uint64_t *host_n_npos_out);
/* ABI 5: the same with "real-data-shaped" extras for the low-complexity leg of bench.py: msat_ppm / 1e6 of the reads are
 * microsatellites (a random motif of 1..6 bases repeated: poly-A, (TA)n, ...), numt_ppm / 1e6 are NUMT-like reads (sampled from
 * the bait, numt_div_ppm / 1e6 substitutions per base). */
int mf_reads_synth_ex(uint64_t n_reads, uint32_t read_len, uint64_t seed,
                      const char *bait_fasta_text, size_t bait_len,
                      uint32_t mito_ppm, uint32_t sub_ppm, uint32_t n_read_ppm, uint32_t n_base_ppm,
                      uint32_t msat_ppm, uint32_t numt_ppm, uint32_t numt_div_ppm,
                      int device, mf_reads **out,
                      uint32_t **host_words_out, uint64_t *host_n_words_out,
                      uint64_t **host_npos_out, uint64_t *host_n_npos_out);
void mf_free_host(void *p);
int mf_reads_info(const mf_reads *r, mf_reads_info_t *info);
int mf_reads_free(mf_reads *r);

/* ---- the hot path (rows B2, B3, B4): k-mer extract -> canonicalise ->
 * open-address probe -> hit threshold.  Inputs already resident in HBM.
 * out_bits: host buffer of ceil(n_reads/32) u32, bit r set = read r passes
 * (hits >= threshold, threshold >= 1).  hits_out: optional host buffer of
 * n_reads u32 receiving the full hit count of every read (disables the
 * early exit; parity/debug).  stats optional. */
int mf_filter(const mf_kmerset *ks, const mf_reads *reads, uint32_t threshold, int mode,
              uint32_t *out_bits, uint32_t *hits_out, mf_filter_stats_t *stats);

/* Same, result left on the device (no D2H); for timing loops.  Runs `steps`
 * passes back to back on the library's stream.  ms_total is the whole loop
 * (one event pair around it) divided by steps; the per-kernel times are
 * averages over the passes whose dispatches carry start/stop events --
 * every pass up to 8 steps, every 8th pass beyond (a profiled dispatch costs
 * a few microseconds of command-processor work). */
int mf_filter_resident(const mf_kmerset *ks, const mf_reads *reads, uint32_t threshold, int mode,
                       int steps, mf_filter_stats_t *stats);

/* Same as mf_filter_resident, and the number of passing reads of EVERY one of the `steps` passes in n_pass_per_step[0 .. steps)
 * (each pass tallies into a block of its own): consecutive passes of a call overlap on two buffer sets, and a fault that
 * touched only the middle passes would not show in the tally of the last one.  No reference counterpart (test support for
 * the pipelined pass). */
int mf_filter_resident_passes(const mf_kmerset *ks, const mf_reads *reads, uint32_t threshold, int mode,
                              int steps, uint64_t *n_pass_per_step, mf_filter_stats_t *stats);

/* One-shot over host buffers (SURVEY.md 8b name): H2D, filter, D2H. */
int mf_filter_packed(const mf_kmerset *ks, int device,
                     const uint32_t *words, const uint64_t *offsets, uint64_t n_reads,
                     const uint64_t *npos, uint64_t n_npos,
                     uint32_t threshold, uint32_t *out_bits);

/* ---- file level: what a MitoFlex stage calls.  Takes the place
 * `bim.bwa_map` has in the reference (bim/bim.py:43-58: reads in, baited
 * reads out as FASTQ) and is hooked ahead of `megahit_core buildlib`
 * (assemble/assemble_wrapper.py:162-193).  fq2/out2 NULL = single end.
 * Records are zipped like the reference's PE reader (filter/filter_bin/src/
 * main.rs:214); survivors keep input order and are written header/seq/+/qual
 * (main.rs:261-268).  Chunks of whole pairs are dealt to n_devices GPUs (one
 * host thread each, no collective); ".gz" inputs/outputs by extension. */
int mf_filter_fastq_files(mf_kmerset *ks, const char *fq1, const char *fq2,
                          const char *out1, const char *out2,
                          uint32_t threshold, int pair_mode, int n_devices,
                          uint64_t *kept, uint64_t *total);
/* The same on a chosen list of devices (ABI 3): devices[0 .. n_devices) are device indices below
 * mf_device_count(), each at most once.  A ".gz" FILE takes the device ingest path: its bytes are
 * copied up as they are, the slabs of the stream are dealt to the listed devices round robin and
 * inflate, line index, 2-bit pack, filter and the copy of the survivors run there, with a bounded
 * amount of device memory whatever the file's size (the reference's readers stream too:
 * filter/filter_bin/src/helper.rs:14-31).  Plain files, pipes and BGZF take the host pipeline,
 * which wants the list to be a run of consecutive devices.  mf_filter_fastq_files(.., n, ..) is
 * this call with devices 0 .. n - 1. */
int mf_filter_fastq_files_on(mf_kmerset *ks, const char *fq1, const char *fq2,
                             const char *out1, const char *out2,
                             uint32_t threshold, int pair_mode,
                             const int *devices, int n_devices,
                             uint64_t *kept, uint64_t *total);

/* Options that select which kernels a filter pass runs (process-wide; every variant gives the same bits and is parity-tested):
 *   pass=default|split|serial   adapt=0|1   finish_streams=0|1|2   screen_streams=1|2   split_pipe=0|1   exact_co=0|1
 * and, read when a k-mer set is BUILT (ABI 5; every form gives the same bits -- tests force them on small baits):
 *   front=-1|0|1|2|3|4 (which screen; -1: by the bait's size)   canon=-1|0|1 (one canonical key per bait s-mer in the screen's tables)
 *   s8_finish=-1|0|1 (k < 28: threshold-1 passes through screen + finish)   front2_log2b=0|6..24   front3_log2b=-1|0|6..27
 *   (MF_FRONT, MF_CANON, MF_S8_FINISH, MF_FRONT2_LOG2B, MF_FRONT3_LOG2B under MF_ENV_KNOBS=1); expect_files=0|1, short_lived=0|1 (ABI 4).
 * The library does NOT take these from the environment in a production process: the MF_PASS, MF_ADAPT, MF_FINISH_STREAMS,
 * MF_SCREEN_STREAMS, MF_SPLIT_PIPE and MF_EXACT_CO variables only count when MF_ENV_KNOBS=1 is set beside them (tests, bench.py,
 * profiling scripts).  ABI 3. */
int mf_set_option(const char *name, const char *value);

/* What the calling thread's last successful mf_filter_fastq_files / _on call did (ABI 3).  path: which
 * of the library's two ingest paths took the input. */
enum { MF_INGEST_PATH_HOST = 0, MF_INGEST_PATH_DEVICE = 1 };
typedef struct {
    int32_t  path, n_devices, consumers, reserved;
    uint64_t input_bytes;          /* bytes of the input files as they lie (compressed, if .gz) -- device path only */
    uint64_t text_bytes;           /* bytes of FASTQ text they hold -- device path only */
    uint64_t records;              /* FASTQ records cut from it, both mates */
    double   seconds;              /* wall time of the call */
    double   decode_busy_seconds;  /* time with at least one inflate kernel running, summed over devices and mates (0: no .gz) */
    uint64_t pool_bytes_peak;      /* this call's device buffers at most, on any one device */
    uint64_t device_bytes_peak;    /* everything in use on a device at most (hipMemGetInfo after each piece) */
    uint64_t chunks, chunks_linked, gaps, gap_bytes;   /* speculative chunks; those the link step took; stretches (bytes of text) the host bridged */
} mf_ingest_stats_t;
int mf_last_ingest_stats(mf_ingest_stats_t *out);

/* The device ingest path keeps device buffers, pinned staging buffers and the consumers' read sets of a call for the process's next
 * call (allocating them anew costs a call a tenth of a second and more).  mf_release_cached() gives all of it back to the runtime;
 * *bytes (may be NULL) receives the device bytes released.  Call it between calls, not during one.  The library calls the same code by
 * itself whenever one of its own allocations finds a device full.  ABI 4. */
int mf_release_cached(uint64_t *bytes);

/* Host-to-device copy rate of this box in GB/s (pinned memory, `bytes` per copy, best of `reps`): the
 * roof of the device ingest path, whose input goes up over PCIe as it lies on disk. */
int mf_h2d_bandwidth(int device, size_t bytes, int reps, double *gb_per_s);

/* ---- FASTQ quality filter: the reference's `filter/filter_v2` (filter/filter_bin/src/main.rs:14-329),
 * the stage that runs on the raw reads before this path (SURVEY.md 8f "next" #2).  Same rules, same
 * output bytes: cut [start, end), drop reads with more than `ns` 'N' or with at least
 * (len * limit) quality bytes <= `quality`, optional de-duplication on mate 1 (first occurrence
 * kept; SipHash-1-3 like Rust's DefaultHasher), stop once `trim` bases have been kept,
 * `truncate_only` skips the tests.  fq1 NULL = standard input; out2 NULL with fq2 set = standard
 * output.  Counting and hashing run on the GPU over the raw FASTQ text; the order-dependent rules
 * are applied on the host.  Regular .gz input files take the device ingest path (inflate, line index,
 * counting, hashing, the de-duplication set, the decisions and the formatting of the kept records on
 * the GPU; mf_last_ingest_stats tells), everything else the host pipeline; MF_QUAL_INGEST=device|host
 * forces either for what both can take.  Same bytes either way.  *panicked is set when the reference would have aborted mid-file (cut
 * start beyond a string, invalid UTF-8): output up to that record is written, as the reference's
 * BufWriter flushes on unwind, and the CLI then exits 101. */
int mf_qualfilter_files(const char *fq1, const char *fq2, const char *out1, const char *out2,
                        uint64_t start, uint64_t end, uint64_t ns, uint32_t quality, float limit,
                        int dedup, uint64_t trim, int truncate_only, int device,
                        uint64_t *kept, uint64_t *total, int *panicked);

#ifdef __cplusplus
}
#endif
#endif /* MITOFILTER_H */
