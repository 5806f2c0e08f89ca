// Host-side parsing and packing for libmitofilter_hip (no HIP in here).
//
// FASTQ conventions are the reference's (filter/filter_bin/src/main.rs:287-321,
// filter/filter_bin/src/helper.rs:14-31): strict 4-line records, line 3
// ignored, CR stripped, partial tail dropped, gzip selected by ".gz" extension.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <memory>
#include <new>
#include <stdio.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <string>
#include <utility>
#include <vector>

namespace mf {

bool has_gz_ext(const char *path);
// whole file into memory; inflates when the path ends in ".gz".  false + err on failure
bool slurp_file(const char *path, std::vector<char> &out, std::string &err);

struct BaitHost {
    std::vector<uint32_t> words;    // 2-bit LE, padded with 8 zero words
    std::vector<uint8_t>  runlen;   // valid run length from each base (cap 255), 0 at invalid bases
    uint64_t total = 0;             // bases over all records (invalid ones included)
    std::vector<uint64_t> rec_len;  // per record
    uint64_t n_windows(int k) const;
    uint64_t n_swindows(int s) const;
};
// B5: '>' at line start opens a record; whitespace in sequence lines ignored;
// anything but ACGTacgt is invalid and breaks windows; windows never span records.
void parse_bait_fasta(const char *text, size_t len, BaitHost &out);

// ---- protein-space baiting (DESIGN.md "Spec P") ------------------------------
struct ProtBaitHost {
    std::vector<uint8_t> aa;        // residue codes 0..19 (rank in "ACDEFGHIKLMNPQRSTVWY"), 0 at invalid letters, 16 B padding
    std::vector<uint8_t> runlen;    // valid run length from each residue (cap 255), 0 at invalid letters
    uint64_t total = 0;             // residues over all records (invalid ones included)
    std::vector<uint64_t> rec_len;
    uint64_t n_windows(int kp) const;
};
// same record rules as parse_bait_fasta; anything but the 20 standard residues (either case) is invalid
void parse_bait_protein(const char *text, size_t len, ProtBaitHost &out);
// NCBI translation table `genetic_code` (1, 2, 3, 4, 5, 9, 11, 13, 14, 21) as the filter kernel wants it:
// 64 entries of 4 dwords, indexed by the codon as it sits in the packed stream (first base in bits 0-1):
//   [0],[1]  forward-strand residue << 5*(kp-1) (64 bit), [2] residue of the reverse-complemented codon,
//   [3]      bit 0: the forward codon is a sense codon, bit 1: same for the reverse codon;
//   a stop codon has the non-residue code 31, which no database key holds
bool codon_lut_for(int genetic_code, int kp, uint32_t out[256]);

struct FqRec { const char *h, *s, *q; uint32_t hl, sl, ql; };
// strict 4-line records over an in-memory buffer (pointers into buf)
void parse_fastq(const char *buf, size_t len, std::vector<FqRec> &recs);

// allocator whose construct() default-initialises: a vector of it can be sized without being zero-filled.
// Large blocks are 2 MiB aligned and marked for transparent huge pages: a batch buffer of ~100 MB is then
// some fifty page faults instead of twenty-five thousand, and just as cheap to give back.
template <class T> struct DefaultInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = DefaultInitAlloc<U>; };
    T *allocate(size_t n)
    {
        const size_t bytes = n * sizeof(T), huge = (size_t)2 << 20;
        void *p;
        if (bytes >= 2 * huge) {
            const size_t rounded = (bytes + huge - 1) / huge * huge;
            p = aligned_alloc(huge, rounded);
            if (p) madvise(p, rounded, MADV_HUGEPAGE);
        } else p = malloc(bytes ? bytes : 1);
        if (!p) throw std::bad_alloc();
        return static_cast<T *>(p);
    }
    void deallocate(T *p, size_t) noexcept { free(p); }
    template <class U> void construct(U *p) noexcept { ::new ((void *)p) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new ((void *)p) U(std::forward<A>(a)...); }
};
// Buffers the device reads by DMA (packed words, offsets): when the GPU library has registered a pinned-memory allocator
// (hipHostMalloc; mf_api.cpp does, the CPU-only test builds do not) their large blocks come from it, so that the
// host-to-device copy of a batch is an asynchronous DMA straight from where the packer wrote, not a staged pageable copy.
void set_dma_allocator(void *(*alloc)(size_t), void (*release)(void *));     // (nullptr, nullptr): none; blocks handed out before stay valid
void *dma_block_alloc(size_t bytes);        // nullptr: no allocator registered (or it failed) -- use the ordinary one
bool dma_block_free(void *p);               // false: p is not one of ours
template <class T> struct DmaInitAlloc : DefaultInitAlloc<T> {
    template <class U> struct rebind { using other = DmaInitAlloc<U>; };
    T *allocate(size_t n)
    {
        const size_t bytes = n * sizeof(T);
        if (bytes >= ((size_t)4 << 20)) { if (void *p = dma_block_alloc(bytes)) return static_cast<T *>(p); }
        return DefaultInitAlloc<T>::allocate(n);
    }
    // (only blocks of at least 4 MiB can be DMA blocks: everything smaller skips the lock and the lookup)
    void deallocate(T *p, size_t n) noexcept { if (n * sizeof(T) < ((size_t)4 << 20) || !dma_block_free(p)) DefaultInitAlloc<T>::deallocate(p, n); }
};
using WordVec = std::vector<uint32_t, DmaInitAlloc<uint32_t>>;
using U64Vec = std::vector<uint64_t, DmaInitAlloc<uint64_t>>;

struct PackedHost {
    WordVec words;                  // padded (see pad_words_for); every word is written by pack_records
    uint64_t n_words = 0;
    U64Vec offsets;                 // n_reads + 1
    std::vector<uint64_t> npos;
    uint32_t uniform_len = 0;       // >0 when all reads share one length
};
// words needed so the screen kernel can walk whole chunks past the end
uint64_t padded_words_for(uint64_t n_words);
// pack records [first, first+count) with `threads` workers; `out` keeps its capacity when it is reused
void pack_records(const FqRec *recs, uint64_t count, int threads, PackedHost &out);
uint32_t detect_uniform_len(const uint64_t *offsets, uint64_t n_reads);

// Output file, plain or gzip by extension.  Gzip output is a single member (what the reference's flate2 GzEncoder writes and
// its GzDecoder reads) whose deflate blocks are compressed in 1 MiB slices on several threads (deflate on one thread does
// ~50 MB/s of text).
class OutFile {
public:
    ~OutFile();
    bool open(const char *path, int threads = 0);          // nullptr: standard output (never compressed); 0: a quarter of the host threads, at most 32
    bool write(const char *p, size_t n);
    bool close();
private:
    bool flush_members();
    FILE *f_ = nullptr; bool gz_ = false, own_ = true, wrote_ = false; int threads_ = 1;
    uint32_t crc_ = 0; uint64_t total_ = 0;                // of the whole member
    std::vector<char, DefaultInitAlloc<char>> pend_;
    std::vector<unsigned char, DefaultInitAlloc<unsigned char>> obuf_;
};

// survivors in input order: header / seq / "+" / qual (filter_bin main.rs:261-268)
bool write_survivors(const char *path, const FqRec *recs, uint64_t n, const uint8_t *keep, std::string &err);

} // namespace mf
