// Streaming gzip decoder for the FASTQ readers (host side, no HIP in here).
//
// Why not zlib's gzread: a single-member .gz can only be inflated serially, so with everything
// else in the file pipeline parallel the inflate loop alone sets the rate of `.fq.gz` input
// (zlib 1.2.11: ~350 MB/s of text per file).  This decoder keeps the whole compressed file mapped,
// decodes with a 64-bit bit buffer that is refilled once per symbol group, 11-bit / 8-bit
// first-level Huffman tables, literal runs and word-wise match copies (the usual structure of a
// fast inflate), and hands out the text in caller-sized pieces.  CRC-32 and ISIZE of every member are
// checked like gzread does (the CRC on a helper thread, one slice behind the decoder).
// Semantics match gzread for what the readers need: concatenated members are decoded back to back,
// bytes after the last member that are not a gzip header are ignored, a damaged stream is an error.
#pragma once
#include <condition_variable>
#include <deque>
#include <mutex>
#include <stddef.h>
#include <stdint.h>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace mf {

class GzInflater {
public:
    GzInflater() = default;
    ~GzInflater();
    GzInflater(const GzInflater &) = delete;
    GzInflater &operator=(const GzInflater &) = delete;
    // `data` must stay valid and unchanged while the object is in use
    void open(const uint8_t *data, size_t size);
    // Decode up to `cap` bytes to `out`.  Returns the number of bytes written; 0 with eof() true at
    // the end of the file; -1 on a damaged stream (message in err).
    long read(uint8_t *out, size_t cap, std::string &err);
    bool eof() const { return state_ == DONE; }

private:
    long read_slice(uint8_t *out, size_t cap, std::string &err);
    // the CRC runs on a helper thread, one slice behind the decoder; crc_wait() returns once it has caught up
    void crc_push(const uint8_t *p, size_t n);
    void crc_wait();
    void crc_loop();
    std::thread crc_thread_; std::mutex crc_mu_; std::condition_variable crc_cv_, crc_idle_cv_;
    std::deque<std::pair<const uint8_t *, size_t>> crc_jobs_; bool crc_busy_ = false, crc_stop_ = false;
    enum State { MEMBER_HEADER, BLOCK_HEADER, STORED, HUFFMAN, MEMBER_TRAILER, DONE };
    bool parse_member_header(std::string &err);
    bool parse_block_header(std::string &err);
    bool build_tables(const uint8_t *lens, unsigned n_litlen, unsigned n_dist, std::string &err);
    bool check_trailer(std::string &err);
    // bit reader (LSB first); refill keeps at least 56 valid bits while input lasts
    inline void refill();
    inline uint32_t peek(unsigned n) const { return (uint32_t)(bitbuf_ & ((1ULL << n) - 1)); }
    inline void drop(unsigned n) { bitbuf_ >>= n; bitcnt_ -= n; }
    bool need_bits(unsigned n);          // false when the input ends before n bits are there

    const uint8_t *in_ = nullptr, *in_end_ = nullptr, *in_begin_ = nullptr;
    uint64_t bitbuf_ = 0; unsigned bitcnt_ = 0;
    size_t overrun_ = 0;                 // zero bytes fed past the end of the input (only legal inside the last few bits)
    State state_ = DONE;
    bool last_block_ = false;
    size_t stored_left_ = 0;
    // a match that did not fit into the caller's buffer
    unsigned pend_len_ = 0, pend_dist_ = 0;
    // history: the last 32 KiB handed out, for matches that reach back behind the current call
    std::vector<uint8_t> hist_; size_t hist_len_ = 0;
    // member accounting
    uint32_t crc_ = 0; uint64_t member_out_ = 0;
    bool any_member_ = false;
    bool transparent_ = false;         // the input is not gzip at all: handed through unchanged, as gzread does
    // decode tables
    std::vector<uint32_t> lit_, dist_;
};

} // namespace mf
