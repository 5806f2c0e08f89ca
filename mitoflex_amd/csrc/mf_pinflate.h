// Parallel decoding of ONE gzip stream (host side).
//
// A deflate stream has no index, so a single-member .gz is normally inflated by one thread.  This
// reader cuts the compressed file into chunks and decodes them concurrently all the same:
//   * chunk 0 of a group starts at a known block boundary with a known 32 KiB window;
//   * every other chunk SEARCHES for the next dynamic-Huffman block header after its nominal start
//     (bit by bit; a candidate must carry complete precode, literal/length and distance codes) and
//     decodes from there into 16-bit symbols, writing a marker instead of a byte wherever a match
//     reaches back into the window it does not know;
//   * the chunks are then linked in order: a chunk is accepted only if the bit position where the
//     accepted data ends is exactly the position it started from -- by induction from chunk 0 every
//     accepted chunk starts at a true block boundary of the true stream.  Whatever does not link
//     (a false candidate, a stored or fixed block at the seam) is decoded serially across the gap;
//   * markers are replaced from the (now known) windows, chunk by chunk in parallel.
// BGZF files (bgzip / htslib: many small members that state their own size) need none of this: a run of
// members is decoded side by side.
// The result is the same byte stream a serial inflate produces; CRC-32 and ISIZE of every member are
// verified.  The idea follows the published two-pass schemes for gzip (pugz, rapidgzip).
#pragma once
#include "mf_host.h"            // DefaultInitAlloc
#include <stddef.h>
#include <stdint.h>
#include <string>
#include <thread>
#include <vector>

namespace mf {

uint32_t crc32_fast(uint32_t crc, const uint8_t *p, size_t n);                       // zlib's crc32() contract; PCLMULQDQ folding where the CPU has it
void resolve_symbols(const uint16_t *s, size_t n, const uint8_t *window, uint8_t *out);    // marker symbols -> bytes (see mf_pinflate.cpp)

// decode across a gap with the serial decoder (used by the device decoder for what does not link); window: 32 KiB, the last wlen valid
bool inflate_gap(const uint8_t *data, size_t size, uint64_t from_bit, uint64_t to_bit, const uint8_t *window, size_t wlen,
                 std::vector<uint8_t> &out, uint64_t &end_bit, bool &member_end, std::string &err);

// (test infrastructure: the stand-in for the device decode kernel in tests/native/ingest_stub.cpp)  One speculative chunk: data[0 .. size) are
// the bytes of the file from byte `origin` on, bit positions are absolute; exact = decode from from_bit itself, else search [from_bit, to_bit) for
// a dynamic block header with complete codes; symbols are bytes or 0x8000 | index into the 32 KiB in front of the chunk.
// Returns 0 nothing found, 1 stopped in front of the first block at or behind to_bit, 2 behind the member's final block, 3 does not decode.
int speculative_chunk(const uint8_t *data, size_t size, uint64_t origin, uint64_t from_bit, uint64_t to_bit, bool exact,
                      std::vector<uint16_t> &sym, uint64_t &start_bit, uint64_t &end_bit);

class ParallelGzReader {
    using ByteBuf = std::vector<uint8_t, DefaultInitAlloc<uint8_t>>;   // sized without being zero-filled
public:
    // data must stay mapped while the reader lives.  chunk_bytes: compressed bytes per speculative chunk.
    ParallelGzReader() = default;
    ~ParallelGzReader();
    ParallelGzReader(const ParallelGzReader &) = delete;
    ParallelGzReader &operator=(const ParallelGzReader &) = delete;
    void open(const uint8_t *data, size_t size, int threads, size_t chunk_bytes = (size_t)2 << 20);
    // same contract as GzInflater::read: bytes written, 0 at the end, -1 on a damaged stream
    long read(uint8_t *out, size_t cap, std::string &err);
    bool eof() const { return !pre_running_ && done_ && opos_ == obuf_.size(); }   // done_ belongs to the helper thread while it runs
    // bookkeeping for tests and tuning
    uint64_t chunks_linked = 0, chunks_discarded = 0, gap_fill_bytes = 0;

private:
    bool fill(ByteBuf &dst, std::string &err);   // decode the next group of chunks into dst
    bool fill_bgzf(ByteBuf &dst, std::string &err, bool &handled);   // ... or a run of BGZF members, side by side
    void start_prefetch();                             // ... on a helper thread, while the current group is being served
    std::thread pre_; bool pre_running_ = false, pre_ok_ = true; std::string pre_err_;
    ByteBuf nbuf_;
    struct Scratch; Scratch *scratch_ = nullptr;       // per-chunk symbol buffers, kept from group to group (no fresh pages per group)
    bool begin_member(std::string &err);               // gzip header at cur_bit_ (byte aligned) -> first block
    const uint8_t *data_ = nullptr; size_t size_ = 0;
    int threads_ = 1; size_t chunk_ = 0;
    size_t cur_bit_ = 0;                               // block boundary where the accepted data ends
    bool in_member_ = false, done_ = false, any_member_ = false, transparent_ = false;
    std::vector<uint8_t> window_; size_t wlen_ = 0;    // last 32 KiB of accepted output, right-aligned
    uint32_t crc_ = 0; uint64_t member_out_ = 0;
    ByteBuf obuf_; size_t opos_ = 0;
};

} // namespace mf
