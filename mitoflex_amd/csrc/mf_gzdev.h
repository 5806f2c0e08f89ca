// DEFLATE decoding on the device (mf_gzdev.hip): one gzip stream decoded by thousands of wavefronts.
//
// The scheme is the two-pass one of the host reader (mf_pinflate.h; pugz / rapidgzip in the literature), laid out for the GPU:
//   * the compressed file is cut into chunks of `chunk_bytes`; ONE WAVEFRONT per chunk.  Chunk 0 starts at the first deflate
//     block of the member; every other chunk SEARCHES its range for a dynamic-Huffman block header with complete codes --
//     64 bit offsets are tested per step, one per lane -- and decodes from there up to the first block boundary at or behind
//     the start of the next chunk's range, into 16-bit symbols: a byte, or MARK | index where a match reaches back into the
//     32 KiB the chunk cannot know;
//   * the chunks are LINKED in order: a chunk is accepted only if it began exactly where the accepted data ends -- by induction
//     from chunk 0 every accepted chunk starts at a true block boundary.  That walk reads descriptors only and runs on the host
//     (gz_link_walk); the windows -- a chunk's markers point into the resolved last 32 KiB of the chunk before it -- are a scan
//     over the chunks, done on the device in two levels (launch_gz_link), nothing of it one chunk at a time;
//   * gz_resolve_kernel turns every other symbol into a byte, in parallel, reading the windows the link left in the text.
// Whatever does not link (a stored or fixed block at a seam, a false candidate, a member boundary) stops the walk with the
// exact bit position; the host decodes across the gap with its serial decoder (and the window) and the walk goes on.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace mf {

constexpr uint16_t GZ_MARK = 0x8000;            // symbol = GZ_MARK | index into the 32 KiB in front of the chunk
constexpr uint32_t GZ_WINDOW = 32768;

enum GzStatus : uint32_t {
    GZ_NONE = 0,            // no block header found in the chunk's range / not decoded
    GZ_AT_BOUNDARY = 1,     // stopped in front of a block that starts at or behind the stop position
    GZ_MEMBER_END = 2,      // stopped behind the final block of the member
    GZ_FAILED = 3,          // the data does not decode (from a speculative start: a false candidate)
    GZ_OVERFLOW = 4         // symbol buffer full: stopped at the last block boundary that fitted (end_bit, n_sym are valid)
};

struct GzChunk {
    uint64_t start_bit;     // first bit of the block header decoding began at
    uint64_t end_bit;       // block boundary where it stopped
    uint32_t n_sym;         // symbols written
    uint32_t status;        // GzStatus
};

// Decode chunks [chunk_lo, chunk_lo + n_chunks) of the deflate data in d_data[0 .. size) (readable, zero padded, up to size + 64).
// ring_bytes != 0: d_data is a ring of that many bytes (a power of two, >= 4096) and byte b of the file lives at d_data[b % ring_bytes];
// limit_bytes: the bytes of the file in front of it are on the device -- a chunk whose decoding would read past it stops as
// GZ_FAILED (n_sym = 8), nothing of it is accepted and the host bridges the stretch.
// Chunk c covers the bits [(base_byte + c * chunk_bytes) * 8, (base_byte + (c + 1) * chunk_bytes) * 8).  exact_chunk /
// exact_bit: that chunk starts at exactly that bit, a known block boundary (the first block of a member, or where a gap
// fill ended); pass exact_chunk = ~0u for none.  Symbols of chunk c go to d_sym + (c - chunk_lo) * sym_cap.
// d_scratch: gz_decode_scratch_bytes(n_chunks) bytes for the lane-parallel kernel's code lists (nullptr, or MF_GZDEV_KERNEL=serial:
// the one-lane walk of round 3 runs instead).
hipError_t launch_gz_decode(const uint8_t *d_data, uint64_t ring_bytes, uint64_t size, uint64_t limit_bytes, uint64_t base_byte, uint64_t chunk_bytes,
                            uint32_t chunk_lo, uint32_t n_chunks, uint32_t exact_chunk, uint64_t exact_bit, uint16_t *d_sym, uint64_t sym_cap,
                            GzChunk *d_chunks, uint32_t *d_scratch, hipStream_t st);
size_t gz_decode_scratch_bytes(uint32_t n_chunks);
bool gz_decode_serial();

// ---- linking the chunks and turning symbols into text
enum GzStop : uint32_t {
    GZ_STOP_NONE = 0,       // every chunk of the range was looked at
    GZ_STOP_GAP = 1,        // chunk `next` does not begin where the accepted data ends (or has no data): the host decodes across the gap
    GZ_STOP_MEMBER_END = 2  // the chunk before `next` ended behind the final block of a member (cur_bit: first bit behind it)
};
// Where the accepted data of a stream ends.  The state lives on the HOST (it is a few scalars; the walk over a slab's descriptors that
// moves it is gz_link_walk below); the one thing that lives on the device is the window -- the last 32 KiB of accepted text, which the
// link kernels read and leave behind (d_window of launch_gz_link) and which comes down only when the host has to decode across a gap
// or the next slab is linked on another device.
struct GzLinkState {
    uint64_t cur_bit = 0;   // block boundary where the accepted data ends
    uint64_t total = 0;     // bytes of text accepted so far (offset of the next byte in the stream's text)
    uint32_t next = 0;      // first chunk not yet accepted or discarded
    uint32_t stop = 0;      // GzStop
    uint32_t linked = 0, discarded = 0;
    uint32_t wlen = 0;      // valid bytes of the window (right-aligned)
};
// Walks the descriptors of chunks [st.next, chunk_hi) in order (h_chunks: indexed by chunk number).  A chunk is accepted iff its status is
// GZ_AT_BOUNDARY / GZ_MEMBER_END and it starts at st.cur_bit -- by induction from the member's first block every accepted chunk starts at
// a true block boundary --; chunks that start inside accepted data, or hold nothing, are discarded.  acc / acc_off: the accepted chunks
// and the text offsets of their first bytes.  Stops (st.stop) at a gap, behind a member's end, or at chunk_hi.
inline void gz_link_walk(const GzChunk *h_chunks, uint32_t chunk_hi, GzLinkState &st, std::vector<uint32_t> &acc, std::vector<uint64_t> &acc_off)
{
    acc.clear(); acc_off.clear();
    st.stop = GZ_STOP_NONE;
    uint32_t c = st.next;
    for (; c < chunk_hi; c++) {
        const GzChunk &ch = h_chunks[c];
        const bool ok = ch.status == GZ_AT_BOUNDARY || ch.status == GZ_MEMBER_END;
        if (!ok || ch.start_bit < st.cur_bit) { st.discarded++; continue; }
        if (ch.start_bit > st.cur_bit) { st.stop = GZ_STOP_GAP; break; }
        acc.push_back(c); acc_off.push_back(st.total);
        st.total += ch.n_sym; st.cur_bit = ch.end_bit; st.linked++;
        st.wlen = st.wlen + ch.n_sym < GZ_WINDOW ? st.wlen + ch.n_sym : GZ_WINDOW;
        if (ch.status == GZ_MEMBER_END) { c++; st.stop = GZ_STOP_MEMBER_END; break; }
    }
    st.next = c;
}
// The accepted chunks d_acc[0 .. n_acc) (chunk numbers, ascending; d_acc_off: their text offsets) become text, in two steps: text is addressed
// by absolute offset minus text_base.
// launch_gz_link -- the part that depends on the chunks in front: every chunk's TAIL (its last 32 Ki symbols) becomes text.  d_window: 32 KiB,
// the text in front of the first chunk (wlen_before valid bytes, right-aligned; they are written to the text in front of first_off = the first
// chunk's offset too) -> the last 32 KiB behind the last chunk.  The symbols' tails are rewritten in place.  d_scratch:
// gz_link_scratch_bytes(most chunks ever passed at once) bytes, used in stream order.  See mf_gzdev.hip for the scheme (a two-level scan
// over the windows: nothing in it is serial per chunk).  This is what one link step waits for the one before with.
// launch_gz_resolve -- the rest, every chunk's BODY: markers looked up in the 32 KiB of text in front of the chunk, which the link step of
// the same chunks has written.  Any stream that runs behind that link step; nothing later waits for it but the text's readers.
// max_sym: the largest n_sym among the chunks (grid sizing).
hipError_t launch_gz_link(const uint32_t *d_acc, const uint64_t *d_acc_off, uint32_t n_acc, uint32_t max_sym, const GzChunk *d_chunks, uint32_t chunk_lo, uint16_t *d_sym,
                          uint64_t sym_cap, uint8_t *d_window, uint32_t wlen_before, uint8_t *d_scratch, uint8_t *d_text, uint64_t text_base, uint64_t first_off, hipStream_t st);
hipError_t launch_gz_resolve(const uint32_t *d_acc, const uint64_t *d_acc_off, uint32_t n_acc, uint32_t max_sym, const GzChunk *d_chunks, uint32_t chunk_lo, const uint16_t *d_sym,
                             uint64_t sym_cap, uint8_t *d_text, uint64_t text_base, hipStream_t st);
inline uint32_t gz_link_group(uint32_t n_acc) { return n_acc <= 16 ? 4 : n_acc <= 64 ? 8 : n_acc <= 512 ? 16 : 32; }          // chunks per group of the scan's first level
inline size_t gz_link_scratch_bytes(uint32_t max_chunks)
{
    uint32_t groups = 1;                                           // the most groups any number of accepted chunks up to max_chunks makes
    for (uint32_t n : {16u, 64u, 512u, max_chunks}) { const uint32_t m = n < max_chunks ? n : max_chunks; if (m) { const uint32_t g = (m + gz_link_group(m) - 1) / gz_link_group(m); if (g > groups) groups = g; } }
    return (size_t)groups * GZ_WINDOW * 3 + 256;
}
void gz_preload();          // loads this file's code object now instead of at its first launch (a cold call's prefetch thread)

// CRC-32 (gzip) of d_text[0 .. n) as 64 KiB pieces: d_piece[i] = the pure polynomial remainder (register starts at 0, no final
// inversion) of piece i; gz_crc_finish() on the host folds them into zlib's crc32() value.
constexpr uint32_t GZ_CRC_PIECE = 65536;
hipError_t launch_gz_crc(const uint8_t *d_text, uint64_t n, uint32_t *d_piece, hipStream_t st);
uint32_t gz_crc_finish(const uint32_t *piece, uint64_t n);                 // crc32(0, text, n)
uint32_t gz_crc_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b);   // crc32 of A||B from crc32(A), crc32(B)

} // namespace mf
