// DEFLATE decoding on the device (mf_gzdev.hip): one gzip stream decoded by thousands of wavefronts.
//
// The scheme is the two-pass one of the host reader (mf_pinflate.h; pugz / rapidgzip in the literature), laid out for the GPU:
//   * the compressed file is cut into chunks of `chunk_bytes`; ONE WAVEFRONT per chunk.  Chunk 0 starts at the first deflate
//     block of the member; every other chunk SEARCHES its range for a dynamic-Huffman block header with complete codes --
//     64 bit offsets are tested per step, one per lane -- and decodes from there up to the first block boundary at or behind
//     the start of the next chunk's range, into 16-bit symbols: a byte, or MARK | index where a match reaches back into the
//     32 KiB the chunk cannot know;
//   * gz_chain_kernel (one workgroup) walks the chunks in order: a chunk is accepted only if it began exactly where the
//     accepted data ends -- by induction from chunk 0 every accepted chunk starts at a true block boundary -- gives it its
//     place in the text and resolves the last 32 KiB of its symbols, which are the window of the chunk behind it;
//   * gz_resolve_kernel turns every other symbol into a byte, in parallel, reading the windows the chain left in the text.
// Whatever does not link (a stored or fixed block at a seam, a false candidate, a member boundary) stops the chain with the
// exact bit position and window; the host decodes across the gap with its serial decoder and restarts the chain.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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

// ---- linking the chunks (one workgroup, in stream order) and turning symbols into text
enum GzStop : uint32_t {
    GZ_STOP_NONE = 0,       // every chunk of the range was looked at
    GZ_STOP_GAP = 1,        // chunk `next` does not begin where the accepted data ends (or has no data): the host decodes across the gap
    GZ_STOP_MEMBER_END = 2  // the chunk before `next` ended behind the final block of a member (cur_bit: first bit behind it)
};
struct GzChain {
    uint64_t cur_bit;       // block boundary where the accepted data ends
    uint64_t total;         // bytes of text accepted so far (offset of the next byte in the member's text)
    uint32_t next;          // first chunk not yet accepted or discarded
    uint32_t stop;          // GzStop
    uint32_t linked, discarded;
    uint32_t wlen, pad;     // valid bytes in `window` (right-aligned)
    uint8_t window[GZ_WINDOW];   // the last 32 KiB of accepted text
};
// Walks chunks [chain->next, chunk_hi) in order.  A chunk is accepted iff its status is GZ_AT_BOUNDARY / GZ_MEMBER_END and it starts
// at chain->cur_bit; chunks that start inside accepted data are discarded.  For an accepted chunk c: out_off[c] = its text offset,
// and the last min(n_sym, 32 KiB) of its symbols are written as bytes to text[out_off[c] + ...] (text is addressed by absolute
// offset minus text_base; the 32 KiB in front of the first accepted chunk are written too, from chain->window).  out_off of
// chunks that are not accepted is set to ~0.
hipError_t launch_gz_chain(GzChain *d_chain, const GzChunk *d_chunks, uint32_t chunk_lo, uint32_t chunk_hi, const uint16_t *d_sym,
                           uint64_t sym_cap, uint64_t *d_out_off, uint8_t *d_text, uint64_t text_base, hipStream_t st);
// Every symbol of the accepted chunks of [chunk_lo, chunk_hi) that the chain did not already write becomes a byte of text.
// max_sym: the largest n_sym among them (grid sizing).
hipError_t launch_gz_resolve(const GzChunk *d_chunks, uint32_t chunk_lo, uint32_t chunk_hi, const uint16_t *d_sym, uint64_t sym_cap,
                             const uint64_t *d_out_off, uint8_t *d_text, uint64_t text_base, uint32_t max_sym, hipStream_t st);
// CRC-32 (gzip) of d_text[0 .. n) as 64 KiB pieces: d_piece[i] = the pure polynomial remainder (register starts at 0, no final
// inversion) of piece i; gz_crc_finish() on the host folds them into zlib's crc32() value.
constexpr uint32_t GZ_CRC_PIECE = 65536;
hipError_t launch_gz_crc(const uint8_t *d_text, uint64_t n, uint32_t *d_piece, hipStream_t st);
uint32_t gz_crc_finish(const uint32_t *piece, uint64_t n);                 // crc32(0, text, n)
uint32_t gz_crc_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b);   // crc32 of A||B from crc32(A), crc32(B)

} // namespace mf
