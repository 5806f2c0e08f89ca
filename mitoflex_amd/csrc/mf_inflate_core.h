// Huffman decode tables shared by the streaming decoder (mf_inflate.cpp) and the parallel one (mf_pinflate.cpp).
// Internal to the library.
#pragma once
#include <stdint.h>
#include <string.h>
#include <vector>

namespace mf {
namespace inflate_core {

constexpr unsigned LIT_BITS = 11, DIST_BITS = 8, PRE_BITS = 7;
// table entry: bits 0-4 codeword bits to drop at this level, 5-7 kind, 8-12 extra bits (or sub-table bits), 16-31 value
enum Kind : uint32_t { LITERAL = 0, LENGTH = 1, END_OF_BLOCK = 2, LINK = 3, INVALID = 4, DISTANCE = 5 };
// bit 15 repeats "kind == LITERAL && this is a litlen table": the hot loop tests it first.
// bit 14: the entry carries TWO literals (both codes fit into the first-level index): second byte in bits 24-31,
// total length in bits 0-4, length of the first code alone in the extra field (for the careful loop).
constexpr uint32_t LITERAL_FLAG = 1u << 15, DOUBLE_FLAG = 1u << 14;
inline uint32_t entry(unsigned len, Kind kind, unsigned extra, unsigned value) { return len | ((uint32_t)kind << 5) | (extra << 8) | (value << 16); }
inline unsigned e_len(uint32_t e) { return e & 31u; }
inline unsigned e_kind(uint32_t e) { return (e >> 5) & 7u; }
inline unsigned e_extra(uint32_t e) { return (e >> 8) & 31u; }
inline unsigned e_value(uint32_t e) { return e >> 16; }

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline unsigned bit_reverse(unsigned code, unsigned len)
{
    unsigned r = 0;
    for (unsigned i = 0; i < len; i++) { r = (r << 1) | (code & 1u); code >>= 1; }
    return r;
}

// Canonical Huffman decode table with one level of sub-tables.  make(sym) gives the entry of a symbol
// without its length field.  false: over-subscribed code.
template <class Make>
bool build_table(const uint8_t *lens, unsigned n, unsigned main_bits, std::vector<uint32_t> &tab, Make make)
{
    unsigned count[16] = {0};
    for (unsigned s = 0; s < n; s++) count[lens[s]]++;
    count[0] = 0;
    unsigned next[16]; unsigned code = 0; uint64_t kraft = 0;
    for (unsigned l = 1; l <= 15; l++) { code = (code + count[l - 1]) << 1; next[l] = code; kraft += (uint64_t)count[l] << (15 - l); }
    if (kraft > (1u << 15)) return false;
    const unsigned main_size = 1u << main_bits;
    tab.assign(main_size, entry(0, INVALID, 0, 0));
    // first pass: longest code behind every main-table prefix
    std::vector<uint8_t> sub_bits;
    unsigned codes[320];
    bool any_long = false;
    for (unsigned s = 0; s < n; s++) {
        const unsigned l = lens[s];
        if (!l) continue;
        codes[s] = bit_reverse(next[l]++, l);
        if (l > main_bits) any_long = true;
    }
    if (any_long) {
        sub_bits.assign(main_size, 0);
        for (unsigned s = 0; s < n; s++) {
            const unsigned l = lens[s];
            if (l > main_bits) { uint8_t &b = sub_bits[codes[s] & (main_size - 1)]; if (l - main_bits > b) b = (uint8_t)(l - main_bits); }
        }
        for (unsigned p = 0; p < main_size; p++)
            if (sub_bits[p]) {
                const unsigned start = (unsigned)tab.size();
                tab.resize(start + (1u << sub_bits[p]), entry(0, INVALID, 0, 0));
                tab[p] = entry(main_bits, LINK, sub_bits[p], start);
            }
    }
    for (unsigned s = 0; s < n; s++) {
        const unsigned l = lens[s];
        if (!l) continue;
        const uint32_t base = make(s);
        if (l <= main_bits) {
            for (unsigned i = codes[s]; i < main_size; i += 1u << l) tab[i] = base | l;
        } else {
            const uint32_t link = tab[codes[s] & (main_size - 1)];
            const unsigned start = e_value(link), sb = e_extra(link);
            for (unsigned i = codes[s] >> main_bits; i < (1u << sb); i += 1u << (l - main_bits)) tab[start + i] = base | (l - main_bits);
        }
    }
    return true;
}

inline uint32_t lit_entry(unsigned s)
{
    if (s < 256) return entry(0, LITERAL, 0, s) | LITERAL_FLAG;
    if (s == 256) return entry(0, END_OF_BLOCK, 0, 0);
    if (s < 286) return entry(0, LENGTH, LEN_EXTRA[s - 257], LEN_BASE[s - 257]);
    return entry(0, INVALID, 0, 0);
}
inline uint32_t dist_entry(unsigned s) { return s < 30 ? entry(0, DISTANCE, DIST_EXTRA[s], DIST_BASE[s]) : entry(0, INVALID, 0, 0); }


// Literal-heavy text (FASTQ bases and qualities) decodes one symbol per dependent table load; where the index
// bits left over after a literal hold a second complete literal code, one load yields both.
inline void pair_literals(std::vector<uint32_t> &lit)
{
    std::vector<uint32_t> single(lit.begin(), lit.begin() + (1u << LIT_BITS));
    for (unsigned i = 0; i < (1u << LIT_BITS); i++) {
        const uint32_t e1 = single[i];
        if (!(e1 & LITERAL_FLAG)) continue;
        const unsigned l1 = e_len(e1);
        if (l1 >= LIT_BITS) continue;
        const uint32_t e2 = single[i >> l1];                  // index bits above l1, zero-extended: valid iff the code is short enough
        if (!(e2 & LITERAL_FLAG)) continue;
        const unsigned l2 = e_len(e2);
        if (l1 + l2 > LIT_BITS) continue;
        lit[i] = (l1 + l2) | ((uint32_t)LITERAL << 5) | (l1 << 8) | LITERAL_FLAG | DOUBLE_FLAG | ((e1 >> 16 & 0xFFu) << 16) | ((e2 >> 16 & 0xFFu) << 24);
    }
}

// code lengths of the fixed Huffman block type: 288 literal/length + 32 distance
inline void fixed_lengths(uint8_t lens[320])
{
    for (unsigned i = 0; i < 144; i++) lens[i] = 8;
    for (unsigned i = 144; i < 256; i++) lens[i] = 9;
    for (unsigned i = 256; i < 280; i++) lens[i] = 7;
    for (unsigned i = 280; i < 288; i++) lens[i] = 8;
    for (unsigned i = 0; i < 32; i++) lens[288 + i] = 5;
}

static const uint8_t PRECODE_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

} // namespace inflate_core
} // namespace mf
