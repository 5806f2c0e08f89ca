// Deterministic synthetic read generator (SURVEY.md 8d inputs) writing the
// packed layout directly.  Counter-based, so the output does not depend on the
// number of worker threads.  Host only.
#pragma once
#include "mf_host.h"

namespace mf {

struct SynthOut {
    std::vector<uint32_t> words;   // padded_words_for(n_words)
    uint64_t n_words = 0;
    std::vector<uint64_t> npos;    // ascending
    uint64_t n_mito = 0;           // reads sampled from the bait
};

// Background: stream word pair (2i, 2i+1) = the two halves of
// mix64(seed ^ (i+1)*0x9E3779B97F4A7C15).  Read r is a bait read iff
// mix64(seed ^ 0xA5A5A5A5 ^ r*C) % 1e6 < mito_ppm; it is then copied from a
// random position / strand of a bait record with length >= read_len, with
// per-base substitution probability sub_ppm/1e6.  Read r carries invalid
// bases iff a third hash % 1e6 < n_read_ppm, each base then invalid with
// probability n_base_ppm/1e6.
// "Real-data-shaped" extras (bench.py extra.realistic): msat_ppm of the reads are microsatellites (a random motif of 1..6 bases
// repeated over the whole read: poly-A/T, (TA)n, ...), numt_ppm are NUMT-like (sampled from the bait like a bait read, but with
// numt_div_ppm / 1e6 substitutions per base instead of sub_ppm).
struct SynthExtra { uint32_t msat_ppm = 0, numt_ppm = 0, numt_div_ppm = 150000; };
bool synth_reads(uint64_t n_reads, uint32_t read_len, uint64_t seed, const BaitHost &bait,
                 uint32_t mito_ppm, uint32_t sub_ppm, uint32_t n_read_ppm, uint32_t n_base_ppm,
                 int threads, SynthOut &out, std::string &err, const SynthExtra &extra = SynthExtra());

} // namespace mf
