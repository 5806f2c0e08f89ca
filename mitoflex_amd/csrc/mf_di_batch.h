// Device ingest path, part 6 of 8: what the consumers work with -- writers, batches, per-consumer device scratch, the quality filter's state, a mate.
#pragma once
#include "mf_di_gznext.h"

namespace mf {
namespace {

// ---- survivors on their way to the output file (one writer thread per mate; pieces arrive in order)
class Writer {
public:
    bool open(const char *path) { ok_ = of_.open(path); if (ok_) th_ = std::thread([this] { run(); }); return ok_; }
    void push(std::vector<char> &&b) { { std::lock_guard<std::mutex> lk(mu_); q_.push_back(std::move(b)); } cv_.notify_one(); }
    bool close()
    {
        if (th_.joinable()) { { std::lock_guard<std::mutex> lk(mu_); fin_ = true; } cv_.notify_one(); th_.join(); }
        return of_.close() && ok_;
    }
    ~Writer() { if (th_.joinable()) { { std::lock_guard<std::mutex> lk(mu_); fin_ = true; } cv_.notify_one(); th_.join(); } }
private:
    void run()
    {
        for (;;) {
            std::vector<char> b;
            { std::unique_lock<std::mutex> lk(mu_); cv_.wait(lk, [&] { return fin_ || !q_.empty(); }); if (q_.empty()) return; b = std::move(q_.front()); q_.pop_front(); }
            if (ok_ && !b.empty() && !of_.write(b.data(), b.size())) ok_ = false;
        }
    }
    OutFile of_; bool ok_ = false, fin_ = false;
    std::thread th_; std::mutex mu_; std::condition_variable cv_; std::deque<std::vector<char>> q_;
};

// bits [r0, r0 + n) of a bitmap -> out (bit 0 = bit r0); out has (n + 31) / 32 words
void extract_bits(const std::vector<uint32_t> &v, uint64_t r0, uint64_t n, uint32_t *out)
{
    const uint64_t nw = (n + 31) / 32, w0 = r0 >> 5; const uint32_t sh = (uint32_t)(r0 & 31);
    for (uint64_t j = 0; j < nw; j++) {
        const uint32_t a = w0 + j < v.size() ? v[w0 + j] : 0, b = w0 + j + 1 < v.size() ? v[w0 + j + 1] : 0;
        out[j] = sh ? (a >> sh) | (b << (32 - sh)) : a;
    }
    if (n & 31) out[nw - 1] &= (1u << (n & 31)) - 1;
}
// the other way: n bits of src (bit 0 first) become bits [r0, r0 + n) of v
void append_bits(std::vector<uint32_t> &v, uint64_t r0, uint64_t n, const uint32_t *src)
{
    if (!n) return;
    { const size_t need = (size_t)((r0 + n + 31) / 32 + 1); if (v.size() < need) v.resize(need, 0); }      // (pieces finish out of order: never shrink)
    const uint64_t nw = (n + 31) / 32, w0 = r0 >> 5; const uint32_t sh = (uint32_t)(r0 & 31);
    for (uint64_t j = 0; j < nw; j++) {
        uint32_t x = src[j];
        if (j == nw - 1 && (n & 31)) x &= (1u << (n & 31)) - 1;
        v[w0 + j] |= x << sh;
        if (sh) v[w0 + j + 1] |= x >> (32 - sh);
    }
}

// records of one piece of text, cut where they lie; waits (with its text) until the other mate's pass bits cover it
struct Batch {
    std::unique_ptr<TextBuf> buf;
    const uint8_t *text = nullptr;       // the first record's header (the piece's text less the carry in front of it)
    DevBuf<uint64_t> line_start;         // offsets from `text`
    uint64_t n_rec = 0, rec_base = 0, n_text = 0, n_lines = 0;
    int ldev = 0;
    bool filtered = false;               // its pass bits are in the mate's bitmap
    // the quality filter's job (QualState below): what one pass over the records found, kept with the batch until its turn to be decided
    DevBuf<uint32_t> q_bad, q_sl, q_ql, q_olen; DevBuf<uint8_t> q_fl; DevBuf<uint64_t> q_hash;
    uint64_t q_done = 0;                                  // (Ingest::mu) records of it that have been decided
};
// records [r0, r0 + n) of a batch, decided: their text goes to bytes [out_at, out_at + bytes) of the mate's output file
struct QPart { uint64_t r0 = 0, n = 0, out_at = 0, bytes = 0; };

// what a consumer thread keeps per device: scratch buffers and the refillable read set (its own context of the device: own streams)
struct DevScratch {
    int ldev = 0, dev = 0, lane = 0; DevCtx *ctx = nullptr;
    mf_reads *reads = nullptr;
    DevBuf<uint32_t> tile_cnt, seq_len, inv_cnt, out_len, minmax, mask; DevBuf<uint64_t> tile_base, scan_tmp, inv_base, out_off, offsets_tmp; DevBuf<uint8_t> d_out;
    // small results the host waits for (counts that size the next buffers), in pinned memory: a copy to pageable memory is a
    // synchronisation of its own.  [0] newlines [1] last byte [2] used [3] bases [4] min/max length [5] invalid bases [6] output bytes;
    // [7], [8]: values on their way TO the device (the virtual end of an unterminated last line, the start value of min/max)
    uint64_t *h_small = nullptr;
    uint32_t *h_bits = nullptr; size_t h_bits_cap = 0;       // pinned: the pass bits of a piece on their way to the host, the keep mask on its way back
    uint8_t *h_out = nullptr; size_t h_out_cap = 0;          // pinned: survivors on their way to the writer
    // the quality filter's job: the other mate's scan results and keep flags on their way up, keep flags on their way down (pinned), per-record scratch
    uint8_t *h_stage = nullptr; size_t h_stage_cap = 0;
    DevBuf<uint32_t> q_bad2; DevBuf<uint8_t> q_fl2, q_alive, q_dup, q_keep;
    hipError_t stage(size_t bytes)
    {
        if (bytes <= h_stage_cap) return hipSuccess;
        if (h_stage) (void)hipHostFree(h_stage);
        h_stage = nullptr; h_stage_cap = 0;
        hipError_t e = hipHostMalloc((void **)&h_stage, bytes + bytes / 2 + 65536, hipHostMallocDefault);
        if (e == hipSuccess) h_stage_cap = bytes + bytes / 2 + 65536;
        return e;
    }
    ~DevScratch()
    {
        reads_release(reads); (void)hipSetDevice(dev);
        if (h_small) (void)hipHostFree(h_small); if (h_bits) (void)hipHostFree(h_bits); if (h_out) (void)hipHostFree(h_out); if (h_stage) (void)hipHostFree(h_stage);
    }
};

// A consumer's scratch -- small pinned buffers, device buffers, the refillable read set with everything the filter hangs on it -- is kept
// from call to call per (device, consumer): making it anew costs a call a few milliseconds at the start and a hipFree per buffer of the
// read set (each waits for the device to go idle) at the end.  MF_KEEP_BUFFERS=0 releases it with the call.
class ScratchCache {
public:
    std::unique_ptr<DevScratch> take(int ldev, int lane)
    {
        std::lock_guard<std::mutex> lk(mu_);
        auto it = kept_.find(std::make_pair(ldev, lane));
        if (it == kept_.end()) return nullptr;
        std::unique_ptr<DevScratch> p = std::move(it->second);
        kept_.erase(it);
        return p;
    }
    void give(std::unique_ptr<DevScratch> p)
    {
        if (!p || g_knobs.starts_0(KN_KEEP_BUFFERS)) return;
        std::lock_guard<std::mutex> lk(mu_);
        kept_[std::make_pair(p->ldev, p->lane)] = std::move(p);
    }
    void clear() { std::map<std::pair<int, int>, std::unique_ptr<DevScratch>> gone; { std::lock_guard<std::mutex> lk(mu_); gone.swap(kept_); } }      // (their device buffers go back to the pool: clear the pool after this)
private:
    std::mutex mu_; std::map<std::pair<int, int>, std::unique_ptr<DevScratch>> kept_;
};
ScratchCache &g_scratch = *new ScratchCache();          // (never destroyed: its entries would call into HIP while the process is being torn down)

// ---- the quality filter's job on this path (the reference's filter_v2: filter/filter_bin/src/main.rs:188-323)
struct QualState {
    QualParams P; bool pe = false; uint64_t cap = ~0ull;          // cap: the longest a cut string gets (end - start), ~0: no end
    SegArray<uint32_t> bad2; SegArray<uint8_t> fl2, keep;         // mate 2's scan results, mate 1's decisions: per record of the file, on the host
    uint64_t panic_rec[2] = {~0ull, ~0ull};                       // (Ingest::mu) the first record at which the reference would panic, per mate, among the scanned pieces
    // decisions are taken a piece of mate 1 at a time, in order (Ingest::emit_mu):
    uint64_t budget = 0, kept = 0, out_pos[2] = {0, 0};
    uint64_t decided = 0; bool decided_final = false, panicked = false;      // (Ingest::mu) records [0, decided) have their keep flags; final: no more will be decided
    int in_flight = 0;                                            // (Ingest::mu) pieces being gathered and written
    // the de-duplication set (keys, smallest file index per key; mf_kernels.hip): on the device for the whole file
    DevBuf<unsigned long long> dd_keys, dd_first, dd_small; uint64_t dd_slots = 0, dd_n = 0;
    OutChunks chunks; QSink sink[2];                              // (the pool first: the sinks' threads give their last chunks back to it)
    double t_scan = 0, t_decide = 0, t_gather = 0, t_chunk = 0;   // (Ingest::mu) summed over the consumers; t_chunk: waiting for a free chunk = for the writers
};

struct Mate {
    std::string path; Mapped map; bool gz = false;
    std::unique_ptr<GzStream> gzs; Slots slots;
    // producer: text pieces in order
    std::thread prod; std::mutex mu; std::condition_variable cv; std::deque<TextPiece> ready; int prod_rc = MF_OK; std::string prod_err; bool prod_done = false;
    std::atomic<bool> stop{false};
    // consumers (under Ingest::mu): pieces are taken in order; their line index is cut in that order too (the carry links them),
    // packing and filtering of several pieces run side by side
    uint64_t taken = 0, a_turn = 0;      // pieces handed to a consumer; the piece whose line index may be cut now
    bool eof = false;                    // the last piece has been taken
    uint64_t rec_indexed = 0;            // records of the pieces indexed so far (the next piece's first record)
    uint64_t rec_filtered = 0;           // ... of the leading pieces whose pass bits are in `bits`
    uint8_t *h_carry = nullptr; size_t h_carry_cap = 0, carry = 0;      // pinned: the head of the record the last piece left unfinished
    std::deque<std::shared_ptr<Batch>> batches;          // indexed, in order; leave when written
    std::vector<uint32_t> bits;                          // pass bits of the whole file so far, one per record
    Writer out;
    ~Mate()
    {
        TRACE("~Mate");
        stop = true; slots.wake();
        batches.clear(); ready.clear();                   // (text buffers give their slots back: a producer waiting for one wakes up)
        if (prod.joinable()) prod.join();
        ready.clear();
        gzs.reset();
        if (h_carry) (void)hipHostFree(h_carry);
    }
};

} // namespace
} // namespace mf
