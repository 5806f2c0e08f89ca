// Device ingest path of mf_filter_fastq_files (see mf_devingest.h).  Streaming: what a file holds in device memory at any time is
// bounded, whatever its size.
//
// Per mate file one PRODUCER thread turns the file into pieces of text in device buffers, in order:
//   * a .gz is decoded in slabs of a few hundred speculative chunks (mf_gzdev.h).  Its compressed bytes pass through a RING per
//     device (a power of two of bytes; byte b of the file lives at ring[b % R]) that an uploader thread fills a piece at a time and
//     that is recycled as slabs are linked; the decode kernels of the slabs ahead run on the device's decode streams, each followed
//     by the copy of its chunks' descriptors to the host; which chunks are accepted is decided on the host from those (gz_link_walk);
//     the windows, marker resolution and CRC are kernels on the mate's post stream, and the piece is handed over WITH AN EVENT --
//     the producer waits for a slab's descriptors and for a text buffer, never for a kernel of its own;
//     symbol room per chunk follows the expansion the file has shown so far, slabs in flight the call's memory budget;
//     every slab's text goes to a buffer of its own (the 32 KiB window and the carry in front of it);
//   * a plain FASTQ file IS the text: it is read straight into such buffers (three staging buffers, the device's copy stream).
//   With n devices the slabs are dealt to them round robin: the link step of slab k needs the state the link step of slab k - 1
//   left (a few scalars on the host, the last 32 KiB of text through pinned memory), nothing else crosses devices.
// The streams all of this runs on are made once per process and device by a maker thread, in the order a cold call needs them
// (StreamSets below): the reference calls this path a process at a time.
// A few CONSUMER threads take the pieces.  Per piece, on the device that holds it: cut the text into records where it lies
// (mf_ingest.h; one piece of a mate at a time, in order: what is behind the last complete record of a piece, the carry, goes to the
// front of the next piece's buffer), then -- several pieces side by side, each consumer on its own streams -- the piece's job:
//   * the bait filter (mf_filter_fastq_files): 2-bit pack into the consumer's refillable read set, ONE filter pass over the piece's
//     reads; the pass bits come back to the host (a bit per read), where the pair rule is applied across the two mates' file-wide
//     bitmaps -- the pieces of the two mates do not cover the same records, the mate that is behind in records is advanced --; a piece
//     whose records the other mate has covered gets its list of kept records, its survivors are gathered on the device and copied out
//     to a writer thread per output file;
//   * the quality filter (mf_qualfilter_files, the reference's filter_v2): one pass over the records' bytes (counts, flags, cut
//     lengths, SipHash), decisions a piece of mate 1 at a time in file order (the other mate's counts through per-record arrays on the
//     host, the de-duplication set on the device, the -t budget on the host), the kept records formatted on the device and sent down
//     through pinned chunks to a writer thread per output file -- nearly every record is kept, so this job writes as much as it reads.
// A piece's buffers go back to the pool when its part of the output is on its way.  A producer blocks when its mate holds
// MF_INGEST_TEXT_BUFS text buffers.
#include "mf_di_qual.h"          // (and through it parts 1-7: mf_di_base.h, _pool, _streams, _upload, _gzstream, _gznext, _batch, _ingest)

namespace mf {


// what the two jobs of this path share: is it an input for the path, set-up, the run, what the caller learns about it
static int run_ingest(Ingest &I, const char *fq1, const char *fq2, const char *out1, const char *out2, std::string &err, IngestStats *stats)
{
    // (declared first: runs after everything of this call is gone.  What a process keeps between calls: up to MF_DEVPOOL_GB per device, default 8
    // -- enough for a caller that filters file after file of up to a gigabyte or two never to ask the runtime twice; a call on a file of several
    // gigabytes holds up to 20 GB and gives the rest back: hipMalloc of gigabytes takes a millisecond on this runtime (profiles/r05/a_cold_calls_before.log),
    // and a process that sits on 24 GB between calls, as round 4's did, is a poor neighbour on a shared GPU.  mf_release_cached() gives back all of it.)
    struct EndOfCall {
        bool timing = false; double t0 = 0;
        ~EndOfCall()
        {
            g_pool.trim(g_knobs.starts_0(KN_KEEP_BUFFERS) ? 0 : (size_t)g_knobs.u64(KN_DEVPOOL_GB, 8) << 30);
            if (timing) fprintf(stderr, "[mf device ingest] streams, threads and buffers of the call put away in %.3f s\n", now_s() - t0);
        }
    } end_of_call;
    I.nm = fq2 ? 2 : 1;
    if (I.devices.empty() || I.devices.size() > 64) { err = "bad device list"; return MF_E_ARG; }
    I.timing = g_knobs.is_set(KN_PIPE_TIMING);
    I.carry_room = (size_t)g_knobs.u64(KN_INGEST_CARRY_ROOM, (size_t)1 << 20);
    const char *in_path[2] = {fq1, fq2}, *out_path[2] = {out1, out2};
    for (int i = 0; i < I.nm; i++) I.out_path_[i] = out_path[i] ? out_path[i] : "";
    // ---- is this an input for the device path?
    for (int i = 0; i < I.nm; i++) {
        Mate &M = I.m[i];
        if (!in_path[i]) return MF_DEVINGEST_DECLINED;          // (standard input)
        M.path = in_path[i]; M.gz = has_gz_ext(in_path[i]);
        bool regular = false;
        if (!M.map.open(in_path[i], regular)) { err = std::string("Cannot open file ") + in_path[i]; return MF_E_IO; }
        if (!regular || M.map.n == 0) return MF_DEVINGEST_DECLINED;
        if (M.gz) {
            const uint8_t *d = M.map.p;
            if (M.map.n < 18 || d[0] != 0x1f || d[1] != 0x8b) return MF_DEVINGEST_DECLINED;      // (gzread hands such a file through; so does the host reader)
            if ((d[3] & 4) && M.map.n >= 18 && d[12] == 'B' && d[13] == 'C') return MF_DEVINGEST_DECLINED;   // BGZF: the host reader decodes its members side by side
        }
    }
    cold_mark("device ingest: inputs mapped");
    for (int d : I.devices) ingest_prefetch(d);          // (the streams' maker and the staging buffers: started now if nobody has yet)
    for (int d : I.devices) { DevCtx *c = nullptr; const int rc = get_ctx(d, &c, 0); if (rc) { err = mf_thread_error(); return rc; } }
    cold_mark("device ingest: device contexts ready");
    const double t_begin = now_s();
    I.t_begin = t_begin;
    g_pool.reset_peak();
    if (g_knobs.is_set(KN_DEVINGEST_TRACE)) { size_t f = 0, t = 0; (void)hipMemGetInfo(&f, &t); TRACE("device memory in use as the call starts %.3f GB, of which idle buffers of earlier calls %.3f GB", (double)(t - f) / 1e9, (double)g_pool.held(phys(I.devices[0])) / 1e9); }
    // text buffers a mate may hold: the consumers each hold one, the decoder one, the rest wait for the other mate or for a consumer (the quality
    // filter's pieces wait longer: their text is written out)
    // (a call that plans for less than 8 GB keeps two fewer in flight: a text buffer is a slab's text, 4.5 times its compressed bytes)
    uint64_t in_bytes = 0;
    for (int i = 0; i < I.nm; i++) in_bytes += I.m[i].map.n;
    const bool small_call = 8 * in_bytes < ((uint64_t)8 << 30) && !g_knobs.is_set(KN_INGEST_BUDGET_GB);
    const int text_bufs = (int)std::max<uint64_t>(2, g_knobs.u64(KN_INGEST_TEXT_BUFS, (I.qual ? 8 : 6) - (small_call ? 2 : 0))) + (int)I.devices.size() - 1;
    int rc = MF_OK;
    // an input that keeps the chip full of decode wavefronts for a long time gets the CU-masked set of streams (16 ms apiece to make and a
    // quarter of a second of the process's exit: a small file must not pay for them)
    uint64_t gz_bytes = 0;
    for (int i = 0; i < I.nm; i++) if (I.m[i].gz) gz_bytes += I.m[i].map.n;
    // Which set of streams: the CU-masked set is faster the moment decode kernels fill the chip for more than a few slabs -- a paired 2 x 1.2 GB input
    // 0.149 s against 0.263 s on plain streams, configs[4] 0.225 against 0.368 (calls of a warm process, profiles/r05/e_masks_ab.txt) -- and costs a
    // quarter of a second of the process's exit.  A library user's process lives on: masked from 256 MB of compressed input.  A process that
    // makes one call and ends (the CLIs say so: mf_set_option("short_lived", "1")) pays the exit with every call: masked only where the
    // difference is larger than that, from 8 GB.  MF_GZDEV_LARGE_MB overrides either.
    const bool large = gz_bytes >= (g_knobs.u64(KN_GZDEV_LARGE_MB, g_short_lived.load() ? 8192 : 256) << 20);
    // Device memory follows the input: everything in use on the device stays within 8 bytes per compressed byte of the call, at least 3 GB,
    // at most 24 (MF_INGEST_BUDGET_GB sets it).  Of that, 1.2 GB are not this path's (the runtime's own 0.83 GB as a process starts, the
    // streams' queues, the bait tables); and for every byte the mates plan for their rings, symbol rooms, code lists and text buffers
    // (GzStream::open) the call holds 0.5-0.7 more -- the consumers' read sets and line indexes, which grow with the text pieces, and
    // buffers of one size idle in the pool while another size is asked for (profiles/r05/g_mem_probe.txt: planned 1.61 GB -> 2.35 GB of
    // buffers, 3.49-3.70 GB in use; planned 5.8 -> 7.7, 9.4-10.0 in use).  The mates share what is left.
    const uint64_t budget = g_knobs.is_set(KN_INGEST_BUDGET_GB) ? g_knobs.u64(KN_INGEST_BUDGET_GB, 24) << 30
                                                         : std::min<uint64_t>((uint64_t)24 << 30, std::max<uint64_t>((uint64_t)3 << 30, 8 * gz_bytes));
    const uint64_t not_ours = (uint64_t)1200 << 20;
    // (what the pool may hold on a device while this call runs, in use and idle together: idle buffers of an earlier call's shapes that this call
    // cannot use go back to the runtime when its own would not fit beside them)
    // (a plain file's text is what its buffers hold: two bytes per byte of it, where a .gz plans eight per compressed byte)
    const uint64_t hold = g_knobs.is_set(KN_INGEST_BUDGET_GB) ? budget
                        : std::min<uint64_t>((uint64_t)24 << 30, std::max<uint64_t>((uint64_t)3 << 30, 8 * gz_bytes + 2 * (in_bytes - gz_bytes)));
    for (int d : I.devices) g_pool.set_limit(phys(d), hold - not_ours);
    struct LimitOff { Ingest &I; ~LimitOff() { for (int d : I.devices) g_pool.set_limit(phys(d), 0); } } limit_off{I};
    const uint64_t gz_budget = (budget > 2 * not_ours ? (budget - not_ours) * 10 / 17 : budget / 4) / (uint64_t)I.nm;
    {   // (the two mates' decoders side by side)
        int rcs[2] = {MF_OK, MF_OK}; std::string errs[2]; std::thread th[2];
        for (int i = 0; i < I.nm; i++) {
            Mate &M = I.m[i];
            M.slots.free_ = text_bufs; M.slots.stop = &M.stop;
            if (!M.gz) continue;
            M.gzs.reset(new GzStream());
            auto open = [&I, &M, &rcs, &errs, i, large, gz_budget, text_bufs] { rcs[i] = M.gzs->open(M.map.p, M.map.n, M.map.fd, I.devices, M.path, &M.slots, I.carry_room, I.nm == 2 ? 7 : 12, large, gz_budget, (uint32_t)text_bufs, I.qual != nullptr, &M.stop, errs[i]); };
            if (i == 0 && I.nm == 2 && I.m[1].gz == false) open(); else if (i == 0 && I.nm == 2) th[0] = std::thread(open); else open();
        }
        for (auto &t : th) if (t.joinable()) t.join();
        for (int i = 0; i < I.nm && !rc; i++) if (rcs[i]) { rc = rcs[i]; err = errs[i]; }
    }
    if (alloc_failure(rc)) { TRACE("declined: %s", err.c_str()); return MF_DEVINGEST_DECLINED; }      // (nothing has been touched: the host pipeline streams the file)
    if (rc) return rc;
    for (int i = 0; i < I.nm; i++)
        if (!(I.qual ? I.qual->sink[i].open(out_path[i], &I.qual->chunks) : I.m[i].out.open(out_path[i]))) { err = std::string("Cannot open file ") + (out_path[i] ? out_path[i] : "<stdout>"); return MF_E_IO; }
    const double t_setup = now_s() - t_begin;
    cold_mark("device ingest: decoders and outputs open");
    rc = I.run(err);
    cold_mark("device ingest: consumers done");
    TRACE("run returned %d", rc);
    for (int i = 0; i < I.nm; i++) { I.m[i].stop = true; I.m[i].slots.wake(); }
    bool wrote = true;
    for (int i = 0; i < I.nm; i++) wrote = (I.qual ? I.qual->sink[i].close() : I.m[i].out.close()) && wrote;
    // out of device memory before a byte of the survivors was written: the host pipeline takes the file (it truncates the outputs again)
    if (alloc_failure(rc) && !I.wrote_any) { TRACE("declined after a failed allocation: %s", err.c_str()); return MF_DEVINGEST_DECLINED; }
    if (rc) return rc;
    if (!wrote) { err = std::string("write error on ") + (out_path[0] ? out_path[0] : "<stdout>"); return MF_E_IO; }
    if (stats) {
        *stats = IngestStats();
        stats->seconds = now_s() - t_begin; stats->n_devices = (int)I.devices.size(); stats->consumers = (int)I.workers.size();
        stats->pool_bytes_peak = g_pool.peak(); stats->device_bytes_peak = I.mem_used_max;
        for (int i = 0; i < I.nm; i++) {
            Mate &M = I.m[i];
            stats->input_bytes += M.map.n; stats->records += M.rec_indexed;
            if (M.gzs) {
                stats->text_bytes += M.gzs->text_bytes(); stats->decode_busy_seconds += M.gzs->decode_busy_seconds();
                stats->chunks += M.gzs->chunks(); stats->chunks_linked += M.gzs->chunks_linked(); stats->gaps += M.gzs->gaps(); stats->gap_bytes += M.gzs->gap_bytes();
            } else stats->text_bytes += M.map.n;
        }
    }
    if (rc == MF_OK) g_streams.stage_later();
    if (I.timing) {
        if (I.qual)
            fprintf(stderr, "[mf device ingest] quality filter: wall %.3f s | set-up %.3f | consumers (summed over %zu): waiting for text %.3f, line index %.3f, scan %.3f, decisions %.3f, gather + copy down %.3f, waiting for the writers %.3f; writers busy %.3f %.3f | %llu + %llu bytes written | buffers of this call at most %.2f GB, device memory in use at most %.2f GB",
                    now_s() - t_begin, t_setup, I.workers.size(), I.t_wait, I.t_index, I.qual->t_scan, I.qual->t_decide, I.qual->t_gather, I.qual->t_chunk, I.qual->sink[0].busy(), I.qual->sink[1].busy(),
                    (unsigned long long)I.qual->out_pos[0], (unsigned long long)I.qual->out_pos[1], (double)g_pool.peak() / 1e9, (double)I.mem_used_max / 1e9);
        else
            fprintf(stderr, "[mf device ingest] wall %.3f s | set-up %.3f | waiting for text (upload, inflate, link, CRC on the producer threads) %.3f | line index %.3f | pack %.3f | filter %.3f | survivors %.3f | buffers of this call at most %.2f GB on a device, device memory in use at most %.2f GB, %zu device(s)",
                    now_s() - t_begin, t_setup, I.t_wait, I.t_index, I.t_pack, I.t_filter, I.t_emit, (double)g_pool.peak() / 1e9, (double)I.mem_used_max / 1e9, I.devices.size());
        { double tm; uint64_t nm; g_pool.malloc_time(tm, nm); fprintf(stderr, " | %llu new device allocations took %.3f s (summed over the threads that asked); idle buffers now %.2f GB", (unsigned long long)nm, tm, (double)g_pool.held(phys(I.devices[0])) / 1e9); }
        fprintf(stderr, " | first text after %.3f s, last after %.3f, consumers done after %.3f", I.t_first_piece, I.t_last_piece, I.t_consumed);
        for (int i = 0; i < I.nm; i++)
            if (I.m[i].gzs) { double a, b, c, d, e; I.m[i].gzs->producer_times(a, b, c, d, e); fprintf(stderr, " | mate %d producer: launching (incl. waiting for the upload) %.3f, waiting for decode %.3f, link %.3f; uploader: ring full %.3f, copy wait %.3f, file read %.3f", i + 1, I.m[i].gzs->launch_seconds(), a, b, c, d, e);
                              { double r, k, n; I.m[i].gzs->other_times(r, k, n); fprintf(stderr, "; giving back the buffers of linked slabs %.3f, CRC launch and results %.3f, all of the producer's steps %.3f", r, k, n); }
                              { double oa, os, ou; I.m[i].gzs->open_parts(oa, os, ou); fprintf(stderr, "; set-up %.3f (streams %.3f, uploader's buffers and thread %.3f)", oa, os, ou); }
                              double x, z; I.m[i].gzs->link_parts(x, z); fprintf(stderr, " (of the link time: text buffer %.3f of which waiting for the consumers to hand one back %.3f; waiting for the post stream to be made %.3f)", x, I.m[i].gzs->slot_seconds(), z); }
        for (int i = 0; i < I.nm; i++)
            if (I.m[i].gzs) fprintf(stderr, " | mate %d: inflate kernels busy %.3f s (%.1f GB/s of text), %llu of %u chunks of %zu KiB linked, %llu gaps bridged on the host, %llu bytes decoded there, ring %zu MiB, %u slab splits", i + 1, I.m[i].gzs->decode_busy_seconds(), I.m[i].gzs->decode_busy_seconds() > 0 ? (double)I.m[i].gzs->text_bytes() / I.m[i].gzs->decode_busy_seconds() / 1e9 : 0.0, (unsigned long long)I.m[i].gzs->chunks_linked(),
                                    I.m[i].gzs->chunks(), I.m[i].gzs->chunk_bytes() >> 10, (unsigned long long)I.m[i].gzs->gaps(), (unsigned long long)I.m[i].gzs->gap_bytes(), I.m[i].gzs->ring_bytes() >> 20, I.m[i].gzs->splits());
        fprintf(stderr, "\n");
        end_of_call.timing = true; end_of_call.t0 = now_s();
    }
    return MF_OK;
}

// file-level calls running in this process right now: what they hold in the caches (consumers' scratch and read sets, pinned staging) is in
// use, so a release of everything (mf_release_cached, an allocation elsewhere that found the device full) leaves those alone meanwhile
static std::atomic<int> g_calls_running{0};
struct CallRunning { CallRunning() { g_calls_running++; } ~CallRunning() { g_calls_running--; } };

int run_device_ingest(mf_kmerset *ks, const char *fq1, const char *fq2, const char *out1, const char *out2, uint32_t threshold,
                      bool pair_both, const int *devices, int n_devices, uint64_t *kept, uint64_t *total, std::string &err, IngestStats *stats)
{
    g_knobs.refresh();          // (the environment as it is now: tests change it between calls of one process)
    CallRunning running;
    Ingest I;
    I.ks = ks; I.threshold = threshold; I.pair_both = pair_both;
    for (int i = 0; i < n_devices; i++) I.devices.push_back(devices[i]);
    const int rc = run_ingest(I, fq1, fq2, out1, out2, err, stats);
    if (rc) return rc;
    if (kept) *kept = I.kept;
    if (total) *total = I.total;
    return MF_OK;
}

int run_device_qualfilter(const char *fq1, const char *fq2, const char *out1, const char *out2, const QualParams &P, int device, uint64_t *kept,
                          uint64_t *total, bool *panicked, std::string &err, IngestStats *stats)
{
    g_knobs.refresh();
    if (out1 && has_gz_ext(out1) && g_knobs.u64(KN_QUAL_DEVICE_GZ_OUT, 0) == 0) return MF_DEVINGEST_DECLINED;      // compressing the output is the host pipeline's (many threads)
    if (out2 && has_gz_ext(out2) && g_knobs.u64(KN_QUAL_DEVICE_GZ_OUT, 0) == 0) return MF_DEVINGEST_DECLINED;
    QualState Q;                      // (before the Ingest: its batches hold buffers the state does not own, but the set's go back to the pool last)
    Q.P = P; Q.pe = fq2 != nullptr; Q.cap = P.end ? P.end - P.start : ~0ull;
    Q.chunks.init((size_t)std::max<uint64_t>(g_knobs.u64(KN_QUAL_OUT_CHUNK, (uint64_t)4 << 20), 4096), (int)std::max<uint64_t>(2, g_knobs.u64(KN_QUAL_OUT_CHUNKS, 24)),
                  [](size_t n) -> void * { void *q = nullptr; return hipHostMalloc(&q, n, hipHostMallocPortable) == hipSuccess ? q : nullptr; }, [](void *q) { (void)hipHostFree(q); });
    CallRunning running;
    Ingest I;
    I.qual = &Q;
    I.devices.push_back(device);
    const int rc = run_ingest(I, fq1, fq2, out1, out2, err, stats);
    if (rc) return rc;
    if (kept) *kept = I.kept;
    if (total) *total = I.total;
    if (panicked) *panicked = Q.panicked;
    return MF_OK;
}

void ingest_short_lived(bool yes) { g_short_lived = yes; }

void ingest_prefetch(int device)
{
    const int dev = phys(device);
    int cur = -1; (void)hipGetDevice(&cur);
    if (hipSetDevice(dev) != hipSuccess) { (void)hipGetLastError(); return; }
    std::string err;
    (void)g_streams.get(dev, false, err);
    g_streams.prefill_pinned(dev);
    if (cur >= 0 && cur != dev) (void)hipSetDevice(cur);
}

size_t release_cached_device_memory(bool all)
{
    // (the pool's free lists hold idle buffers only -- giving those back is always safe, if slow beside a running call: hipFree waits for the
    // device --; the scratch, read-set and pinned caches are taken apart only when no file-level call of this process is running)
    if (all) { if (g_calls_running.load() == 0) { g_scratch.clear(); g_streams.forget_staging(); g_pinned.clear(); } return g_pool.release_all(); }
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return g_pool.release(dev);
}

} // namespace mf
