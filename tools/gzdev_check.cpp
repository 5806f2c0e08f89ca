// Stand-alone check and timing of the device DEFLATE decoder (mitoflex_amd/csrc/mf_gzdev.hip) against zlib.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip tools/gzdev_check.cpp mitoflex_amd/csrc/mf_gzdev.hip -lz -o tools/gzdev_check
//   tools/gzdev_check file.gz [chunk_KiB=256] [expansion=8] [reps=3]
// Decodes every chunk on the GPU and checks every chunk's symbols against zlib's output (markers resolved from the reference text);
// then links the chunks the way the product does -- the walk over the descriptors on the host, the window scan and the marker resolution
// on the device (launch_gz_link, launch_gz_resolve) -- and compares the device's text with zlib's, byte for byte.
#include "../mitoflex_amd/csrc/mf_gzdev.h"
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <zlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s file.gz [chunk_KiB] [expansion] [reps]\n", argv[0]); return 2; }
    const size_t chunk = (argc > 2 ? strtoull(argv[2], 0, 10) : 256) << 10;
    const size_t expansion = argc > 3 ? strtoull(argv[3], 0, 10) : 8;
    const int reps = argc > 4 ? atoi(argv[4]) : 3;
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END); const size_t size = (size_t)ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> gz(size + 64, 0);
    if (fread(gz.data(), 1, size, f) != size) { perror("read"); return 2; }
    fclose(f);
    // gzip header -> first deflate bit
    if (size < 18 || gz[0] != 0x1f || gz[1] != 0x8b || gz[2] != 8) { fprintf(stderr, "not gzip\n"); return 2; }
    size_t p = 10; const unsigned flg = gz[3];
    if (flg & 4) { p += 2 + (gz[p] | (gz[p + 1] << 8)); }
    if (flg & 8) { while (gz[p]) p++; p++; }
    if (flg & 16) { while (gz[p]) p++; p++; }
    if (flg & 2) p += 2;
    // reference text (first member only)
    std::vector<uint8_t> ref;
    {
        z_stream z; memset(&z, 0, sizeof z);
        inflateInit2(&z, -15);
        z.next_in = gz.data() + p; z.avail_in = (uInt)(size - p > 0xFFFFFFFFu ? 0xFFFFFFFFu : size - p);
        ref.resize(size * 3 + (1 << 20));
        size_t got = 0; int rc;
        for (;;) {
            if (ref.size() - got < (1u << 20)) ref.resize(ref.size() * 2);
            z.next_out = ref.data() + got; z.avail_out = (uInt)((ref.size() - got) > 0x40000000u ? 0x40000000u : (ref.size() - got));
            const size_t before = z.total_out;
            rc = inflate(&z, Z_NO_FLUSH);
            got += z.total_out - before;
            if (rc != Z_OK) break;
        }
        if (rc != Z_STREAM_END) { fprintf(stderr, "zlib: %d\n", rc); return 2; }
        ref.resize(got);
        inflateEnd(&z);
    }
    const size_t base_byte = p;
    const uint32_t n_chunks = (uint32_t)((size - base_byte + chunk - 1) / chunk);
    const size_t cap = chunk * expansion + 262144;          // (the product's rule, mf_devingest.cpp: a chunk may decode one more chunk's worth of stored or fixed blocks behind its range)
    printf("%s: %zu bytes -> %zu bytes of text (%.2fx), %u chunks of %zu KiB, %zu symbols of room each\n", argv[1], size, ref.size(),
           (double)ref.size() / size, n_chunks, chunk >> 10, cap);
    uint8_t *d_data; uint16_t *d_sym; mf::GzChunk *d_chunks;
    CK(hipMalloc(&d_data, size + 64)); CK(hipMemcpy(d_data, gz.data(), size + 64, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_sym, (size_t)n_chunks * cap * 2));
    CK(hipMalloc(&d_chunks, n_chunks * sizeof(mf::GzChunk)));
    uint32_t *d_scratch = nullptr;          // the lane-parallel kernel's code lists (MF_GZDEV_KERNEL=serial: the one-lane walk)
    if (!mf::gz_decode_serial()) CK(hipMalloc(&d_scratch, mf::gz_decode_scratch_bytes(n_chunks)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        CK(hipMemset(d_chunks, 0, n_chunks * sizeof(mf::GzChunk)));
        CK(hipEventRecord(e0, 0));
        CK(mf::launch_gz_decode(d_data, 0, size, size, base_byte, chunk, 0, n_chunks, 0, (uint64_t)base_byte * 8, d_sym, cap, d_chunks, d_scratch, 0));
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        printf("  decode kernel: %.3f ms\n", ms);
    }
#ifdef GZ_PROFILE
    if (d_scratch) {          // cycle counts per phase, summed over the chunks (the kernel leaves them at the head of every chunk's list scratch)
        const size_t per = mf::gz_decode_scratch_bytes(1);
        unsigned long long tot[17] = {0}, v[17];
        unsigned long long slow[17] = {0};
        for (uint32_t c = 0; c < n_chunks; c++) {
            CK(hipMemcpy(v, (const char *)d_scratch + (size_t)c * per, sizeof v, hipMemcpyDeviceToHost));
            for (int i = 0; i < 17; i++) tot[i] += v[i];
            if (v[16] > slow[16]) memcpy(slow, v, sizeof slow);
        }
        printf("  the slowest chunk: search + header %llu | tables %llu | walks %llu | expansion %llu | whole %llu ; %llu blocks, %llu steps\n", slow[0], slow[1], slow[2], slow[3], slow[16], slow[7], slow[4]);
        printf("  cycles per chunk (shader clock): search + header %.0f | tables %.0f | walks %.0f | expansion %.0f | whole %.0f ;  per chunk: %.1f blocks, %.1f steps, %.1f walk rounds, %.1f expansion rounds\n",
               (double)tot[0] / n_chunks, (double)tot[1] / n_chunks, (double)tot[2] / n_chunks, (double)tot[3] / n_chunks, (double)tot[16] / n_chunks,
               (double)tot[7] / n_chunks, (double)tot[4] / n_chunks, (double)tot[5] / n_chunks, (double)tot[6] / n_chunks);
        printf("  inside the expansion rounds: list + sums %.0f | cells %.0f | loads of earlier output %.0f | chase %.0f (%.1f passes a chunk) | store %.0f\n",
               (double)tot[8] / n_chunks, (double)tot[9] / n_chunks, (double)tot[13] / n_chunks, (double)tot[10] / n_chunks, (double)tot[12] / n_chunks, (double)tot[11] / n_chunks);
    }
#endif
    std::vector<mf::GzChunk> ch(n_chunks);
    CK(hipMemcpy(ch.data(), d_chunks, n_chunks * sizeof(mf::GzChunk), hipMemcpyDeviceToHost));
    std::vector<uint16_t> sym(cap);
    uint64_t cur = (uint64_t)base_byte * 8, total = 0, linked = 0, sym_total = 0; bool ok = true, ended = false, gap = false;
    uint32_t hist[5] = {0, 0, 0, 0, 0};
    for (uint32_t c = 0; c < n_chunks; c++) {
        hist[ch[c].status < 5 ? ch[c].status : 0]++;
        if (ch[c].status == mf::GZ_FAILED) printf("chunk %u failed: reason %u at bit %llu (started at %llu)\n", c, ch[c].n_sym, (unsigned long long)ch[c].end_bit, (unsigned long long)ch[c].start_bit);
        else sym_total += ch[c].n_sym;
    }
    for (uint32_t c = 0; c < n_chunks && ok && !ended; c++) {
        const mf::GzChunk &k = ch[c];
        // the rules of gz_link_walk: a chunk that found nothing, failed, or started inside accepted data is passed over; one that
        // starts behind the accepted data is a gap, which the product bridges on the host (inflate_gap) -- this check stops there
        if (k.status == mf::GZ_NONE || k.status == mf::GZ_FAILED || k.start_bit < cur) continue;
        if (k.start_bit > cur) {
            printf("chunk %u starts %lld bits behind the accepted data (status %u): a gap for the host's inflate_gap; not followed here\n", c,
                   (long long)(k.start_bit - cur), k.status);
            gap = true; break;
        }
        CK(hipMemcpy(sym.data(), d_sym + (size_t)c * cap, (size_t)k.n_sym * 2, hipMemcpyDeviceToHost));
        if (total + k.n_sym > ref.size()) { printf("chunk %u: more output than the reference holds\n", c); ok = false; break; }
        for (uint32_t i = 0; i < k.n_sym; i++) {
            const uint16_t v = sym[i];
            uint8_t b;
            if (v & mf::GZ_MARK) {
                const int64_t src = (int64_t)total - 32768 + (v & 0x7FFF);
                if (src < 0) { printf("chunk %u symbol %u: marker reaches in front of the text\n", c, i); ok = false; break; }
                b = ref[(size_t)src];
            } else b = (uint8_t)v;
            if (v < 0x8000 && v > 255) { printf("chunk %u symbol %u: stray value %04x\n", c, i, v); ok = false; break; }
            if (b != ref[total + i]) { printf("chunk %u symbol %u (text offset %llu): %02x, expected %02x\n", c, i, (unsigned long long)(total + i), b, ref[total + i]); ok = false; break; }
        }
        total += k.n_sym; cur = k.end_bit; linked++;
        if (k.status == mf::GZ_MEMBER_END) ended = true;
        if (k.status == mf::GZ_OVERFLOW) { printf("chunk %u overflowed its symbol buffer\n", c); ok = false; }
    }
    printf("status histogram: none %u, boundary %u, member end %u, failed %u, overflow %u\n", hist[0], hist[1], hist[2], hist[3], hist[4]);
    printf("linked %llu of %u chunks, %llu of %zu bytes verified, member end %s\n", (unsigned long long)linked, n_chunks,
           (unsigned long long)total, ref.size(), ended ? "reached" : "NOT reached");
    printf("kernel %.3f ms: %.2f GB/s of text, %.2f GB/s of compressed input (%llu symbols written)\n", best, sym_total / best / 1e6, size / best / 1e6,
           (unsigned long long)sym_total);
    // ---- the link step on the device (launch_gz_link: the two-level window scan + marker resolution), the way the product drives it: the walk
    // over the descriptors on the host (gz_link_walk), slabs of `slab` chunks, the window carried on the device from slab to slab.  The text
    // the device produces is compared with zlib's byte for byte, and the link kernels are timed (events around every slab's launches).
    bool link_ok = true; double link_ms = 0; uint64_t link_bytes = 0; uint32_t link_chunks = 0, link_slabs = 0;
    {
        const uint32_t slab = (uint32_t)(getenv("GZCHECK_SLAB") ? atoi(getenv("GZCHECK_SLAB")) : 512);
        uint8_t *d_text, *d_window, *d_link; uint32_t *d_acc; uint64_t *d_acc_off;
        const size_t front = 32768 + 256;
        CK(hipMalloc(&d_text, ref.size() + front + 64)); CK(hipMemset(d_text, 0, front));
        CK(hipMalloc(&d_window, 32768)); CK(hipMemset(d_window, 0, 32768));
        CK(hipMalloc(&d_link, mf::gz_link_scratch_bytes(slab))); CK(hipMalloc(&d_acc, (size_t)slab * 4)); CK(hipMalloc(&d_acc_off, (size_t)slab * 8));
        mf::GzLinkState st; st.cur_bit = (uint64_t)base_byte * 8;
        std::vector<uint32_t> acc; std::vector<uint64_t> acc_off;
        for (uint32_t lo = 0; lo < n_chunks && st.stop == mf::GZ_STOP_NONE; lo += slab) {
            const uint32_t hi = lo + slab < n_chunks ? lo + slab : n_chunks;
            const uint32_t wlen_before = st.wlen;
            if (st.next < lo) st.next = lo;
            mf::gz_link_walk(ch.data(), hi, st, acc, acc_off);
            if (acc.empty()) continue;
            uint32_t mx = 0; for (uint32_t c : acc) mx = ch[c].n_sym > mx ? ch[c].n_sym : mx;
            CK(hipMemcpy(d_acc, acc.data(), acc.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_acc_off, acc_off.data(), acc_off.size() * 8, hipMemcpyHostToDevice));
            CK(hipEventRecord(e0, 0));
            CK(mf::launch_gz_link(d_acc, d_acc_off, (uint32_t)acc.size(), mx, d_chunks, lo, d_sym + (size_t)lo * cap, cap, d_window, wlen_before, d_link, d_text + front, 0, acc_off[0], 0));
            CK(mf::launch_gz_resolve(d_acc, d_acc_off, (uint32_t)acc.size(), mx, d_chunks, lo, d_sym + (size_t)lo * cap, cap, d_text + front, 0, 0));
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            link_ms += ms; link_chunks += (uint32_t)acc.size(); link_slabs++;
        }
        link_bytes = st.total;
        std::vector<uint8_t> got(link_bytes);
        if (link_bytes) CK(hipMemcpy(got.data(), d_text + front, link_bytes, hipMemcpyDeviceToHost));
        if (link_bytes > ref.size()) { printf("device link: more text than the reference holds\n"); link_ok = false; }
        else for (uint64_t i = 0; i < link_bytes; i++) if (got[i] != ref[i]) { printf("device link: text offset %llu: %02x, expected %02x\n", (unsigned long long)i, got[i], ref[i]); link_ok = false; break; }
        if (link_bytes != total) { printf("device link: %llu bytes of text, the host's walk over the same descriptors %llu\n", (unsigned long long)link_bytes, (unsigned long long)total); link_ok = false; }
        std::vector<uint8_t> w(32768);
        CK(hipMemcpy(w.data(), d_window, 32768, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < st.wlen && link_ok; i++) if (w[32768 - st.wlen + i] != ref[link_bytes - st.wlen + i]) { printf("device link: the window left behind differs at %u\n", i); link_ok = false; }
        printf("device link (launch_gz_link, slabs of %u chunks): %u chunks in %u slabs, %llu bytes of text %s zlib's; link + resolve kernels %.3f ms in all = %.2f us a chunk, %.1f GB/s of text\n", slab,
               link_chunks, link_slabs, (unsigned long long)link_bytes, link_ok ? "equal to" : "DIFFER from", link_ms, link_chunks ? link_ms * 1e3 / link_chunks : 0.0, link_ms > 0 ? link_bytes / link_ms / 1e6 : 0.0);
        ok = ok && link_ok;
    }
    const bool pass = ok && ended && total == ref.size();
    const bool partial = ok && !pass && (gap || !ended) && total <= ref.size();      // everything the device linked is right; the rest is the host's (stored-only files, a lost candidate)
    printf(pass ? "PASS\n" : partial ? "PASS (as far as the chunks link without the host: %llu of %zu bytes)\n" : "FAIL\n", (unsigned long long)total, ref.size());
    return pass || partial ? 0 : 1;
}
