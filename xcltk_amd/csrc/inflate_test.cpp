// inflate_test.cpp - checks inflate_fast.h against zlib: (1) every BGZF block of the BAM files given on the
// command line, (2) synthetic buffers deflated with all levels / strategies (stored, fixed, dynamic blocks).
// Prints throughput of both decoders. Exit code 0 = every block identical.
#include <zlib.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "inflate_fast.h"
using namespace xck;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static bool zinflate(const uint8_t* in, size_t n, uint8_t* out, size_t on) { z_stream z; memset(&z, 0, sizeof z); inflateInit2(&z, -15);
    z.next_in = (Bytef*)in; z.avail_in = (uInt)n; z.next_out = out; z.avail_out = (uInt)on; int rc = inflate(&z, Z_FINISH); bool ok = rc == Z_STREAM_END && z.avail_out == 0; inflateEnd(&z); return ok; }
int main(int argc, char** argv) {
    static InflateTables tabs; long bad = 0, blocks = 0, fallback = 0; double tz = 0, tf = 0; size_t bytes = 0;
    std::vector<uint8_t> a(1 << 17), b(1 << 17);
    for (int f = 1; f < argc; f++) {
        FILE* fp = fopen(argv[f], "rb"); if (!fp) { perror(argv[f]); return 2; }
        std::vector<uint8_t> d; { fseek(fp, 0, SEEK_END); long n = ftell(fp); fseek(fp, 0, SEEK_SET); d.resize(n); if (fread(d.data(), 1, n, fp) != (size_t)n) return 2; fclose(fp); }
        size_t off = 0;
        while (off + 18 <= d.size()) {
            const uint8_t* p = d.data() + off; uint32_t xlen = p[10] | (p[11] << 8); uint32_t bsize = 0; const uint8_t* x = p + 12;
            while (x + 4 <= p + 12 + xlen) { uint32_t sl = x[2] | (x[3] << 8); if (x[0] == 'B' && x[1] == 'C') bsize = (x[4] | (x[5] << 8)) + 1; x += 4 + sl; }
            if (!bsize || off + bsize > d.size()) break;
            const uint8_t* cd = p + 12 + xlen; size_t cl = bsize - 12 - xlen - 8; uint32_t isz; memcpy(&isz, p + bsize - 4, 4);
            if (isz) { double t0 = now(); bool okz = zinflate(cd, cl, a.data(), isz); double t1 = now(); int rc = inflate_raw(cd, cl, b.data(), isz, &tabs); double t2 = now();
                tz += t1 - t0; tf += t2 - t1; bytes += isz; blocks++;
                if (rc != 0) fallback++; else if (!okz || memcmp(a.data(), b.data(), isz) != 0) { bad++; fprintf(stderr, "MISMATCH %s block at %zu\n", argv[f], off); } }
            off += bsize;
        }
    }
#ifdef XCK_INFLATE_PROF
    printf("prof (BAM part): dynamic blocks %llu over %ld BGZF blocks, build cycles %llu = %.1f%% of fast time at 2.1GHz\n", g_prof_nblocks, blocks, g_prof_build, 100.0 * g_prof_build / 2.1e9 / (tf > 0 ? tf : 1));
#endif
    // synthetic: random texts with varying entropy / repetition, all levels and strategies
    std::mt19937_64 rng(7);
    for (int it = 0; it < 600; it++) {
        size_t n = 1 + rng() % 65000; std::vector<uint8_t> src(n);
        int mode = it % 6; uint32_t alpha = mode == 0 ? 2 : mode == 1 ? 4 : mode == 2 ? 20 : mode == 3 ? 256 : 64;
        for (size_t i = 0; i < n; i++) { if (mode >= 4 && i > 300 && rng() % 3) { size_t back = 1 + rng() % (mode == 4 ? 8 : 300); src[i] = src[i - back]; } else src[i] = (uint8_t)(rng() % alpha); }
        int level = (int)(rng() % 10); static const int strat[5] = {Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED};
        z_stream z; memset(&z, 0, sizeof z); deflateInit2(&z, level, Z_DEFLATED, -15, 1 + (int)(rng() % 9), strat[rng() % 5]);
        std::vector<uint8_t> c(deflateBound(&z, n) + 64); z.next_in = src.data(); z.avail_in = (uInt)n; z.next_out = c.data(); z.avail_out = (uInt)c.size();
        if (it % 7 == 0) { z.avail_in = (uInt)(n / 2); deflate(&z, Z_FULL_FLUSH); z.avail_in = (uInt)(n - n / 2); }   // several deflate blocks incl. empty stored
        deflate(&z, Z_FINISH); size_t cl = z.total_out; deflateEnd(&z);
        std::vector<uint8_t> o(n + 1, 0xEE);
        int rc = inflate_raw(c.data(), cl, o.data(), n, &tabs); blocks++;
        if (rc != 0) { fallback++; fprintf(stderr, "synthetic %d: rc %d (level %d)\n", it, rc, level); }
        else if (memcmp(o.data(), src.data(), n) != 0 || o[n] != 0xEE) { bad++; fprintf(stderr, "synthetic %d MISMATCH\n", it); }
        // truncated / corrupted input must fail cleanly or still be exact, never write past the end
        if (cl > 8) { std::vector<uint8_t> o2(n + 1, 0xEE); std::vector<uint8_t> c2(c.begin(), c.begin() + cl); c2[cl / 2] ^= 0x5a; (void)inflate_raw(c2.data(), cl, o2.data(), n, &tabs); if (o2[n] != 0xEE) { bad++; fprintf(stderr, "synthetic %d wrote past the end\n", it); }
                      (void)inflate_raw(c.data(), cl / 2, o2.data(), n, &tabs); if (o2[n] != 0xEE) { bad++; fprintf(stderr, "synthetic %d (truncated) wrote past the end\n", it); } }
    }
#ifdef XCK_INFLATE_PROF
    printf("prof: dynamic blocks %llu, build cycles %llu (%.1f%% of fast time at 2.1GHz)\n", g_prof_nblocks, g_prof_build, 100.0 * g_prof_build / 2.1e9 / (tf > 0 ? tf : 1));
#endif
    printf("blocks %ld mismatches %ld fast-decoder-declined %ld | BAM bytes %.1f MB: zlib %.0f MB/s, fast %.0f MB/s\n", blocks, bad, fallback,
           bytes / 1e6, tz > 0 ? bytes / 1e6 / tz : 0, tf > 0 ? bytes / 1e6 / tf : 0);
    return bad ? 1 : 0;
}
