// inflate_stats.cpp - what the DEFLATE streams of a BAM's BGZF blocks are made of: literals, matches, match lengths and distances.
// Sizing input for the GPU decoder (csrc/inflate_dev.hip: which share of the matches an LDS history of H bytes would serve).
//   g++ -O2 -std=c++17 -Ixcltk_amd/csrc tools/inflate_stats.cpp -o /tmp/inflate_stats && /tmp/inflate_stats FILE.bam [max_bytes]
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <vector>
static unsigned long long g_n = 0, g_bytes = 0, g_dist_le[16] = {0}, g_bytes_le[16] = {0};
#define XCK_INFLATE_MATCH_HOOK(offset, length) do { g_n++; g_bytes += (length); for (int k_ = 0; k_ < 16; k_++) if ((offset) <= (1u << k_)) { g_dist_le[k_]++; g_bytes_le[k_] += (length); } } while (0)
#include "inflate_fast.h"
int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb"); if (!f) { perror("open"); return 1; }
    size_t maxb = argc > 2 ? strtoull(argv[2], nullptr, 10) : (size_t)256 << 20;
    std::vector<uint8_t> d(maxb); size_t n = fread(d.data(), 1, maxb, f); fclose(f);
    std::vector<uint8_t> out(65536 + 512); xck::InflateTables* t = new xck::InflateTables();
    size_t o = 0, tot = 0, comp = 0, nb = 0;
    while (o + 18 < n) { size_t bs = (size_t)(d[o + 16] | (d[o + 17] << 8)) + 1; if (o + bs > n) break; uint32_t isz; memcpy(&isz, &d[o + bs - 4], 4);
        const uint32_t xlen = d[o + 10] | (d[o + 11] << 8);
        const int rc = xck::inflate_raw(&d[o + 12 + xlen], bs - 12 - xlen - 8, out.data(), isz, t);
        if (rc) { fprintf(stderr, "block at %zu: rc %d\n", o, rc); return 1; }
        tot += isz; comp += bs - 12 - xlen - 8; o += bs; nb++; }
    const double lit = (double)tot - (double)g_bytes;
    printf("%zu blocks, %.1f MB compressed, %.1f MB inflated: %.1f %% of the bytes are literals, %llu matches (mean length %.1f), %.2f compressed bits per symbol, %.2f output bytes per symbol\n",
           nb, comp / 1e6, tot / 1e6, 100.0 * lit / tot, g_n, (double)g_bytes / g_n, comp * 8.0 / (lit + g_n), tot / (lit + g_n));
    printf("matches with distance <= 2^k (share of the matches / of the matched bytes):");
    for (int k = 6; k < 16; k++) printf("  %d: %.2f / %.2f", 1 << k, (double)g_dist_le[k] / g_n, (double)g_bytes_le[k] / g_bytes);
    printf("\n");
    return 0;
}
