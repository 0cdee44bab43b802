// gpu_inflate_bench.hip - correctness + throughput of csrc/inflate_dev.hip against zlib on the BGZF blocks of a BAM.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ixcltk_amd/csrc tools/gpu_inflate_bench.hip xcltk_amd/csrc/inflate_dev.hip -lz -o tools/scratch/gpu_inflate_bench
//   [INFLATE_VARIANT=0|1] tools/scratch/gpu_inflate_bench FILE.bam [max_bytes [blocks_per_launch]]     (variant: see dev_inflate_launch)
#include <zlib.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <hip/hip_runtime.h>
#include "inflate_dev.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const int variant = getenv("INFLATE_VARIANT") ? atoi(getenv("INFLATE_VARIANT")) : 0;
    FILE* f = fopen(argv[1], "rb"); if (!f) { perror("open"); return 1; }
    size_t maxb = argc > 2 ? strtoull(argv[2], nullptr, 10) : (size_t)2 << 30;
    std::vector<uint8_t> d(maxb); size_t n = fread(d.data(), 1, maxb, f); fclose(f);
    std::vector<xck::DevBlock> bl; size_t o = 0, tot = 0;
    while (o + 18 < n) { size_t bs = (size_t)(d[o + 16] | (d[o + 17] << 8)) + 1; if (o + bs > n) break; uint32_t isz; memcpy(&isz, &d[o + bs - 4], 4);
        const uint32_t xlen = d[o + 10] | (d[o + 11] << 8);
        bl.push_back({(uint32_t)(o + 12 + xlen), (uint32_t)(bs - 12 - xlen - 8), (uint32_t)tot, isz}); tot += isz; o += bs; if (tot > (size_t)3 << 30) break; }
    printf("%zu blocks, %.1f MB compressed, %.1f MB inflated\n", bl.size(), o / 1e6, tot / 1e6);
    uint8_t *d_in, *d_out; xck::DevBlock* d_bl; int32_t* d_st;
    CK(hipMalloc((void**)&d_in, o)); CK(hipMalloc((void**)&d_out, tot + 64)); CK(hipMalloc((void**)&d_bl, bl.size() * sizeof(xck::DevBlock))); CK(hipMalloc((void**)&d_st, bl.size() * 4));
    CK(hipMemcpy(d_in, d.data(), o, hipMemcpyHostToDevice)); CK(hipMemcpy(d_bl, bl.data(), bl.size() * sizeof(xck::DevBlock), hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++) {
        CK(hipMemsetAsync(d_st, 0xff, bl.size() * 4, s));
        CK(hipEventRecord(e0, s));
        // chunks of 740 blocks like the ingest would launch them
        const int per = argc > 3 ? atoi(argv[3]) : (int)bl.size();
        for (size_t b0 = 0; b0 < bl.size(); b0 += per) { int nb = (int)std::min<size_t>(per, bl.size() - b0);
            if (xck::dev_inflate_launch(s, d_in, d_bl + b0, nb, d_out, d_st + b0, nullptr, variant)) { fprintf(stderr, "launch failed\n"); return 1; } }
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (variant >= 10) { unsigned long long pr[8]; xck::dev_inflate_read_prof(pr); double t = 0; for (int k = 0; k < 8; k++) t += (double)pr[k];
            static const char* nm[8] = {"header", "tables", "window", "decode", "walk", "scan+literals", "matches", "flush+rest"};
            printf("  wave cycles per block %.0f:", t / bl.size()); for (int k = 0; k < 8; k++) printf(" %s %.1f%%", nm[k], 100.0 * pr[k] / t); printf("\n"); }
        printf("rep %d: %.2f ms  = %.1f GB/s inflated, %.1f GB/s compressed\n", rep, ms, tot / ms / 1e6, o / ms / 1e6);
    }
    std::vector<uint8_t> got(tot); std::vector<int32_t> st(bl.size());
    CK(hipMemcpy(got.data(), d_out, tot, hipMemcpyDeviceToHost)); CK(hipMemcpy(st.data(), d_st, bl.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0, left = 0; std::vector<uint8_t> ref(65536 + 8); int hist[64] = {0};
    for (size_t b = 0; b < bl.size(); b++) {
        if (st[b] != 0) { left++; hist[st[b] & 63]++; continue; }
        z_stream zs; memset(&zs, 0, sizeof zs); inflateInit2(&zs, -15); zs.next_in = &d[bl[b].in_off]; zs.avail_in = bl[b].in_len; zs.next_out = ref.data(); zs.avail_out = 65536;
        int rc = inflate(&zs, Z_FINISH); inflateEnd(&zs);
        if (rc != Z_STREAM_END || zs.total_out != bl[b].out_len || memcmp(ref.data(), &got[bl[b].out_off], bl[b].out_len)) { if (bad < 5) { size_t k = 0; while (k < bl[b].out_len && ref[k] == got[bl[b].out_off + k]) k++; fprintf(stderr, "block %zu differs (rc %d) at byte %zu of %u\n", b, rc, k, bl[b].out_len); } bad++; }
    }
    printf("verified against zlib: %zu blocks wrong, %zu left to the host (status histogram:", bad, left);
    for (int i = 0; i < 64; i++) if (hist[i]) printf(" %d:%d", i, hist[i]);
    printf(")\n");
    return bad ? 1 : 0;
}
