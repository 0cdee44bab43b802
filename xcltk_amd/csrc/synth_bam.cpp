// synth_bam.cpp - fast seeded synthetic 10x-style BAM writer (benchmark tooling, not part of the ABI).
//
// Produces a coordinate-sorted BAM with CB/UB tags whose shape follows SURVEY.md section 8d
// (1-8 reads per UMI, read length 91, CIGAR mix 80/12/3/3/2, MAPQ 255/low, flags, 3 % missing
// CB / UB, 2 % barcodes outside the list), large enough (millions of reads) to measure the
// end-to-end ingest rate on the GPU box, where no BAM files exist.  Blocks are deflated in
// parallel.  Usage:
//   xck_synth_bam OUT.bam CONTIGS.tsv REGIONS.tsv BARCODES.tsv N_READS SEED [THREADS] [LEVEL]
// CONTIGS.tsv: name<TAB>length ; REGIONS.tsv: chrom start end name (chrom must match a contig
// name after stripping "chr") ; BARCODES.tsv: one barcode per line.
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

static inline uint64_t mix(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
struct Rng { uint64_t s; explicit Rng(uint64_t seed) : s(seed) {} uint64_t next() { s = mix(s); return s; } uint32_t below(uint32_t n) { return (uint32_t)((next() >> 11) % n); } double uni() { return (next() >> 11) * (1.0 / 9007199254740992.0); } };

struct Rec { int32_t tid, pos; uint32_t mol; uint32_t k; };
struct Gene { int tid; int32_t s, e; };

static std::string strip_chr(const std::string& c) { if (c.size() >= 3 && (c[0] == 'c' || c[0] == 'C') && (c[1] == 'h' || c[1] == 'H') && (c[2] == 'r' || c[2] == 'R')) return c.substr(3); return c; }

static void put32(std::string& b, uint32_t v) { b.append((const char*)&v, 4); }
static void put16(std::string& b, uint16_t v) { b.append((const char*)&v, 2); }

static int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

static void bgzf_block(const char* p, size_t n, int level, std::string& out) {
    z_stream zs; memset(&zs, 0, sizeof zs);
    deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
    std::string c(deflateBound(&zs, n) + 64, '\0');
    zs.next_in = (Bytef*)p; zs.avail_in = (uInt)n; zs.next_out = (Bytef*)&c[0]; zs.avail_out = (uInt)c.size();
    deflate(&zs, Z_FINISH);
    size_t clen = zs.total_out; deflateEnd(&zs);
    uint16_t bsize = (uint16_t)(clen + 25);
    const unsigned char hdr[12] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0};
    out.append((const char*)hdr, 12); out.push_back('B'); out.push_back('C'); put16(out, 2); put16(out, bsize);
    out.append(c.data(), clen);
    put32(out, (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef*)p, (uInt)n)); put32(out, (uint32_t)n);
}

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: %s OUT.bam CONTIGS.tsv REGIONS.tsv BARCODES.tsv N_READS SEED [THREADS] [LEVEL]\n", argv[0]); return 2; }
    const std::string out_fn = argv[1];
    const int64_t n_reads = atoll(argv[5]); const uint64_t seed = strtoull(argv[6], nullptr, 10);
    int n_thr = argc > 7 ? atoi(argv[7]) : (int)std::thread::hardware_concurrency(); if (n_thr <= 0) n_thr = 4;
    const int level = argc > 8 ? atoi(argv[8]) : 6;
    const bool no_tags = getenv("XCK_SYNTH_NOTAGS") && atoi(getenv("XCK_SYNTH_NOTAGS"));
    std::vector<std::string> cname; std::vector<int32_t> clen;
    { std::ifstream f(argv[2]); std::string a; int64_t l; while (f >> a >> l) { cname.push_back(a); clen.push_back((int32_t)l); } }
    std::vector<Gene> genes;
    { std::ifstream f(argv[3]); std::string line;
      while (std::getline(f, line)) { std::istringstream ss(line); std::string c; int32_t s, e; if (!(ss >> c >> s >> e)) continue;
        std::string sc = strip_chr(c); int tid = -1; for (size_t i = 0; i < cname.size(); i++) if (strip_chr(cname[i]) == sc) { tid = (int)i; break; }
        if (tid >= 0) genes.push_back({tid, s, e}); } }
    std::vector<std::string> bcs;
    { std::ifstream f(argv[4]); std::string b; while (f >> b) bcs.push_back(b); }
    if (cname.empty() || genes.empty() || bcs.empty()) { fprintf(stderr, "empty contigs / regions / barcodes\n"); return 2; }
    const int L = 91;
    // expression weights ~ Zipf over a seeded permutation of the genes
    std::vector<double> cdf(genes.size());
    { std::vector<uint32_t> perm(genes.size()); for (size_t i = 0; i < perm.size(); i++) perm[i] = (uint32_t)i;
      Rng r(seed ^ 0xabcdef); for (size_t i = perm.size(); i > 1; i--) std::swap(perm[i - 1], perm[r.below((uint32_t)i)]);
      double acc = 0; std::vector<double> w(genes.size()); for (size_t i = 0; i < perm.size(); i++) w[perm[i]] = 1.0 / std::pow((double)(i + 1), 0.8);
      for (size_t i = 0; i < w.size(); i++) { acc += w[i]; cdf[i] = acc; } for (auto& x : cdf) x /= acc; }
    // molecules -> reads
    std::vector<Rec> recs; recs.reserve((size_t)n_reads + 8);
    { Rng r(seed);
      for (uint32_t mol = 0; (int64_t)recs.size() < n_reads; mol++) {
          size_t g = std::lower_bound(cdf.begin(), cdf.end(), r.uni()) - cdf.begin(); if (g >= genes.size()) g = genes.size() - 1;
          const Gene& ge = genes[g];
          int64_t span = (int64_t)ge.e - ge.s + 40;
          int64_t anchor = (int64_t)ge.s - 41 + (int64_t)(r.uni() * span);
          uint32_t k = 1 + r.below(8);
          for (uint32_t j = 0; j < k && (int64_t)recs.size() < n_reads; j++) {
              int64_t pos = anchor + r.below(200); if (pos < 0) pos = 0; if (pos > clen[ge.tid] - L - 25000) pos = std::max<int64_t>(0, clen[ge.tid] - L - 25000);
              recs.push_back({ge.tid, (int32_t)pos, mol, j});
          } } }
    std::stable_sort(recs.begin(), recs.end(), [](const Rec& a, const Rec& b) { return a.tid != b.tid ? a.tid < b.tid : a.pos < b.pos; });
    // header
    std::string hdr;
    { std::string text = "@HD\tVN:1.6\tSO:coordinate\n"; for (size_t i = 0; i < cname.size(); i++) text += "@SQ\tSN:" + cname[i] + "\tLN:" + std::to_string(clen[i]) + "\n";
      hdr.append("BAM\1", 4); put32(hdr, (uint32_t)text.size()); hdr += text; put32(hdr, (uint32_t)cname.size());
      for (size_t i = 0; i < cname.size(); i++) { put32(hdr, (uint32_t)cname[i].size() + 1); hdr += cname[i]; hdr.push_back('\0'); put32(hdr, (uint32_t)clen[i]); } }
    FILE* fp = fopen(out_fn.c_str(), "wb"); if (!fp) { perror("open"); return 1; }
    { std::string o; for (size_t off = 0; off < hdr.size(); off += 0xff00) bgzf_block(hdr.data() + off, std::min<size_t>(0xff00, hdr.size() - off), level, o); fwrite(o.data(), 1, o.size(), fp); }
    // records: parallel over slabs, written in order
    const size_t SLAB = 200000; const size_t n_slab = (recs.size() + SLAB - 1) / SLAB;
    std::vector<std::string> outs(n_slab); std::atomic<size_t> next{0};
    auto work = [&]() {
        std::string rec, payload, nameb;
        for (;;) {
            size_t sl = next.fetch_add(1); if (sl >= n_slab) break;
            std::string& o = outs[sl]; payload.clear();
            size_t i0 = sl * SLAB, i1 = std::min(recs.size(), i0 + SLAB);
            for (size_t i = i0; i < i1; i++) {
                const Rec& R = recs[i];
                Rng m(seed * 0x100000001b3ull + R.mol), r(seed ^ (((uint64_t)R.mol << 8) | R.k) * 0x9E3779B97F4A7C15ull);
                // molecule-level: cell, umi
                double u = m.uni(); int cell = -1; bool outside = false;
                if (u < 0.03) cell = -1; else if (u < 0.05) outside = true; else cell = (int)m.below((uint32_t)bcs.size());
                bool has_ub = m.uni() >= 0.03; char umi[13]; for (int j = 0; j < 12; j++) umi[j] = "ACGT"[m.below(4)]; umi[12] = 0;
                // read-level
                double kd = r.uni(); uint32_t cig[3]; int nc = 1; int a = 10 + (int)r.below(L - 25);
                if (kd < 0.80) { cig[0] = (L << 4) | 0; }
                else if (kd < 0.92) { cig[0] = (a << 4) | 0; cig[1] = ((50 + r.below(19950)) << 4) | 3; cig[2] = ((L - a) << 4) | 0; nc = 3; }
                else if (kd < 0.95) { int x = 1 + (int)r.below(3); cig[0] = (a << 4) | 0; cig[1] = (x << 4) | 1; cig[2] = ((L - a - x) << 4) | 0; nc = 3; }
                else if (kd < 0.98) { cig[0] = (a << 4) | 0; cig[1] = ((1 + r.below(5)) << 4) | 2; cig[2] = ((L - a) << 4) | 0; nc = 3; }
                else { int sc = 1 + (int)r.below(39); if (r.below(2)) { cig[0] = (sc << 4) | 4; cig[1] = ((L - sc) << 4) | 0; } else { cig[0] = ((L - sc) << 4) | 0; cig[1] = (sc << 4) | 4; } nc = 2; }
                int64_t rlen = 0; for (int c = 0; c < nc; c++) { int op = cig[c] & 15; if (op == 0 || op == 2 || op == 3) rlen += cig[c] >> 4; }
                uint8_t mapq = r.uni() < 0.9 ? 255 : (uint8_t)(r.below(3) == 2 ? 3 : r.below(2));
                uint16_t flag = r.below(2) ? 16 : 0; if (r.uni() < 0.03) flag |= 256; if (r.uni() < 0.05) flag |= 1024;
                char qn[24]; int ql = snprintf(qn, sizeof qn, "r%010zu", i) + 1;
                rec.clear();
                put32(rec, (uint32_t)R.tid); put32(rec, (uint32_t)R.pos); rec.push_back((char)ql); rec.push_back((char)mapq);
                put16(rec, (uint16_t)reg2bin(R.pos, R.pos + (rlen ? rlen : 1))); put16(rec, (uint16_t)nc); put16(rec, flag); put32(rec, L);
                put32(rec, (uint32_t)-1); put32(rec, (uint32_t)-1); put32(rec, 0);
                rec.append(qn, ql); rec.append((const char*)cig, nc * 4);
                for (int j = 0; j < (L + 1) / 2; j++) { uint32_t x = (uint32_t)r.next(); rec.push_back((char)(((1u << (x & 3)) << 4) | (1u << ((x >> 2) & 3)))); }
                for (int j = 0; j < L; j++) { static const char q4[4] = {11, 25, 37, 37}; rec.push_back(q4[r.below(4)]); }
                rec.append("NHC", 3); rec.push_back(1);
                if (!no_tags) {                                       // XCK_SYNTH_NOTAGS=1: well-based (SMART-seq) style BAM without CB / UB
                if (cell >= 0) { rec.append("CBZ", 3); rec += bcs[cell]; rec.push_back('\0'); }
                else if (outside) { rec.append("CBZ", 3); for (int j = 0; j < 16; j++) rec.push_back("ACGT"[m.below(4)]); rec.append("-9", 2); rec.push_back('\0'); }
                if (has_ub) { rec.append("UBZ", 3); rec.append(umi, 13); }
                }
                uint32_t bs = (uint32_t)rec.size();
                if (payload.size() + 4 + bs > 0xff00 && !payload.empty()) { bgzf_block(payload.data(), payload.size(), level, o); payload.clear(); }
                payload.append((const char*)&bs, 4); payload += rec;
                while (payload.size() > 0xff00) { bgzf_block(payload.data(), 0xff00, level, o); payload.erase(0, 0xff00); }
            }
            if (!payload.empty()) { bgzf_block(payload.data(), payload.size(), level, o); payload.clear(); }
        }
    };
    std::vector<std::thread> th; for (int t = 0; t < n_thr; t++) th.emplace_back(work); for (auto& t : th) t.join();
    for (auto& o : outs) fwrite(o.data(), 1, o.size(), fp);
    static const unsigned char eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    fwrite(eof, 1, 28, fp); fclose(fp);
    fprintf(stderr, "wrote %zu records to %s\n", recs.size(), out_fn.c_str());
    return 0;
}
