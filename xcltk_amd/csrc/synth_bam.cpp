// synth_bam.cpp - seeded synthetic 10x-style BAM writer (benchmark tooling, not part of the ABI).
//
// Produces a coordinate-sorted BAM (+ .bai) with CB/UB tags whose shape follows SURVEY.md section 8d
// (1-8 reads per UMI, read length 91, CIGAR mix 80/12/3/3/2, MAPQ 255/low, flags, 3 % missing
// CB / UB, 2 % barcodes outside the list), at the size the metric is quoted on (BASELINE.json configs[2]:
// 500 M reads) in well under a minute on 16 cores, with bounded memory:
//   1. per gene, a fixed number of molecules (expression ~ Zipf over a seeded permutation of the genes), every
//      molecule / read drawn from its own counter-based RNG stream -> the records of a contig are generated and
//      sorted independently of the other contigs (parallel over contigs, largest first);
//   2. slabs of 64 k records are encoded and deflated by the worker threads and written in file order through a
//      bounded window (the file never sits in memory);
//   3. the .bai holds, per reference, its virtual-offset range (bin 0, one chunk) and the samtools pseudo-bin with the
//      record counts - what csrc/bam.cpp uses to shard contigs over GPUs; a region query through it is correct but
//      scans the reference.
// The result depends on the arguments only, not on the thread count.  Usage:
//   xck_synth_bam OUT.bam CONTIGS.tsv REGIONS.tsv BARCODES.tsv N_READS SEED [THREADS] [LEVEL]
// LEVEL 1-9: zlib; 0: csrc/deflate_fast.h (greedy LZ77 + dynamic Huffman, ~10x less CPU than zlib -6; streams are
// about 8 % larger).  CONTIGS.tsv: name<TAB>length ; REGIONS.tsv: chrom start end name (chrom must match a contig name
// after stripping "chr") ; BARCODES.tsv: one barcode per line.  XCK_SYNTH_NOTAGS=1: no CB / UB tags (well-based style).
// XCK_SYNTH_SHAPE selects the record shape:
//   (unset)     the SURVEY 8d shape: 12-character read names, tags NH CB UB, ~230 bytes per record
//   cellranger  what a Cell Ranger possorted_genome_bam.bam record looks like: 39-character Illumina read names and 16 aux tags
//               (NH HI AS nM RE xf li RG TX GX GN CR CY CB UR UY UB - CB / UB near the END of the tag list), ~470 bytes per record
//   smartseq    BASELINE configs[4]: paired-end 2 x 75, mates share the read name, flags 99 / 147 / 83 / 163 (3 % of the pairs
//               not "proper": orphans for --countORPHAN), no CB / UB tags (N_READS counts records = 2 x pairs)
#include <zlib.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>
#include "deflate_fast.h"

static inline uint64_t mix(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
struct Rng { uint64_t s; explicit Rng(uint64_t seed) : s(seed) {} uint64_t next() { s = mix(s); return s; } uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); } double uni() { return (next() >> 11) * (1.0 / 9007199254740992.0); } };

struct Rec { int32_t pos; uint32_t mol; uint32_t k; };
struct Gene { int tid; int32_t s, e; };

static std::string strip_chr(const std::string& c) { if (c.size() >= 3 && (c[0] == 'c' || c[0] == 'C') && (c[1] == 'h' || c[1] == 'H') && (c[2] == 'r' || c[2] == 'R')) return c.substr(3); return c; }

typedef std::vector<uint8_t> Bytes;
static inline void put32(Bytes& b, uint32_t v) { const size_t z = b.size(); b.resize(z + 4); memcpy(&b[z], &v, 4); }
static inline void put16(Bytes& b, uint16_t v) { const size_t z = b.size(); b.resize(z + 2); memcpy(&b[z], &v, 2); }
static inline void putn(Bytes& b, const void* p, size_t n) { const size_t z = b.size(); b.resize(z + n); memcpy(&b[z], p, n); }

static int reg2bin(int64_t beg, int64_t end) {
    --end;
    if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
    if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
    if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
    if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
    if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
    return 0;
}

struct Deflater {
    int level; xck::DeflateFast fast; Bytes tmp;
    explicit Deflater(int lv) : level(lv) {}
    void bgzf_block(const uint8_t* p, size_t n, Bytes& out) {
        tmp.clear();
        if (level <= 0) fast.compress(p, n, tmp);
        else {
            z_stream zs; memset(&zs, 0, sizeof zs);
            deflateInit2(&zs, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY);
            tmp.resize(deflateBound(&zs, n) + 64);
            zs.next_in = (Bytef*)p; zs.avail_in = (uInt)n; zs.next_out = tmp.data(); zs.avail_out = (uInt)tmp.size();
            deflate(&zs, Z_FINISH);
            tmp.resize(zs.total_out); deflateEnd(&zs);
        }
        const uint16_t bsize = (uint16_t)(tmp.size() + 25);
        static const unsigned char hdr[12] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0};
        putn(out, hdr, 12); out.push_back('B'); out.push_back('C'); put16(out, 2); put16(out, bsize);
        putn(out, tmp.data(), tmp.size());
        put32(out, (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef*)p, (uInt)n)); put32(out, (uint32_t)n);
    }
};

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: %s OUT.bam CONTIGS.tsv REGIONS.tsv BARCODES.tsv N_READS SEED [THREADS] [LEVEL]\n", argv[0]); return 2; }
    const std::string out_fn = argv[1];
    const int64_t n_reads = atoll(argv[5]); const uint64_t seed = strtoull(argv[6], nullptr, 10);
    int n_thr = argc > 7 ? atoi(argv[7]) : (int)std::thread::hardware_concurrency(); if (n_thr <= 0) n_thr = 4;
    const int level = argc > 8 ? atoi(argv[8]) : 6;
    const char* shape_env = getenv("XCK_SYNTH_SHAPE");
    const bool cr_shape = shape_env && !strcmp(shape_env, "cellranger"), paired = shape_env && !strcmp(shape_env, "smartseq");
    if (shape_env && *shape_env && !cr_shape && !paired) { fprintf(stderr, "unknown XCK_SYNTH_SHAPE '%s'\n", shape_env); return 2; }
    const bool no_tags = paired || (getenv("XCK_SYNTH_NOTAGS") && atoi(getenv("XCK_SYNTH_NOTAGS")));
    std::vector<std::string> cname; std::vector<int32_t> clen;
    { std::ifstream f(argv[2]); std::string a; int64_t l; while (f >> a >> l) { cname.push_back(a); clen.push_back((int32_t)l); } }
    std::vector<Gene> genes;
    { std::ifstream f(argv[3]); std::string line;
      while (std::getline(f, line)) { std::istringstream ss(line); std::string c; int32_t s, e; if (!(ss >> c >> s >> e)) continue;
        std::string sc = strip_chr(c); int tid = -1; for (size_t i = 0; i < cname.size(); i++) if (strip_chr(cname[i]) == sc) { tid = (int)i; break; }
        if (tid >= 0) genes.push_back({tid, s, e}); } }
    std::vector<std::string> bcs;
    { std::ifstream f(argv[4]); std::string b; while (f >> b) bcs.push_back(b); }
    if (cname.empty() || genes.empty() || bcs.empty() || n_reads <= 0) { fprintf(stderr, "empty contigs / regions / barcodes\n"); return 2; }
    const int L = paired ? 75 : 91;
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
    const size_t n_ctg = cname.size();
    // molecules per gene: expression ~ Zipf over a seeded permutation of the genes; 4.5 reads per molecule on average,
    // 0.5 % more molecules than needed (the surplus reads are cut from the end of the last contig)
    std::vector<uint32_t> mol_n(genes.size()), mol_base(genes.size() + 1, 0);
    { std::vector<uint32_t> perm(genes.size()); for (size_t i = 0; i < perm.size(); i++) perm[i] = (uint32_t)i;
      Rng r(seed ^ 0xabcdef); for (size_t i = perm.size(); i > 1; i--) std::swap(perm[i - 1], perm[r.below((uint32_t)i)]);
      std::vector<double> w(genes.size()); double acc = 0;
      for (size_t i = 0; i < perm.size(); i++) { w[perm[i]] = 1.0 / std::pow((double)(i + 1), 0.8); }
      for (double x : w) acc += x;
      const double M = std::ceil((double)n_reads / (paired ? 9.0 : 4.5) * 1.005) + 8;
      double run = 0; uint64_t prev = 0;
      for (size_t g = 0; g < genes.size(); g++) { run += w[g] / acc; const uint64_t cur = (uint64_t)std::floor(M * std::min(run, 1.0) + 1e-9); mol_n[g] = (uint32_t)(cur - prev); prev = cur; mol_base[g + 1] = (uint32_t)cur; } }
    // ---- phase 1: the records of every contig, sorted (parallel over contigs, largest first) ----
    std::vector<std::vector<Rec>> recs(n_ctg);
    { std::vector<std::vector<uint32_t>> genes_of(n_ctg); std::vector<uint64_t> weight(n_ctg, 0);
      for (size_t g = 0; g < genes.size(); g++) { genes_of[genes[g].tid].push_back((uint32_t)g); weight[genes[g].tid] += mol_n[g]; }
      std::vector<int> order(n_ctg); for (size_t i = 0; i < n_ctg; i++) order[i] = (int)i;
      std::sort(order.begin(), order.end(), [&](int a, int b) { return weight[a] != weight[b] ? weight[a] > weight[b] : a < b; });
      std::atomic<size_t> next{0};
      auto work = [&]() {
          for (;;) {
              const size_t oi = next.fetch_add(1); if (oi >= n_ctg) break;
              const int t = order[oi];
              std::vector<Rec>& v = recs[t];
              v.reserve((size_t)(weight[t] * (paired ? 9.2 : 4.6)) + 16);
              for (uint32_t g : genes_of[t]) {
                  const Gene& ge = genes[g];
                  const int64_t span = (int64_t)ge.e - ge.s + 40;
                  for (uint32_t j = 0; j < mol_n[g]; j++) {
                      const uint32_t mol = mol_base[g] + j;
                      Rng r(mix(seed * 0x2545F4914F6CDD1Dull + mol));
                      const int64_t anchor = (int64_t)ge.s - 41 + (int64_t)(r.uni() * span);
                      const uint32_t k = 1 + r.below(8);
                      for (uint32_t q = 0; q < k; q++) {
                          int64_t pos = anchor + r.below(200); if (pos < 0) pos = 0;
                          if (pos > clen[t] - L - 25000) pos = std::max<int64_t>(0, clen[t] - L - 25000);
                          if (!paired) { v.push_back({(int32_t)pos, mol, q}); continue; }
                          int64_t mpos = pos + 60 + r.below(300);                     // the mate: same read name, its own record
                          if (mpos > clen[t] - L - 25000) mpos = std::max<int64_t>(0, clen[t] - L - 25000);
                          v.push_back({(int32_t)pos, mol, 2 * q}); v.push_back({(int32_t)mpos, mol, 2 * q + 1});
                      }
                  }
              }
              std::sort(v.begin(), v.end(), [](const Rec& a, const Rec& b) { return a.pos != b.pos ? a.pos < b.pos : (a.mol != b.mol ? a.mol < b.mol : a.k < b.k); });
          }
      };
      std::vector<std::thread> th; for (int t = 0; t < n_thr; t++) th.emplace_back(work); for (auto& t : th) t.join(); }
    { int64_t tot = 0; for (auto& v : recs) tot += (int64_t)v.size();
      for (size_t t = n_ctg; t-- > 0 && tot > n_reads;) { const int64_t cut = std::min<int64_t>(tot - n_reads, (int64_t)recs[t].size()); recs[t].resize(recs[t].size() - (size_t)cut); tot -= cut; } }
    const double t_phase1 = since();
    std::vector<uint64_t> rec_base(n_ctg + 1, 0);
    for (size_t t = 0; t < n_ctg; t++) rec_base[t + 1] = rec_base[t] + recs[t].size();
    // ---- header ----
    Bytes hdr;
    { std::string text = "@HD\tVN:1.6\tSO:coordinate\n"; for (size_t i = 0; i < n_ctg; i++) text += "@SQ\tSN:" + cname[i] + "\tLN:" + std::to_string(clen[i]) + "\n";
      putn(hdr, "BAM\1", 4); put32(hdr, (uint32_t)text.size()); putn(hdr, text.data(), text.size()); put32(hdr, (uint32_t)n_ctg);
      for (size_t i = 0; i < n_ctg; i++) { put32(hdr, (uint32_t)cname[i].size() + 1); putn(hdr, cname[i].c_str(), cname[i].size() + 1); put32(hdr, (uint32_t)clen[i]); } }
    FILE* fp = fopen(out_fn.c_str(), "wb"); if (!fp) { perror("open"); return 1; }
    uint64_t fpos = 0;
    { Deflater df(level); Bytes o; for (size_t off = 0; off < hdr.size(); off += 0xff00) df.bgzf_block(hdr.data() + off, std::min<size_t>(0xff00, hdr.size() - off), o);
      fwrite(o.data(), 1, o.size(), fp); fpos += o.size(); }
    // ---- phase 2: slabs (never across a contig boundary) encoded in parallel, written in order ----
    struct Slab { int tid; size_t i0, i1; };
    std::vector<Slab> slabs;
    const size_t SLAB = 65536;
    for (size_t t = 0; t < n_ctg; t++) for (size_t i = 0; i < recs[t].size(); i += SLAB) slabs.push_back({(int)t, i, std::min(recs[t].size(), i + SLAB)});
    const size_t n_slab = slabs.size();
    const size_t RING = (size_t)n_thr * 4 + 4;
    std::vector<Bytes> outs(RING); std::vector<char> ready(RING, 0);
    std::mutex mu; std::condition_variable cv_ready, cv_space; size_t written = 0;
    std::atomic<size_t> next{0};
    auto work = [&]() {
        Deflater df(level); Bytes rec, payload, o;
        for (;;) {
            const size_t sl = next.fetch_add(1); if (sl >= n_slab) break;
            { std::unique_lock<std::mutex> lk(mu); cv_space.wait(lk, [&] { return sl < written + RING; }); }
            const Slab& S = slabs[sl]; const std::vector<Rec>& v = recs[S.tid];
            o.clear(); payload.clear();
            for (size_t i = S.i0; i < S.i1; i++) {
                const Rec& R = v[i];
                Rng m(seed * 0x100000001b3ull + R.mol), r(seed ^ (((uint64_t)R.mol << 8) | R.k) * 0x9E3779B97F4A7C15ull);
                // molecule-level: cell, umi
                const double u = m.uni(); int cell = -1; bool outside = false;
                if (u < 0.03) cell = -1; else if (u < 0.05) outside = true; else cell = (int)m.below((uint32_t)bcs.size());
                const bool has_ub = m.uni() >= 0.03; char umi[13]; { uint64_t x = m.next(); for (int j = 0; j < 12; j++) { umi[j] = "ACGT"[x & 3]; x >>= 2; } } umi[12] = 0;
                // read-level
                const double kd = r.uni(); uint32_t cig[3]; int nc = 1; const int a = 10 + (int)r.below(L - 25);
                if (kd < 0.80) { cig[0] = (L << 4) | 0; }
                else if (kd < 0.92) { cig[0] = (a << 4) | 0; cig[1] = ((50 + r.below(19950)) << 4) | 3; cig[2] = ((L - a) << 4) | 0; nc = 3; }
                else if (kd < 0.95) { const int x = 1 + (int)r.below(3); cig[0] = (a << 4) | 0; cig[1] = (x << 4) | 1; cig[2] = ((L - a - x) << 4) | 0; nc = 3; }
                else if (kd < 0.98) { cig[0] = (a << 4) | 0; cig[1] = ((1 + r.below(5)) << 4) | 2; cig[2] = ((L - a) << 4) | 0; nc = 3; }
                else { const int sc = 1 + (int)r.below(39); if (r.below(2)) { cig[0] = (sc << 4) | 4; cig[1] = ((L - sc) << 4) | 0; } else { cig[0] = ((L - sc) << 4) | 0; cig[1] = (sc << 4) | 4; } nc = 2; }
                int64_t rlen = 0; for (int c = 0; c < nc; c++) { const int op = cig[c] & 15; if (op == 0 || op == 2 || op == 3) rlen += cig[c] >> 4; }
                const uint8_t mapq = r.uni() < 0.9 ? 255 : (uint8_t)(r.below(3) == 2 ? 3 : r.below(2));
                uint16_t flag = r.below(2) ? 16 : 0; if (r.uni() < 0.03) flag |= 256; if (r.uni() < 0.05) flag |= 1024;
                char qn[48]; int ql;                                        // "r%010llu" + NUL
                { unsigned long long v = (unsigned long long)(rec_base[S.tid] + i); int nd = 10; for (unsigned long long t = 10000000000ull; v >= t; t *= 10) nd++;
                  qn[0] = 'r'; for (int j = nd; j >= 1; j--) { qn[j] = (char)('0' + v % 10); v /= 10; } qn[nd + 1] = 0; ql = nd + 2; }
                if (cr_shape) {                                             // instrument:run:flowcell:lane:tile:x:y - 39 characters
                    const unsigned long long v = (unsigned long long)(rec_base[S.tid] + i);
                    ql = 1 + snprintf(qn, sizeof qn, "A00228:279:HFWFVDMXX:%u:%04u:%05u:%05u", 1 + (unsigned)(v % 4), 1101 + (unsigned)((v / 40000000000ull) % 8899),
                                      (unsigned)((v / 4) % 100000), (unsigned)((v / 400000) % 100000));
                }
                if (paired) {                                               // the pair's name (both mates), pair-level strand / properness
                    Rng pr(seed * 0x9E3779B1ull + (((uint64_t)R.mol << 4) | (R.k >> 1)));
                    const bool rev = pr.below(2) != 0, proper = pr.uni() >= 0.03, first = (R.k & 1) == 0;
                    flag = (uint16_t)(1 | (proper ? 2 : 0) | (first ? 64 : 128) | ((first ? rev : !rev) ? 16 : 32));
                    if (pr.uni() < 0.03) flag |= 256;
                    ql = 1 + snprintf(qn, sizeof qn, "SRR.%u.%u", (unsigned)R.mol, (unsigned)(R.k >> 1));
                }
                rec.clear();
                put32(rec, (uint32_t)S.tid); put32(rec, (uint32_t)R.pos); rec.push_back((uint8_t)ql); rec.push_back(mapq);
                put16(rec, (uint16_t)reg2bin(R.pos, R.pos + (rlen ? rlen : 1))); put16(rec, (uint16_t)nc); put16(rec, flag); put32(rec, L);
                if (paired) { put32(rec, (uint32_t)S.tid); put32(rec, (uint32_t)R.pos); put32(rec, 0); }   // (mate fields: same reference; xcltk reads the flags only)
                else { put32(rec, (uint32_t)-1); put32(rec, (uint32_t)-1); put32(rec, 0); }
                putn(rec, qn, ql); putn(rec, cig, nc * 4);
                { uint8_t sq[(91 + 1) / 2]; uint64_t x = 0; int left = 0;                // two random bases per byte (4 bits of entropy)
                  for (int j = 0; j < (L + 1) / 2; j++) { if (!left) { x = r.next(); left = 16; } sq[j] = (uint8_t)(((1u << (x & 3)) << 4) | (1u << ((x >> 2) & 3))); x >>= 4; left--; }
                  putn(rec, sq, (L + 1) / 2); }
                { uint8_t ql_[91]; uint64_t x = 0; int left = 0; static const uint8_t q4[4] = {11, 25, 37, 37};
                  for (int j = 0; j < L; j++) { if (!left) { x = r.next(); left = 32; } ql_[j] = q4[x & 3]; x >>= 2; left--; }
                  putn(rec, ql_, L); }
                putn(rec, "NHC", 3); rec.push_back(1);
                if (cr_shape) {                                             // the tags Cell Ranger writes before the barcode / UMI tags
                    const uint32_t gene = (uint32_t)(std::upper_bound(mol_base.begin(), mol_base.end(), R.mol) - mol_base.begin()) - 1;
                    char tx[64];
                    putn(rec, "HIC", 3); rec.push_back(1); putn(rec, "ASC", 3); rec.push_back((uint8_t)(L - 2)); putn(rec, "nMC", 3); rec.push_back((uint8_t)r.below(3));
                    putn(rec, "REA", 3); rec.push_back('E'); putn(rec, "xfC", 3); rec.push_back(25); putn(rec, "liC", 3); rec.push_back(0);
                    putn(rec, "RGZ", 3); putn(rec, "synth:0:1:HFWFVDMXX:1", 22);
                    putn(rec, "TXZ", 3); putn(rec, tx, 1 + (size_t)snprintf(tx, sizeof tx, "ENST%011u,+%u,%dM", gene * 7 + 3, (unsigned)r.below(3000), L));
                    putn(rec, "GXZ", 3); putn(rec, tx, 1 + (size_t)snprintf(tx, sizeof tx, "ENSG%011u", gene + 100000));
                    putn(rec, "GNZ", 3); putn(rec, tx, 1 + (size_t)snprintf(tx, sizeof tx, "GENE%05u", gene));
                    char raw[17]; if (cell >= 0) memcpy(raw, bcs[cell].c_str(), 16); else { uint64_t x = mix(R.mol); for (int j = 0; j < 16; j++) { raw[j] = "ACGT"[x & 3]; x >>= 2; } } raw[16] = 0;
                    char q16[17]; { uint64_t x = r.next(); for (int j = 0; j < 16; j++) { q16[j] = (x & 7) ? 'F' : ','; x >>= 3; } q16[16] = 0; }
                    putn(rec, "CRZ", 3); putn(rec, raw, 17); putn(rec, "CYZ", 3); putn(rec, q16, 17);
                }
                if (!no_tags) {                                       // XCK_SYNTH_NOTAGS=1: well-based (SMART-seq) style BAM without CB / UB
                    if (cell >= 0) { putn(rec, "CBZ", 3); putn(rec, bcs[cell].c_str(), bcs[cell].size() + 1); }
                    else if (outside) { putn(rec, "CBZ", 3); uint64_t x = m.next(); for (int j = 0; j < 16; j++) { rec.push_back((uint8_t)"ACGT"[x & 3]); x >>= 2; } putn(rec, "-9", 3); }
                    if (cr_shape) { char q12[13]; uint64_t x = r.next(); for (int j = 0; j < 12; j++) { q12[j] = (x & 7) ? 'F' : ':'; x >>= 3; } q12[12] = 0;
                                    putn(rec, "URZ", 3); putn(rec, umi, 13); putn(rec, "UYZ", 3); putn(rec, q12, 13); }
                    if (has_ub) { putn(rec, "UBZ", 3); putn(rec, umi, 13); }
                }
                const uint32_t bs = (uint32_t)rec.size();
                if (payload.size() + 4 + bs > 0xff00 && !payload.empty()) { df.bgzf_block(payload.data(), payload.size(), o); payload.clear(); }   // like htslib: a record does not straddle blocks
                put32(payload, bs); putn(payload, rec.data(), rec.size());
            }
            if (!payload.empty()) { df.bgzf_block(payload.data(), payload.size(), o); payload.clear(); }
            { std::lock_guard<std::mutex> lk(mu); outs[sl % RING].swap(o); ready[sl % RING] = 1; }
            cv_ready.notify_all();
        }
    };
    std::vector<uint64_t> ref_beg(n_ctg, 0), ref_end(n_ctg, 0);
    std::vector<std::thread> th; for (int t = 0; t < n_thr; t++) th.emplace_back(work);
    bool io_ok = true;
    for (size_t sl = 0; sl < n_slab; sl++) {
        Bytes o;
        { std::unique_lock<std::mutex> lk(mu); cv_ready.wait(lk, [&] { return ready[sl % RING] != 0; }); o.swap(outs[sl % RING]); ready[sl % RING] = 0; }
        if (slabs[sl].i0 == 0) ref_beg[slabs[sl].tid] = fpos;
        if (fwrite(o.data(), 1, o.size(), fp) != o.size()) io_ok = false;
        fpos += o.size(); ref_end[slabs[sl].tid] = fpos;
        { std::lock_guard<std::mutex> lk(mu); written = sl + 1; }
        cv_space.notify_all();
    }
    for (auto& t : th) t.join();
    static const unsigned char eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    fwrite(eof, 1, 28, fp);
    if (fclose(fp) != 0 || !io_ok) { fprintf(stderr, "write error on %s\n", out_fn.c_str()); return 1; }
    // ---- .bai ----
    { Bytes b; putn(b, "BAI\1", 4); put32(b, (uint32_t)n_ctg);
      auto put64 = [&](uint64_t v) { putn(b, &v, 8); };
      for (size_t t = 0; t < n_ctg; t++) {
          if (recs[t].empty()) { put32(b, 0); put32(b, 0); continue; }
          put32(b, 2);
          put32(b, 0); put32(b, 1); put64(ref_beg[t] << 16); put64(ref_end[t] << 16);                       // bin 0: the whole reference
          put32(b, 37450); put32(b, 2); put64(ref_beg[t] << 16); put64(ref_end[t] << 16); put64(recs[t].size()); put64(0);
          put32(b, 0);                                                                                       // no linear index
      }
      put64(0);                                                                                              // n_no_coor
      FILE* fi = fopen((out_fn + ".bai").c_str(), "wb"); if (!fi) { perror("open .bai"); return 1; }
      fwrite(b.data(), 1, b.size(), fi); fclose(fi); }
    fprintf(stderr, "wrote %llu records to %s (records drawn and sorted in %.1f s, encoded and written in %.1f s, %d threads, level %d)\n",
            (unsigned long long)rec_base[n_ctg], out_fn.c_str(), t_phase1, since() - t_phase1, n_thr, level);
    return 0;
}
