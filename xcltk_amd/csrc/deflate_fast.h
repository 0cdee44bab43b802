// deflate_fast.h - whole-buffer RFC 1951 encoder for BGZF blocks (<= 64 KiB of input per call).
//
// Used where this repo WRITES BGZF: the synthetic-BAM generator (csrc/synth_bam.cpp; zlib level 6 needs 17 us of
// CPU per record, which would make a 500 M-read benchmark input take minutes to produce) and the bgzip text writer
// behind utils/zfile.py (reference xcltk/utils/zfile.py:14-133 writes its .gz outputs through pysam.BGZFile).
// One dynamic-Huffman block per call: greedy LZ77 with a single-probe hash of 4-byte strings (matches >= 4 bytes,
// distance < 32 KiB, stride grows inside runs without matches so incompressible spans - packed bases, qualities -
// cost ~1 ns per byte), canonical Huffman codes limited to 15 / 7 bits, code lengths sent with the 16 / 17 / 18
// run-length symbols.  Any RFC 1951 inflater reads the result (checked against zlib and csrc/inflate_fast.h in
// tests/test_host_logic.py).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

namespace xck {

class DeflateFast {
public:
    // appends the raw deflate stream of src[0..n) (n <= 65535) to out; returns bytes appended
    size_t compress(const uint8_t* src, size_t n, std::vector<uint8_t>& out) {
        parse(src, n);
        const size_t z0 = out.size();
        const size_t c = emit(src, n, out);
        if (c <= n + 5) return c;
        out.resize(z0);                                                          // incompressible: one stored block
        out.push_back(1); out.push_back((uint8_t)(n & 0xff)); out.push_back((uint8_t)(n >> 8));
        out.push_back((uint8_t)(~n & 0xff)); out.push_back((uint8_t)((~n >> 8) & 0xff));
        out.insert(out.end(), src, src + n);
        return n + 5;
    }

private:
    static constexpr int HASH_BITS = 15;
    static constexpr size_t MIN_MATCH = 6;
    int32_t head_[1 << HASH_BITS];
    struct Tok { uint16_t len; uint16_t dist; uint32_t lit_run; };   // lit_run literals, then a match (len 0: tail literals only)
    std::vector<Tok> toks_;
    uint32_t fll_[288], fd_[32];
    uint8_t  lll_[288], ld_[32];
    uint16_t cll_[288], cd_[32];

    static inline uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
    static inline uint64_t ld64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
    struct SymTab {                                                            // length / distance -> (symbol, extra bits, extra value)
        uint8_t len_sym[259], len_eb[259]; uint16_t len_base[259];
        uint8_t d_sym[512], d_eb[30]; uint16_t d_base[30];
        SymTab() {
            static const uint16_t lb[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
            static const uint8_t  le[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
            for (int l = 3; l <= 258; l++) { int c = 28; while (lb[c] > l) c--; len_sym[l] = (uint8_t)c; len_eb[l] = le[c]; len_base[l] = lb[c]; }
            static const uint16_t db[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
            static const uint8_t  de[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
            for (int c = 0; c < 30; c++) { d_eb[c] = de[c]; d_base[c] = db[c]; }
            for (int d = 1; d <= 256; d++) { int c = 29; while (db[c] > d) c--; d_sym[d - 1] = (uint8_t)c; }              // d_sym[d - 1] for d <= 256
            for (int q = 2; q < 256; q++) { const int d = (q << 7) + 1; int c = 29; while (db[c] > d) c--; d_sym[256 + q] = (uint8_t)c; }   // d_sym[256 + ((d - 1) >> 7)] beyond
        }
    };
    static const SymTab& tab() { static const SymTab t; return t; }
    static inline int len_code(int len, int* ebits, int* eval) {               // len 3..258 -> symbol 257..285
        const SymTab& t = tab(); *ebits = t.len_eb[len]; *eval = len - t.len_base[len]; return 257 + t.len_sym[len];
    }
    static inline int dist_code(int d, int* ebits, int* eval) {                // d 1..32768 -> symbol 0..29
        const SymTab& t = tab(); const int c = d <= 256 ? t.d_sym[d - 1] : t.d_sym[256 + ((d - 1) >> 7)];
        *ebits = t.d_eb[c]; *eval = d - t.d_base[c]; return c;
    }

    void parse(const uint8_t* s, size_t n) {
        toks_.clear();
        memset(fll_, 0, sizeof fll_); memset(fd_, 0, sizeof fd_);
        memset(head_, 0xff, sizeof head_);
        size_t i = 0, lit0 = 0; uint32_t miss = 0;
        const size_t lim = n >= 8 ? n - 8 : 0;                                   // 8-byte loads stay inside the buffer
        while (i < lim) {
            const uint32_t w = ld32(s + i);
            const uint32_t h = (w * 2654435761u) >> (32 - HASH_BITS);
            const int32_t c = head_[h];
            head_[h] = (int32_t)i;
            if (c >= 0 && i - (size_t)c <= 32768 && ld32(s + c) == w) {
                size_t len = 4; const size_t mx = std::min<size_t>(258, n - i);
                while (len + 8 <= mx) { const uint64_t x = ld64(s + c + len) ^ ld64(s + i + len); if (x) { len += (size_t)(__builtin_ctzll(x) >> 3); goto done; } len += 8; }
                while (len < mx && s[c + len] == s[i + len]) len++;
            done:
                if (len > mx) len = mx;
                if (len < MIN_MATCH) { i += 1 + (miss++ >> 3); continue; }            // a short match costs more bits than its literals in low-entropy spans
                toks_.push_back({(uint16_t)len, (uint16_t)(i - (size_t)c), (uint32_t)(i - lit0)});
                { int eb, ev; fll_[len_code((int)len, &eb, &ev)]++; fd_[dist_code((int)(i - (size_t)c), &eb, &ev)]++; }
                if (i + len < lim) { const size_t j = i + len - 2; head_[(ld32(s + j) * 2654435761u) >> (32 - HASH_BITS)] = (int32_t)j; }
                i += len; lit0 = i; miss = 0;
            } else {
                i += 1 + (miss++ >> 3);
            }
        }
        toks_.push_back({0, 0, (uint32_t)(n - lit0)});
        // literal frequencies: every byte not covered by a match
        size_t p = 0;
        for (const Tok& t : toks_) { for (uint32_t k = 0; k < t.lit_run; k++) fll_[s[p + k]]++; p += t.lit_run + t.len; }
        fll_[256] = 1;
    }

    // code lengths (<= limit) for n symbols from their frequencies; unused symbols get 0
    static void huff_lengths(const uint32_t* f, int n, int limit, uint8_t* out) {
        std::vector<uint32_t> fr(f, f + n);
        for (;;) {
            struct Node { uint64_t w; int l, r; };
            std::vector<Node> nd; std::vector<int> leaf;
            for (int i = 0; i < n; i++) if (fr[i]) { leaf.push_back((int)nd.size()); nd.push_back({fr[i], -1, i}); }
            memset(out, 0, (size_t)n);
            if (leaf.empty()) return;
            if (leaf.size() == 1) { out[nd[0].r] = 1; return; }
            std::sort(leaf.begin(), leaf.end(), [&](int a, int b) { return nd[a].w != nd[b].w ? nd[a].w < nd[b].w : nd[a].r < nd[b].r; });
            std::vector<int> q2; size_t a = 0, b = 0;                            // two-queue Huffman construction
            auto pop = [&]() { int x; if (b >= q2.size() || (a < leaf.size() && nd[leaf[a]].w <= nd[q2[b]].w)) x = leaf[a++]; else x = q2[b++]; return x; };
            const size_t n_leaf = leaf.size();
            for (size_t k = 0; k + 1 < n_leaf; k++) { const int x = pop(), y = pop(); nd.push_back({nd[x].w + nd[y].w, x, y}); q2.push_back((int)nd.size() - 1); }
            // depths: internal nodes were appended after their children, so walk from the root down
            std::vector<int> depth(nd.size(), 0); int mx = 0;
            for (int k = (int)nd.size() - 1; k >= 0; k--) if (nd[k].l >= 0) { depth[nd[k].l] = depth[k] + 1; depth[nd[k].r] = depth[k] + 1; }
            for (size_t k = 0; k < n_leaf; k++) mx = std::max(mx, depth[k]);
            if (mx <= limit) { for (size_t k = 0; k < n_leaf; k++) out[nd[k].r] = (uint8_t)depth[k]; return; }
            for (auto& x : fr) if (x) x = (x + 1) >> 1;                           // too deep: flatten the distribution and rebuild
        }
    }
    static void canon_codes(const uint8_t* len, int n, uint16_t* code) {        // bit-reversed (deflate packs codes MSB first)
        int cnt[16] = {0}, next[16];
        for (int i = 0; i < n; i++) cnt[len[i]]++;
        cnt[0] = 0; int c = 0;
        for (int b = 1; b < 16; b++) { c = (c + cnt[b - 1]) << 1; next[b] = c; }
        for (int i = 0; i < n; i++) {
            const int l = len[i]; if (!l) { code[i] = 0; continue; }
            int v = next[l]++, r = 0;
            for (int k = 0; k < l; k++) { r = (r << 1) | (v & 1); v >>= 1; }
            code[i] = (uint16_t)r;
        }
    }

    // bit writer over a pre-sized buffer (the caller reserves the worst case)
    struct BitW {
        uint8_t* p; uint64_t acc = 0; int nb = 0;
        explicit BitW(uint8_t* dst) : p(dst) {}
        inline void put(uint32_t v, int n) { acc |= (uint64_t)v << nb; nb += n; if (nb >= 32) { const uint32_t w = (uint32_t)acc; memcpy(p, &w, 4); p += 4; acc >>= 32; nb -= 32; } }
        uint8_t* finish() { while (nb > 0) { *p++ = (uint8_t)acc; acc >>= 8; nb -= 8; } nb = 0; acc = 0; return p; }
    };

    size_t emit(const uint8_t* s, size_t n, std::vector<uint8_t>& out) {
        const size_t z0 = out.size();
        huff_lengths(fll_, 286, 15, lll_);
        bool any_d = false; for (int i = 0; i < 30; i++) any_d |= fd_[i] != 0;
        if (!any_d) { fd_[0] = 1; fd_[1] = 1; }                                  // a complete (two-code) distance tree that is never used
        huff_lengths(fd_, 30, 15, ld_);
        if (!any_d) { ld_[0] = 1; ld_[1] = 1; }
        canon_codes(lll_, 286, cll_); canon_codes(ld_, 30, cd_);
        int hlit = 286; while (hlit > 257 && lll_[hlit - 1] == 0) hlit--;
        int hdist = 30; while (hdist > 1 && ld_[hdist - 1] == 0) hdist--;
        // run-length code the hlit + hdist lengths
        uint8_t seq[320]; int ns = 0;
        for (int i = 0; i < hlit; i++) seq[ns++] = lll_[i];
        for (int i = 0; i < hdist; i++) seq[ns++] = ld_[i];
        struct CL { uint8_t sym, ebits, eval; };
        std::vector<CL> cl; uint32_t fcl[19] = {0};
        for (int i = 0; i < ns;) {
            int j = i; while (j < ns && seq[j] == seq[i]) j++;
            int run = j - i; const uint8_t v = seq[i];
            if (v == 0) {
                while (run >= 11) { const int r = std::min(run, 138); cl.push_back({18, 7, (uint8_t)(r - 11)}); fcl[18]++; run -= r; }
                if (run >= 3) { cl.push_back({17, 3, (uint8_t)(run - 3)}); fcl[17]++; run = 0; }
                while (run-- > 0) { cl.push_back({0, 0, 0}); fcl[0]++; }
            } else {
                cl.push_back({v, 0, 0}); fcl[v]++; run--;
                while (run >= 3) { const int r = std::min(run, 6); cl.push_back({16, 2, (uint8_t)(r - 3)}); fcl[16]++; run -= r; }
                while (run-- > 0) { cl.push_back({v, 0, 0}); fcl[v]++; }
            }
            i = j;
        }
        uint8_t lcl[19]; uint16_t ccl[19];
        huff_lengths(fcl, 19, 7, lcl); canon_codes(lcl, 19, ccl);
        static const uint8_t ord[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        int hclen = 19; while (hclen > 4 && lcl[ord[hclen - 1]] == 0) hclen--;
        out.resize(z0 + n * 2 + 1024);                                          // <= 15 bits per literal, 48 per match (>= 4 bytes), header < 400 bytes
        BitW bw(out.data() + z0);
        bw.put(1, 1); bw.put(2, 2);                                              // BFINAL, BTYPE = dynamic
        bw.put((uint32_t)(hlit - 257), 5); bw.put((uint32_t)(hdist - 1), 5); bw.put((uint32_t)(hclen - 4), 4);
        for (int i = 0; i < hclen; i++) bw.put(lcl[ord[i]], 3);
        for (const CL& c : cl) { bw.put(ccl[c.sym], lcl[c.sym]); if (c.ebits) bw.put(c.eval, c.ebits); }
        size_t p = 0;
        for (const Tok& t : toks_) {
            for (uint32_t k = 0; k < t.lit_run; k++) { const uint8_t b = s[p + k]; bw.put(cll_[b], lll_[b]); }
            p += t.lit_run;
            if (t.len) {
                int eb, ev; const int ls = len_code(t.len, &eb, &ev);
                bw.put(cll_[ls], lll_[ls]); if (eb) bw.put((uint32_t)ev, eb);
                const int ds = dist_code(t.dist, &eb, &ev);
                bw.put(cd_[ds], ld_[ds]); if (eb) bw.put((uint32_t)ev, eb);
                p += t.len;
            }
        }
        bw.put(cll_[256], lll_[256]);
        out.resize((size_t)(bw.finish() - out.data()));
        return out.size() - z0;
    }
};

}  // namespace xck
