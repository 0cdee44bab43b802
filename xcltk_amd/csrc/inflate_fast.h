// inflate_fast.h - whole-buffer raw-DEFLATE decoder for BGZF blocks (RFC 1951), written for this decoder.
//
// zlib 1.2.11's streaming inflate() is the host-side bottleneck of the ingest (~80 % of decode time).
// BGZF gives us what a streaming decoder cannot assume: the whole compressed block is in memory and the
// exact output size is known.  That allows a 64-bit bit buffer refilled with one unaligned load, 11-bit /
// 8-bit first-level decode tables with second-level subtables, and 8-byte match copies - while never writing
// one byte past `out + out_len` (neighbouring blocks of the chunk are being written by other threads).
// Any irregularity returns a negative code and the caller falls back to zlib for that block.
#pragma once
#include <cstdint>
#include <cstring>

#ifndef XCK_INFLATE_MATCH_HOOK
#define XCK_INFLATE_MATCH_HOOK(offset, length)       // tools/inflate_stats.cpp counts the matches of a file through this; nothing in the product
#endif

namespace xck {

#ifdef XCK_INFLATE_PROF
#include <x86intrin.h>
static unsigned long long g_prof_build = 0, g_prof_nblocks = 0;
#endif

struct HuffEnt { uint32_t val; uint8_t len; uint8_t op; uint16_t pad; };
// op: 0x00 literal (val = byte) | 0x01..0x03 TWO..FOUR literals (val = the bytes in output order, len = all code lengths)
//     0x10+x length/distance base in val with x extra bits | 0x20 end of block
//     0x40 invalid code | 0x80+s link to a subtable at index val, indexed by the next s bits (len = primary bits)

constexpr int LIT_TB = 11, DIST_TB = 8, PRE_TB = 7;
constexpr int LIT_TABLE_MAX = (1 << LIT_TB) + 288 * 16;  // primary + worst case: every symbol a 15-bit code in its own subtable
constexpr int DIST_TABLE_MAX = (1 << DIST_TB) + 30 * 128;

struct InflateTables {
    HuffEnt lit[LIT_TABLE_MAX];
    HuffEnt dist[DIST_TABLE_MAX];
};

static const uint16_t kLenBase[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
static const uint8_t  kLenExtra[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const uint16_t kDistBase[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
static const uint8_t  kDistExtra[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};

static inline uint32_t bitrev(uint32_t v, int n) {                 // reverse the low n (<= 15) bits
    v = ((v & 0x5555u) << 1) | ((v >> 1) & 0x5555u); v = ((v & 0x3333u) << 2) | ((v >> 2) & 0x3333u);
    v = ((v & 0x0f0fu) << 4) | ((v >> 4) & 0x0f0fu); v = ((v & 0x00ffu) << 8) | ((v >> 8) & 0x00ffu);
    return v >> (16 - n);
}

// Build a two-level decode table from code lengths. kind: 0 literal/length alphabet, 1 distance alphabet,
// 2 code-length alphabet (plain symbols in val).  Returns false on an over-subscribed / unusable code.
static inline bool build_table(const uint8_t* lens, int n_sym, int tb, HuffEnt* tab, int tab_max, int kind) {
    int count[16] = {0};
    for (int i = 0; i < n_sym; i++) count[lens[i]]++;
    count[0] = 0;
    int left = 1;
    for (int l = 1; l <= 15; l++) { left = left * 2 - count[l]; if (left < 0) return false; }
    uint32_t next[16]; uint32_t code = 0;
    for (int l = 1; l <= 15; l++) { code = (code + (uint32_t)count[l - 1]) << 1; next[l] = code; }
    const int psize = 1 << tb;
    HuffEnt inval; inval.val = 0; inval.len = 1; inval.op = 0x40;
    for (int i = 0; i < psize; i++) tab[i] = inval;
    // first pass: how many extra bits does every long-code prefix need
    uint8_t sub_bits[1 << LIT_TB];
    bool any_long = false;
    for (int l = tb + 1; l <= 15; l++) if (count[l]) any_long = true;
    if (any_long) {
        memset(sub_bits, 0, (size_t)psize);
        uint32_t nx[16]; memcpy(nx, next, sizeof nx);
        for (int s = 0; s < n_sym; s++) { int l = lens[s]; if (l <= tb) { if (l) nx[l]++; continue; }
            uint32_t r = bitrev(nx[l]++, l); uint32_t p = r & (uint32_t)(psize - 1); if (l - tb > sub_bits[p]) sub_bits[p] = (uint8_t)(l - tb); }
    }
    int used = psize;
    uint16_t sub_off[1 << LIT_TB];
    if (any_long) for (int p = 0; p < psize; p++) if (sub_bits[p]) {
        if (used + (1 << sub_bits[p]) > tab_max) return false;
        sub_off[p] = (uint16_t)used;
        tab[p].val = (uint16_t)used; tab[p].len = (uint8_t)tb; tab[p].op = (uint8_t)(0x80 | sub_bits[p]);
        for (int k = 0; k < (1 << sub_bits[p]); k++) tab[used + k] = inval;
        used += 1 << sub_bits[p];
    }
    for (int s = 0; s < n_sym; s++) {
        int l = lens[s]; if (!l) continue;
        HuffEnt e; e.len = (uint8_t)l;
        if (kind == 0) {
            if (s < 256) { e.val = (uint16_t)s; e.op = 0; }
            else if (s == 256) { e.val = 0; e.op = 0x20; }
            else if (s <= 285) { e.val = kLenBase[s - 257]; e.op = (uint8_t)(0x10 | kLenExtra[s - 257]); }
            else { e.val = 0; e.op = 0x40; }
        } else if (kind == 1) {
            if (s < 30) { e.val = kDistBase[s]; e.op = (uint8_t)(0x10 | kDistExtra[s]); } else { e.val = 0; e.op = 0x40; }
        } else { e.val = (uint16_t)s; e.op = 0; }
        uint32_t r = bitrev(next[l]++, l);
        if (l <= tb) { for (uint32_t k = r; k < (uint32_t)psize; k += 1u << l) tab[k] = e; }
        else {
            uint32_t p = r & (uint32_t)(psize - 1); int sb = sub_bits[p];
            e.len = (uint8_t)(l - tb);                               // bits consumed inside the subtable
            for (uint32_t k = r >> tb; k < (1u << sb); k += 1u << (l - tb)) tab[sub_off[p] + k] = e;
        }
    }
    return true;
}

// Literal-heavy streams (BAM qualities / packed bases) are bound by the serial table-lookup chain, one
// literal per lookup.  Wherever several literal codes fit together in the 11 index bits, fuse them: the entry
// then yields up to four bytes per lookup (binned base qualities have 1-3 bit codes, two packed bases ~5 bits).
// Only primary-table single-literal entries are fused.
static inline void pair_literals(HuffEnt* tab) {
    static thread_local HuffEnt orig[1 << LIT_TB];
    memcpy(orig, tab, sizeof orig);
    for (uint32_t i = 0; i < (1u << LIT_TB); i++) {
        const HuffEnt a = orig[i];
        if (a.op != 0 || a.len >= LIT_TB) continue;
        uint32_t val = a.val; int total = a.len, n = 1;
        while (n < 4) {
            const HuffEnt b = orig[i >> total];                   // upper (unknown) bits read as 0: valid iff b fits the known bits
            if (b.op != 0 || total + b.len > LIT_TB) break;
            val |= b.val << (8 * n); total += b.len; n++;
        }
        tab[i].val = val; tab[i].len = (uint8_t)total; tab[i].op = (uint8_t)(n - 1);
    }
}

struct FixedTables { InflateTables t; bool ok; FixedTables() { uint8_t l[288]; for (int i = 0; i < 144; i++) l[i] = 8; for (int i = 144; i < 256; i++) l[i] = 9;
    for (int i = 256; i < 280; i++) l[i] = 7; for (int i = 280; i < 288; i++) l[i] = 8; uint8_t d[30]; for (int i = 0; i < 30; i++) d[i] = 5;
    ok = build_table(l, 288, LIT_TB, t.lit, LIT_TABLE_MAX, 0) && build_table(d, 30, DIST_TB, t.dist, DIST_TABLE_MAX, 1); if (ok) pair_literals(t.lit); } };

// Decode one raw DEFLATE stream of known output size. 0 = ok, <0 = error (caller falls back to zlib).
static inline int inflate_raw(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len, InflateTables* scratch) {
    static const FixedTables fixed;
    const uint8_t* ip = in; const uint8_t* const in_end = in + in_len;
    uint8_t* op = out; uint8_t* const out_end = out + out_len;
    uint64_t bb = 0; int bc = 0;
#define XCK_REFILL() do { if (ip + 8 <= in_end) { uint64_t w_; memcpy(&w_, ip, 8); bb |= w_ << bc; ip += (63 - bc) >> 3; bc |= 56; } \
                          else { while (bc <= 56 && ip < in_end) { bb |= (uint64_t)*ip++ << bc; bc += 8; } } } while (0)
#define XCK_BITS(n) ((uint32_t)(bb & ((1ull << (n)) - 1)))
#define XCK_DROP(n) do { bb >>= (n); bc -= (n); } while (0)
    for (;;) {
        XCK_REFILL();
        if (bc < 3) return -1;
        const uint32_t bfinal = XCK_BITS(1); const uint32_t btype = (uint32_t)((bb >> 1) & 3); XCK_DROP(3);
        if (btype == 0) {                                         // stored
            XCK_DROP(bc & 7);                                     // to a byte boundary
            XCK_REFILL();
            if (bc < 32) return -2;
            uint32_t len = XCK_BITS(16); uint32_t nlen = (uint32_t)((bb >> 16) & 0xffff); XCK_DROP(32);
            if ((len ^ nlen) != 0xffff) return -3;
            ip -= bc >> 3; bb = 0; bc = 0;                        // give whole bytes back to the input pointer
            if ((size_t)(in_end - ip) < len || (size_t)(out_end - op) < len) return -4;
            memcpy(op, ip, len); op += len; ip += len;
        } else if (btype == 3) return -5;
        else {
            const HuffEnt* lit; const HuffEnt* dist;
            if (btype == 1) { if (!fixed.ok) return -6; lit = fixed.t.lit; dist = fixed.t.dist; }
            else {
#ifdef XCK_INFLATE_PROF
                unsigned long long t_b0 = __rdtsc(); g_prof_nblocks++;
#endif
                if (bc < 14) return -7;
                const int hlit = (int)XCK_BITS(5) + 257; XCK_DROP(5);
                const int hdist = (int)XCK_BITS(5) + 1; XCK_DROP(5);
                const int hclen = (int)XCK_BITS(4) + 4; XCK_DROP(4);
                if (hlit > 286 || hdist > 30) return -8;
                static const uint8_t order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
                uint8_t pl[19] = {0};
                for (int i = 0; i < hclen; i++) { if (bc < 3) XCK_REFILL(); if (bc < 3) return -9; pl[order[i]] = (uint8_t)XCK_BITS(3); XCK_DROP(3); }
                HuffEnt pre[1 << PRE_TB];
                if (!build_table(pl, 19, PRE_TB, pre, 1 << PRE_TB, 2)) return -10;
                uint8_t lens[286 + 30 + 140]; int n = 0; const int tot = hlit + hdist;
                while (n < tot) {
                    XCK_REFILL();
                    HuffEnt e = pre[XCK_BITS(PRE_TB)];
                    if (e.op || e.len > bc) return -11;
                    XCK_DROP(e.len);
                    if (e.val < 16) lens[n++] = (uint8_t)e.val;
                    else if (e.val == 16) { if (!n) return -12; int r = 3 + (int)XCK_BITS(2); XCK_DROP(2); uint8_t v = lens[n - 1]; while (r--) lens[n++] = v; }
                    else if (e.val == 17) { int r = 3 + (int)XCK_BITS(3); XCK_DROP(3); while (r--) lens[n++] = 0; }
                    else { int r = 11 + (int)XCK_BITS(7); XCK_DROP(7); while (r--) lens[n++] = 0; }
                    if (bc < 0) return -13;
                }
                if (n != tot || lens[256] == 0) return -14;
                if (!build_table(lens, hlit, LIT_TB, scratch->lit, LIT_TABLE_MAX, 0)) return -15;
                if (!build_table(lens + hlit, hdist, DIST_TB, scratch->dist, DIST_TABLE_MAX, 1)) return -16;
                pair_literals(scratch->lit);
#ifdef XCK_INFLATE_PROF
                g_prof_build += __rdtsc() - t_b0;
#endif
                lit = scratch->lit; dist = scratch->dist;
            }
            // ---- fast loop: far from both buffer ends, so no per-symbol bounds checks; one branch-free refill per
            //      iteration gives >= 56 bits, enough for 2 literals (<= 15 + 11 bits) or a full match (<= 48 bits)
            bool eob = false;
            HuffEnt e; bool have = false;                         // have: e was looked up by the previous iteration and not consumed yet
            while (!eob && ip + 8 <= in_end && (size_t)(out_end - op) >= 280) {
                { uint64_t w_; memcpy(&w_, ip, 8); bb |= w_ << bc; ip += (63 - bc) >> 3; bc |= 56; }
                if (!have) e = lit[XCK_BITS(LIT_TB)];             // (a refill only adds bits above the ones an earlier lookup used)
                have = false;
                if (e.op & 0x80) { XCK_DROP(LIT_TB); e = lit[e.val + XCK_BITS(e.op & 15)]; }
                XCK_DROP(e.len);
                if (e.op <= 3) {                                  // one to four literals per lookup; up to four lookups per refill
                    memcpy(op, &e.val, 4); op += 1 + e.op;
                    e = lit[XCK_BITS(LIT_TB)];
                    if (e.op <= 3) {
                        XCK_DROP(e.len); memcpy(op, &e.val, 4); op += 1 + e.op;
                        e = lit[XCK_BITS(LIT_TB)];
                        if (e.op <= 3) {
                            XCK_DROP(e.len); memcpy(op, &e.val, 4); op += 1 + e.op;
                            e = lit[XCK_BITS(LIT_TB)];            // >= 19 bits are left here: still a lookup on valid bits
                            if (e.op <= 3) { XCK_DROP(e.len); memcpy(op, &e.val, 4); op += 1 + e.op; continue; }
                        }
                    }
                    have = true;                                  // the entry that ended the run of literals opens the next iteration
                    continue;
                }
                if (e.op == 0x20) { eob = true; break; }
                if (e.op & 0x40) return -18;
                uint32_t length = e.val + XCK_BITS(e.op & 15); XCK_DROP(e.op & 15);
                HuffEnt d = dist[XCK_BITS(DIST_TB)];
                if (d.op & 0x80) { XCK_DROP(DIST_TB); d = dist[d.val + XCK_BITS(d.op & 15)]; }
                XCK_DROP(d.len);
                if (!(d.op & 0x10) || (d.op & 0x40)) return -20;
                uint32_t offset = d.val + XCK_BITS(d.op & 15); XCK_DROP(d.op & 15);
                if (offset > (size_t)(op - out)) return -22;
                XCK_INFLATE_MATCH_HOOK(offset, length);
                const uint8_t* src = op - offset;
                uint8_t* const stop = op + length;
                if (offset >= 8) {
                    do { uint64_t w_; memcpy(&w_, src, 8); memcpy(op, &w_, 8); src += 8; op += 8; } while (op < stop);
                } else if (offset == 1) {
                    uint64_t w_ = 0x0101010101010101ull * *src;
                    do { memcpy(op, &w_, 8); op += 8; } while (op < stop);
                } else {
                    do { *op++ = *src++; } while (op < stop);
                }
                op = stop;
            }
            while (!eob) {                                        // careful loop near the buffer ends
                if (bc < 48) XCK_REFILL();
                HuffEnt e = lit[XCK_BITS(LIT_TB)];
                if (e.op & 0x80) { XCK_DROP(LIT_TB); e = lit[e.val + XCK_BITS(e.op & 15)]; }
                XCK_DROP(e.len);
                if (e.op <= 3) {
                    if ((size_t)(out_end - op) < 1u + e.op) return -17;
                    for (uint32_t k = 0, v_ = e.val; k <= e.op; k++, v_ >>= 8) *op++ = (uint8_t)v_;
                    continue;
                }
                if (e.op == 0x20) break;                          // end of block
                if (e.op & 0x40) return -18;
                if (bc < 0) return -19;
                uint32_t length = e.val + XCK_BITS(e.op & 15); XCK_DROP(e.op & 15);
                if (bc < 32) XCK_REFILL();
                HuffEnt d = dist[XCK_BITS(DIST_TB)];
                if (d.op & 0x80) { XCK_DROP(DIST_TB); d = dist[d.val + XCK_BITS(d.op & 15)]; }
                XCK_DROP(d.len);
                if (!(d.op & 0x10) || (d.op & 0x40)) return -20;
                uint32_t offset = d.val + XCK_BITS(d.op & 15); XCK_DROP(d.op & 15);
                if (bc < 0) return -21;
                if (offset > (size_t)(op - out) || length > (size_t)(out_end - op)) return -22;
                XCK_INFLATE_MATCH_HOOK(offset, length);
                const uint8_t* src = op - offset;
                for (uint32_t k = 0; k < length; k++) op[k] = src[k];
                op += length;
            }
            if (bc < 0) return -24;
        }
        if (bfinal) break;
    }
#undef XCK_REFILL
#undef XCK_BITS
#undef XCK_DROP
    return op == out_end ? 0 : -23;
}

}  // namespace xck
