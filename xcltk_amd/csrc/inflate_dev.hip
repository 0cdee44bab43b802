// inflate_dev.hip - raw-DEFLATE decoder for BGZF blocks on the GPU (gfx950): one 64-lane wave per block.
//
// Replaces, for the share of the chunks the host decoder hands over (csrc/bam.cpp, XCK_GPU_INFLATE), the inflate that pysam / htslib
// do inside AlignmentFile.fetch (xcltk/rdr/fc/core.py:73-76, utils/sam.py:105-118); the record walk and the field parse stay on the host.
// Why: end to end the engine was bound by the host's BGZF inflate (73 - 85 % of the ingest CPU time on 16 cores,
// profiles/r02_d_ingest_*.log) while the GPU idled.  BGZF blocks are independent <= 64 KiB deflate streams, so a chunk of
// ~750 blocks is 750 waves of work.  Inside a block the SYMBOL BOUNDARIES are serial - they are only known after the previous
// symbol - but what would be decoded at a given bit offset is not:
//   * block headers and the code-length code: lane 0, serially (a few per cent of a block's time);
//   * decode tables: built by all lanes into LDS (9-bit primary + sub-tables for literal / length, 8-bit for distance; a code that
//     needs more LDS than the block's budget marks the block "not done" and the host inflates it - the compressed bytes never left
//     the host);
//   * symbols: 64 bit offsets per round - every lane decodes the code that would start at ITS offset, a scalar walk over the lanes'
//     code lengths (four symbols per hop, prepared by two rounds of ds_bpermute) finds the offsets where symbols really start, those
//     lanes store their literals at positions from a prefix sum, and the wave copies the matches
//     (out[dst + i] = out[dst - dist + i % dist] reads only bytes that already exist, so overlapping matches need no serial loop);
//   * the finished block goes to the chunk's pinned host block by the same wave (16-byte stores over PCIe).
// Measured: DESIGN.md section 6, profiles/experiments/gpu_inflate/ (every step of the way, against zlib block by block).
// Status per block: 0 = inflated (exactly isize bytes), non-zero = left to the host decoder (csrc/inflate_fast.h / zlib).
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <hip/hip_runtime.h>
#include "inflate_dev.h"

namespace xck {

namespace {

struct DHuff { uint16_t val; uint8_t len; uint8_t op; };
// op: 0 literal | 0x10+x length / distance base in val with x extra bits | 0x20 end of block | 0x40 invalid | 0x80+s sub-table link

// A 9-bit primary literal / length table: in the 64-offsets loop some lane takes the sub-table path in (nearly) every round anyway, so
// a smaller primary costs nothing there and 3.5 KB less LDS is one to four more waves per CU (measured in the harness: + 23 - 30 %).
#ifndef XCK_D_LIT_TB
#define XCK_D_LIT_TB 9
#endif
constexpr int D_LIT_TB = XCK_D_LIT_TB, D_DIST_TB = 8;
constexpr int D_LIT_MAX = (1 << D_LIT_TB) + 768;          // primary + sub-table budget (blocks that need more go to the host)
constexpr int D_DIST_MAX = (1 << D_DIST_TB) + 256;

__device__ const uint16_t d_len_base[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
__device__ const uint8_t  d_len_extra[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
__device__ const uint16_t d_dist_base[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
__device__ const uint8_t  d_dist_extra[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
__device__ const uint8_t  d_cl_order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};

struct Smem {
    DHuff lit[D_LIT_MAX];
    DHuff dist[D_DIST_MAX];
    uint8_t lens[320];                    // code lengths: literal / length alphabet, then distance alphabet
    uint16_t code[320];                   // canonical code (bit-reversed) of every symbol
    uint16_t sub_off[1 << D_LIT_TB];      // sub-table offset per primary index (0 = none)
    uint8_t  sub_bits[1 << D_LIT_TB];
    int32_t  count[16], first[16];
    int32_t  used, ok;
};

__device__ __forceinline__ uint32_t brev(uint32_t v, int n) { return __brev(v) >> (32 - n); }

// all 64 lanes: decode table for `n_sym` code lengths at lens[]; kind 0 literal / length, 1 distance.  Returns false (uniform)
// when the code is over-subscribed or does not fit the LDS budget.
__device__ bool build_table(Smem& sm, const uint8_t* lens, uint16_t* code, int n_sym, int tb, DHuff* tab, int tab_max, int kind, int lane) {
    if (lane < 16) sm.count[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {                                       // counts, first codes, canonical code per symbol: 300 cheap serial steps
        for (int i = 0; i < n_sym; i++) sm.count[lens[i]]++;
        sm.count[0] = 0;
        int left = 1, okf = 1; uint32_t c = 0;
        for (int l = 1; l <= 15; l++) { left = left * 2 - sm.count[l]; if (left < 0) okf = 0; }
        int nxt[16];
        for (int l = 1; l <= 15; l++) { c = (c + (uint32_t)sm.count[l - 1]) << 1; nxt[l] = (int)c; }
        for (int i = 0; i < n_sym; i++) { const int l = lens[i]; code[i] = l ? (uint16_t)brev((uint32_t)nxt[l]++, l) : 0; }
        sm.ok = okf; sm.used = 1 << tb;
    }
    __builtin_amdgcn_wave_barrier();
    if (!sm.ok) return false;
    const int psize = 1 << tb;
    DHuff inval; inval.val = 0; inval.len = 1; inval.op = 0x40;
    for (int i = lane; i < psize; i += 64) { tab[i] = inval; sm.sub_bits[i] = 0; sm.sub_off[i] = 0; }
    __builtin_amdgcn_wave_barrier();
    // long codes: extra bits every primary prefix needs (max over its codes) - LDS atomicMax on bytes is not available, use lane 0
    if (lane == 0) {
        bool any = false;
        for (int s = 0; s < n_sym; s++) { const int l = lens[s]; if (l > tb) { any = true; const uint32_t p = code[s] & (uint32_t)(psize - 1); if (l - tb > sm.sub_bits[p]) sm.sub_bits[p] = (uint8_t)(l - tb); } }
        if (any) {
            int used = psize;
            for (int p = 0; p < psize && used <= tab_max; p++) if (sm.sub_bits[p]) {
                const int sz = 1 << sm.sub_bits[p];
                if (used + sz > tab_max) { used = tab_max + 1; break; }
                sm.sub_off[p] = (uint16_t)used;
                DHuff lk; lk.val = (uint16_t)used; lk.len = (uint8_t)tb; lk.op = (uint8_t)(0x80 | sm.sub_bits[p]);
                tab[p] = lk;
                for (int k = 0; k < sz; k++) tab[used + k] = inval;
                used += sz;
            }
            sm.used = used;
            if (used > tab_max) sm.ok = 0;
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (!sm.ok) return false;
    for (int s = lane; s < n_sym; s += 64) {               // every lane fills the entries of its symbols
        const int l = lens[s]; if (!l) continue;
        DHuff e; e.len = (uint8_t)l;
        if (kind == 0) {
            if (s < 256) { e.val = (uint16_t)s; e.op = 0; }
            else if (s == 256) { e.val = 0; e.op = 0x20; }
            else if (s <= 285) { e.val = d_len_base[s - 257]; e.op = (uint8_t)(0x10 | d_len_extra[s - 257]); }
            else { e.val = 0; e.op = 0x40; }
        } else {
            if (s < 30) { e.val = d_dist_base[s]; e.op = (uint8_t)(0x10 | d_dist_extra[s]); } else { e.val = 0; e.op = 0x40; }
        }
        const uint32_t r = code[s];
        if (l <= tb) { for (uint32_t k = r; k < (uint32_t)psize; k += 1u << l) tab[k] = e; }
        else {
            const uint32_t p = r & (uint32_t)(psize - 1); const int sb = sm.sub_bits[p];
            e.len = (uint8_t)(l - tb);
            for (uint32_t k = r >> tb; k < (1u << sb); k += 1u << (l - tb)) tab[sm.sub_off[p] + k] = e;
        }
    }
    __builtin_amdgcn_wave_barrier();
    return true;
}

// The compressed stream reaches lane 0 through an LDS window: the whole wave loads IN_WIN bytes at a time (aligned dwords,
// coalesced), lane 0 takes aligned dwords out of LDS.  (A per-lane byte pointer into global memory cost four dependent byte
// loads - about a microsecond - per 32 bits of input.)  Coordinates: byte s of the stream sits at c = s + A of the 4-byte
// aligned buffer that starts at (stream address - A), A = address & 3.
constexpr int IN_WIN = 1024;                               // bytes staged per fill (a multiple of 256)
struct BitIn {
    uint32_t pos;                                          // next aligned-buffer coordinate to load (multiple of 2)
    uint32_t wlo;                                          // the window holds coordinates [wlo, wlo + IN_WIN)
    uint64_t bb; int bc;
    __device__ __forceinline__ uint32_t bits(int n) const { return (uint32_t)(bb & ((1ull << n) - 1)); }
    __device__ __forceinline__ void drop(int n) { bb >>= n; bc -= n; }
    // 16 bits at a time until more than 48 bits are buffered (a length + distance pair needs at most 48); false = the window
    // is used up before that
    __device__ __forceinline__ bool refill(const uint32_t* win) {
        const uint16_t* w16 = (const uint16_t*)win;
        while (bc <= 48) {
            if (pos + 2 > wlo + IN_WIN) return false;
            bb |= (uint64_t)w16[(pos - wlo) >> 1] << bc; bc += 16; pos += 2;
        }
        return true;
    }
};

}  // namespace

// inclusive prefix sum over the wave (DPP: shifts inside the rows of 16, then the row totals travel on)
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);    // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);    // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);    // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);    // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);    // row_bcast:15 -> rows 1, 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);    // row_bcast:31 -> rows 2, 3
    return x;
}

// Output bytes go straight to global memory and a match reads them back from there (behind a fence when they were stored since the
// last one).  (A variant that kept the last 4 / 8 KB in an LDS ring - near matches as LDS-to-LDS copies, aligned 1 KB flushes - cost a
// third of the occupancy and measured equal or slower: profiles/experiments/gpu_inflate/inflate_dev_with_lds_ring.hip.)
__device__ unsigned long long d_prof[8];          // PROF builds only (tools/gpu_inflate_bench): cycles of wave-time per phase, summed over the blocks
#define XCK_PROF_AT(k) do { if constexpr (PROF) { const long long t_ = clock64(); prof[k] += (unsigned long long)(t_ - t_prev); t_prev = t_; } } while (0)

template <bool PROF>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_inflate(const uint8_t* __restrict__ in, const DevBlock* __restrict__ blocks, int n_blocks,
                                                uint8_t* out, int32_t* __restrict__ status, uint8_t* host_out) {
    __shared__ Smem sm;
    __shared__ uint32_t win[IN_WIN / 4];
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= n_blocks) return;
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long t_prev = PROF ? clock64() : 0;   // 0 header, 1 tables, 2 window, 3 decode, 4 walk, 5 scan + literals, 6 matches, 7 flush / rest
    const DevBlock blk = blocks[b];
    uint8_t* const o0v = out + blk.out_off;                            // the block's output
    const uint32_t out_end = blk.out_len;
    if (blk.out_len == 0) { if (lane == 0) status[b] = 0; return; }
    const uint8_t* const sp = in + blk.in_off;
    const uint32_t A = (uint32_t)((uintptr_t)sp & 3u);
    const uint32_t* const abase = (const uint32_t*)(sp - A);           // 4-byte aligned view of the stream
    const uint32_t c_end = A + blk.in_len;                             // first coordinate past the stream
    auto fill = [&](uint32_t wlo) {                                    // all lanes; words past the stream read as zero
#pragma unroll
        for (int j = 0; j < IN_WIN / 256; j++) { const uint32_t c = wlo + (uint32_t)(j * 256 + lane * 4); win[j * 64 + lane] = c < c_end ? abase[c >> 2] : 0u; }
        __builtin_amdgcn_wave_barrier();
    };
    BitIn bi; bi.pos = 0; bi.wlo = 0; bi.bb = 0; bi.bc = 0;            // (lane 0's state; the other lanes carry dead copies)
    fill(0);
    if (lane == 0) { bi.refill(win); bi.drop((int)(8 * A)); }          // skip the A bytes before the stream
    uint32_t op = 0;                                                   // output position (wave-uniform)
    int err = 0;
    for (;;) {
        // ---- block header (lane 0), broadcast; the window is moved first so that a whole header (< 400 bytes) fits ----
        { uint32_t pos = __builtin_amdgcn_readfirstlane(bi.pos), wlo = __builtin_amdgcn_readfirstlane(bi.wlo);
          if (pos > wlo + IN_WIN - 512) { fill(pos & ~3u); bi.wlo = pos & ~3u; } }
        int btype = 0, bfinal = 0, hlit = 0, hdist = 0;
        if (lane == 0) {
            bi.refill(win);
            if (bi.bc < 3) err = 1;
            bfinal = (int)bi.bits(1); btype = (int)((bi.bb >> 1) & 3); bi.drop(3);
            if (btype == 3) err = 2;
            if (!err && btype == 2) {
                bi.refill(win);
                hlit = (int)bi.bits(5) + 257; bi.drop(5); hdist = (int)bi.bits(5) + 1; bi.drop(5);
                const int hclen = (int)bi.bits(4) + 4; bi.drop(4);
                if (hlit > 286 || hdist > 30) err = 3;
                uint8_t pl[19];
                for (int i = 0; i < 19; i++) pl[i] = 0;
                for (int i = 0; i < hclen && !err; i++) { if (bi.bc < 3) bi.refill(win); if (bi.bc < 3) { err = 4; break; } pl[d_cl_order[i]] = (uint8_t)bi.bits(3); bi.drop(3); }
                // code-length code: canonical decode by counting (19 symbols, <= 7 bits)
                int cnt[8] = {0}, firstc[8], firsts[8]; uint8_t sorted[19];
                for (int i = 0; i < 19; i++) cnt[pl[i]]++;
                cnt[0] = 0; { int c = 0, q = 0; for (int l = 1; l <= 7; l++) { c = (c + cnt[l - 1]) << 1; firstc[l] = c; firsts[l] = q; q += cnt[l]; } }
                { int pos[8]; for (int l = 1; l <= 7; l++) pos[l] = firsts[l]; for (int i = 0; i < 19; i++) if (pl[i]) sorted[pos[pl[i]]++] = (uint8_t)i; }
                int n = 0; const int tot = hlit + hdist;
                while (n < tot && !err) {
                    bi.refill(win);
                    int code = 0, l = 0, sym = -1;
                    for (l = 1; l <= 7; l++) {                         // bit-serial canonical decode (codes arrive most significant bit first)
                        code = (code << 1) | (int)((bi.bb >> (l - 1)) & 1);
                        const int d = code - firstc[l];
                        if (cnt[l] && d >= 0 && d < cnt[l]) { sym = sorted[firsts[l] + d]; break; }
                    }
                    if (sym < 0 || l > bi.bc) { err = 5; break; }
                    bi.drop(l);
                    if (sym < 16) sm.lens[n++] = (uint8_t)sym;
                    else {
                        int r; uint8_t v = 0;
                        if (sym == 16) { if (!n) { err = 6; break; } r = 3 + (int)bi.bits(2); bi.drop(2); v = sm.lens[n - 1]; }
                        else if (sym == 17) { r = 3 + (int)bi.bits(3); bi.drop(3); }
                        else { r = 11 + (int)bi.bits(7); bi.drop(7); }
                        if (n + r > tot) { err = 7; break; }
                        while (r--) sm.lens[n++] = v;
                    }
                    if (bi.bc < 0) { err = 8; break; }
                }
                if (!err && sm.lens[256] == 0) err = 9;
            } else if (!err && btype == 1) {
                for (int i = 0; i < 144; i++) sm.lens[i] = 8; for (int i = 144; i < 256; i++) sm.lens[i] = 9;
                for (int i = 256; i < 280; i++) sm.lens[i] = 7; for (int i = 280; i < 288; i++) sm.lens[i] = 8;
                for (int i = 0; i < 30; i++) sm.lens[288 + i] = 5;
                hlit = 288; hdist = 30;
            }
        }
        err = __builtin_amdgcn_readfirstlane(err);
        if (err) break;
        btype = __builtin_amdgcn_readfirstlane(btype); bfinal = __builtin_amdgcn_readfirstlane(bfinal);
        hlit = __builtin_amdgcn_readfirstlane(hlit); hdist = __builtin_amdgcn_readfirstlane(hdist);
        __builtin_amdgcn_wave_barrier();
        XCK_PROF_AT(0);
        if (btype == 0) {                                              // stored block: lane 0 finds the byte position, the wave copies
            uint32_t len = 0, s_at = 0;
            if (lane == 0) {
                bi.drop(bi.bc & 7);
                uint32_t c = bi.pos - (uint32_t)(bi.bc >> 3);          // coordinate of the next unread byte
                bi.bb = 0; bi.bc = 0;
                if (c + 4 > c_end) err = 10;
                else {
                    const uint8_t* q = sp - A + c;
                    len = (uint32_t)q[0] | ((uint32_t)q[1] << 8); const uint32_t nlen = (uint32_t)q[2] | ((uint32_t)q[3] << 8);
                    if ((len ^ nlen) != 0xffff) err = 11;
                    c += 4;
                    if (!err && (c + len > c_end || out_end - op < len)) err = 12;
                    s_at = c;
                    // resume the bit stream after the stored bytes: aligned coordinate + partial dword
                    const uint32_t nxt = c + len;
                    bi.pos = nxt & ~3u;                                // the window is moved here below (4-byte aligned for the fill)
                    bi.bc = -(int)(8 * (nxt & 3u));                    // bits to discard once the next words are loaded
                }
            }
            err = __builtin_amdgcn_readfirstlane(err);
            if (err) break;
            len = __builtin_amdgcn_readfirstlane(len); s_at = __builtin_amdgcn_readfirstlane(s_at);
            const uint8_t* s0 = sp - A + s_at;
            for (uint32_t i = lane; i < len; i += 64) o0v[op + i] = s0[i];
            op += len;
            // re-prime the bit buffer for whatever follows the stored block
            { const uint32_t pos = __builtin_amdgcn_readfirstlane(bi.pos); fill(pos & ~3u); bi.wlo = pos & ~3u; }
            if (lane == 0) { const int skip = -bi.bc; bi.bc = 0; bi.bb = 0; bi.refill(win); if (skip) bi.drop(skip); }
        } else {
            if (!build_table(sm, sm.lens, sm.code, hlit, D_LIT_TB, sm.lit, D_LIT_MAX, 0, lane)) { err = 20; break; }
            if (!build_table(sm, sm.lens + hlit, sm.code, hdist, D_DIST_TB, sm.dist, D_DIST_MAX, 1, lane)) { err = 21; break; }
            XCK_PROF_AT(1);
            // ---- symbols, 64 bit offsets at a time.  A symbol's boundaries depend on all symbols before it, but WHAT would be decoded at a
            // given bit offset does not: lane i decodes the literal / length (+ distance) code that starts at bit `bitpos + i` - two
            // table look-ups per lane, all lanes at once -, then a scalar walk (readlane of the code lengths, no vector work) hops from
            // symbol start to symbol start across the 64 offsets.  The lanes that turned out to be starts store their literals in one
            // go (positions by a prefix sum of the output lengths); the matches among them are copied by the whole wave, in order.
            // (The first form of this kernel - lane 0 decoding symbol after symbol, ~45 vector instructions per literal - is archived
            // under profiles/experiments/gpu_inflate/; this loop is 2 - 2.9x faster on its own, r04_symbol_loop_ab.txt.)
            uint32_t bitpos = (uint32_t)__builtin_amdgcn_readfirstlane((int)(bi.pos * 8u - (uint32_t)bi.bc));
            uint32_t wlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)bi.wlo);
            uint32_t dirty_lo = 0;                                     // lowest position stored since the last fence (0 = assume everything)
            bool eob = false;
            while (!eob) {
                const uint32_t byte0 = bitpos >> 3;
                if (byte0 < wlo || byte0 + 32 > wlo + IN_WIN) { wlo = byte0 & ~3u; fill(wlo); }   // (bits lane 0 had buffered may lie before a window the header code moved)
                XCK_PROF_AT(2);
                const uint32_t wbase = wlo >> 2;
                const uint32_t bp = bitpos + (uint32_t)lane;
                uint32_t v;
                { const uint32_t wi = (bp >> 5) - wbase; v = __builtin_amdgcn_alignbit(win[wi + 1], win[wi], bp & 31u); }   // 32 stream bits from offset bp
                // a table entry as one 32-bit LDS read: val | len << 16 | op << 24
                uint32_t e = ((const uint32_t*)sm.lit)[v & ((1u << D_LIT_TB) - 1)];
                uint32_t used = (e >> 16) & 0xffu;
                if (e >> 31) { e = ((const uint32_t*)sm.lit)[(e & 0xffffu) + ((v >> D_LIT_TB) & ((1u << ((e >> 24) & 15u)) - 1))]; used = D_LIT_TB + ((e >> 16) & 0xffu); }
                const uint32_t eop = e >> 24;
                // kind: 0 literal, 1 match, 2 end of block, 3 invalid
                uint32_t kind = (eop & 0x50u) == 0x10u ? 1u : 3u;
                kind = eop == 0x20u ? 2u : kind; kind = eop == 0u ? 0u : kind;
                uint32_t mlen = 0, mdist = 0;
                if (kind == 1) {
                    const uint32_t x = eop & 15u;
                    mlen = (e & 0xffffu) + ((v >> used) & ((1u << x) - 1)); used += x;         // <= 20 bits so far
                    const uint32_t bp2 = bp + used;
                    uint32_t v2;
                    { const uint32_t wi = (bp2 >> 5) - wbase; v2 = __builtin_amdgcn_alignbit(win[wi + 1], win[wi], bp2 & 31u); }
                    uint32_t dd = ((const uint32_t*)sm.dist)[v2 & ((1u << D_DIST_TB) - 1)];
                    uint32_t u2 = (dd >> 16) & 0xffu;
                    if (dd >> 31) { dd = ((const uint32_t*)sm.dist)[(dd & 0xffffu) + ((v2 >> D_DIST_TB) & ((1u << ((dd >> 24) & 15u)) - 1))]; u2 = D_DIST_TB + ((dd >> 16) & 0xffu); }
                    if (((dd >> 24) & 0xd0u) != 0x10u) kind = 3;
                    else { const uint32_t y = (dd >> 24) & 15u; mdist = (dd & 0xffffu) + ((v2 >> u2) & ((1u << y) - 1)); used += u2 + y; }   // <= 48 bits
                }
                if (used == 0) kind = 3;
                uint32_t olen = kind == 0 ? 1u : (kind == 1 ? mlen : 0u);
                const uint32_t packed = used | (kind << 6) | (olen << 8);
                if constexpr (PROF) { asm volatile("" :: "v"(packed)); }
                XCK_PROF_AT(3);
                // the walk: which lanes start a symbol (scalar); a round ends after 64 offsets or at the end of the block
                uint64_t starts = 0; uint32_t cur = 0, stop = 0, total = 0;
                {
                    // ... four symbols per hop: every lane first learns, through two rounds of ds_bpermute, what a walk that STARTED at it would
                    // cross with its next four symbols (bits, output bytes, which lanes, whether it stops) - the scalar walk then needs
                    // three readlanes per four symbols instead of a readlane and two branches per symbol (the walk was 51 % of a round on the
                    // literal-heavy streams, r04_phase_clocks.txt).  hop word: bits (8) | stop (2) << 8 | output bytes << 10
                    auto join = [](uint32_t a, uint32_t b) { return a + (b & 0xffu) + (b & ~0x3ffu) + (b & 0x300u); };     // (a's stop field is 0 where this is used)
                    const uint32_t h1 = used | ((kind >= 2 ? kind : 0u) << 8) | (olen << 10);
                    const uint32_t lo1 = lane < 32 ? 1u << lane : 0u, hi1 = lane >= 32 ? 1u << (lane - 32) : 0u;
                    const uint32_t s1 = (uint32_t)lane + used; const bool v1 = kind < 2 && s1 < 64;
                    const uint32_t q1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((s1 & 63u) << 2), (int)h1);
                    const uint32_t h2 = v1 ? join(h1, q1) : h1;
                    const uint32_t lo2 = lo1 | (v1 && s1 < 32 ? 1u << s1 : 0u), hi2 = hi1 | (v1 && s1 >= 32 ? 1u << (s1 - 32) : 0u);
                    const uint32_t s2 = (uint32_t)lane + (h2 & 0xffu); const bool v2 = !(h2 & 0x300u) && s2 < 64;
                    const int a2 = (int)((s2 & 63u) << 2);
                    const uint32_t q2 = (uint32_t)__builtin_amdgcn_ds_bpermute(a2, (int)h2);
                    const uint32_t ql = (uint32_t)__builtin_amdgcn_ds_bpermute(a2, (int)lo2), qh = (uint32_t)__builtin_amdgcn_ds_bpermute(a2, (int)hi2);
                    const uint32_t h4 = v2 ? join(h2, q2) : h2, lo4 = v2 ? lo2 | ql : lo2, hi4 = v2 ? hi2 | qh : hi2;
                    uint32_t slo = 0, shi = 0;
                    while (cur < 64) {
                        const uint32_t p = (uint32_t)__builtin_amdgcn_readlane((int)h4, (int)cur);
                        slo |= (uint32_t)__builtin_amdgcn_readlane((int)lo4, (int)cur); shi |= (uint32_t)__builtin_amdgcn_readlane((int)hi4, (int)cur);
                        total += p >> 10; cur += p & 0xffu;
                        if (p & 0x300u) { stop = (p >> 8) & 3u; break; }
                    }
                    starts = ((uint64_t)shi << 32) | slo;
                }
                XCK_PROF_AT(4);
                if (stop == 3) { err = 30; break; }
                eob = stop == 2;
                if (total > out_end - op) { err = 33; break; }
                const bool is_start = (starts >> lane) & 1;
                olen = is_start ? olen : 0u;
                const uint32_t off = op + wave_scan_incl(olen) - olen;  // where this lane's output goes
                if (is_start && kind == 0) o0v[off] = (uint8_t)e;
                uint64_t mm = starts & __ballot(kind == 1);
                if (starts & ~mm) dirty_lo = min(dirty_lo, op);
                XCK_PROF_AT(5);
                {
                    // All matches of the round in ONE load / store pair, lane j taking the j-th matched byte - when they fit 64 bytes and none
                    // reads what an earlier match of the same round writes (the usual case: a round makes 30 - 60 bytes and a BAM record's
                    // matches reach back a record or more).  The scalar loop over the matches only hands out (source, destination) per lane;
                    // the memory round trip - and the fence, if any source was stored since the last one - is paid once per round instead of
                    // once per match (matches were 44 - 48 % of a round on zlib-6 / Cell Ranger-shaped streams, r04_phase_clocks_4hop.txt).
                    if (mm) {
                        const bool is_m = is_start && kind == 1;
                        const uint32_t minc = wave_scan_incl(is_m ? mlen : 0u);
                        const uint32_t mtot = (uint32_t)__builtin_amdgcn_readlane((int)minc, 63);
                        const uint32_t first_dst = (uint32_t)__builtin_amdgcn_readlane((int)off, __builtin_ctzll(mm));
                        const uint32_t src_lo = off - mdist, src_hi = src_lo + min(mlen, mdist);
                        const bool bad = is_m && mdist > off;
                        const bool dep = is_m && off != first_dst && src_hi > first_dst;        // reads bytes an earlier match of this round makes
                        if (__ballot(bad)) { err = 33; break; }
                        if (mtot <= 64 && !__ballot(dep)) {
                            const bool need_fence = __ballot(is_m && src_hi > dirty_lo) != 0;
                            uint32_t my_src = 0, my_dst = 0; bool act = false;
                            for (uint64_t m2 = mm; m2; m2 &= m2 - 1) {
                                const int l = __builtin_ctzll(m2);
                                const uint32_t ml = (uint32_t)__builtin_amdgcn_readlane((int)mlen, l), md = (uint32_t)__builtin_amdgcn_readlane((int)mdist, l);
                                const uint32_t dst = (uint32_t)__builtin_amdgcn_readlane((int)off, l), mo = (uint32_t)__builtin_amdgcn_readlane((int)minc, l) - ml;
                                const uint32_t t = (uint32_t)lane - mo;
                                if (t < ml) { act = true; my_dst = dst + t; my_src = dst - md + (md >= ml ? t : t % md); }   // (overlapping: the pattern repeats)
                            }
                            if (need_fence) { __threadfence_block(); dirty_lo = 0xffffffffu; }   // a source was stored since the last fence
                            uint8_t bv = 0;
                            if (act) bv = o0v[my_src];
                            if (act) o0v[my_dst] = bv;
                            dirty_lo = min(dirty_lo, first_dst);
                            mm = 0;
                        }
                    }
                }
                while (mm) {
                    const int l = __builtin_ctzll(mm); mm &= mm - 1;
                    const uint32_t ml = (uint32_t)__builtin_amdgcn_readlane((int)mlen, l), md = (uint32_t)__builtin_amdgcn_readlane((int)mdist, l);
                    const uint32_t dst = (uint32_t)__builtin_amdgcn_readlane((int)off, l);
                    if (md > dst) { err = 33; break; }
                    const uint32_t src_hi = dst - md + min(ml, md);
                    if (src_hi > dirty_lo) { __threadfence_block(); dirty_lo = 0xffffffffu; }   // the bytes it reads were stored since the last fence
                    const uint8_t* src = o0v + dst - md;
                    if (md >= ml) { for (uint32_t i = lane; i < ml; i += 64) o0v[dst + i] = src[i]; }
                    else { for (uint32_t i = lane; i < ml; i += 64) o0v[dst + i] = src[i % md]; }   // overlapping: the pattern repeats
                    dirty_lo = min(dirty_lo, dst);
                }
                if (err) break;
                XCK_PROF_AT(6);
                op += total; bitpos += cur;
                XCK_PROF_AT(7);
            }
            if (err) break;
            // hand the bit position back to lane 0's serial reader (block headers, stored blocks)
            { const uint32_t pos = (bitpos >> 4) << 1;
              if (pos < wlo || pos + 16 > wlo + IN_WIN) { wlo = pos & ~3u; fill(wlo); }
              bi.wlo = wlo; bi.pos = pos; bi.bb = 0; bi.bc = 0; bi.refill(win); bi.drop((int)(bitpos & 15u)); }
        }
        if (bfinal) break;
    }
    if constexpr (PROF) { XCK_PROF_AT(7); if (lane == 0) for (int k = 0; k < 8; k++) atomicAdd(&d_prof[k], prof[k]); }
    {
        // consumed input must lie inside the stream (the window is zero beyond it: a truncated stream must not pass)
        const long long used_bits = (long long)((uint32_t)__builtin_amdgcn_readfirstlane((int)bi.pos) - A) * 8 - __builtin_amdgcn_readfirstlane(bi.bc);
        if (!err && used_bits > (long long)blk.in_len * 8) err = 41;
        if (!err && op != out_end) err = 40;
        // the block's bytes -> the chunk's (mapped, pinned) host block, by the wave that made them: the copy-out then overlaps the other
        // waves' decoding instead of following the whole launch as a kernel of its own (which cost every chunk 1 - 3 ms of latency and
        // sent its 50 MB over PCIe in one burst).  16-byte stores where the block's own bytes fill an aligned 16, single bytes at its ends
        // (the neighbours' bytes are theirs to write).  host_out and out are equally aligned (both allocations are page aligned).
        if (host_out && !err) {
            __threadfence_block();
            const uint8_t* ob = out + blk.out_off; uint8_t* hb = host_out + blk.out_off; const uint32_t n = blk.out_len;
            const uint32_t head = min((16u - (uint32_t)((uintptr_t)ob & 15u)) & 15u, n), body = (n - head) & ~15u;
            if ((uint32_t)lane < head) hb[lane] = ob[lane];
            for (uint32_t i = (uint32_t)lane * 16; i < body; i += 1024) *(uint4*)(hb + head + i) = *(const uint4*)(ob + head + i);
            for (uint32_t i = head + body + (uint32_t)lane; i < n; i += 64) hb[i] = ob[i];
        }
        if (lane == 0) status[b] = err;
    }
}


// PROF variants (10, 11): the per-phase cycle sums since the last call (and zeroes them)
void dev_inflate_read_prof(unsigned long long out[8]) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(d_prof), 8 * sizeof(unsigned long long));
    unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(d_prof), z, sizeof z);
}

int dev_inflate_launch(hipStream_t stream, const uint8_t* d_in, const DevBlock* d_blocks, int n_blocks, uint8_t* d_out, int32_t* d_status, uint8_t* host_out, int variant) {
    if (n_blocks <= 0) return 0;
    const dim3 g((unsigned)n_blocks), t(64);
    if (variant >= 10) hipLaunchKernelGGL((k_inflate<true>), g, t, 0, stream, d_in, d_blocks, n_blocks, d_out, d_status, host_out);
    else hipLaunchKernelGGL((k_inflate<false>), g, t, 0, stream, d_in, d_blocks, n_blocks, d_out, d_status, host_out);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- one chunk in flight (see inflate_dev.h) ------------------------------------------------------------------------------
GpuInflateSlot* gpu_inflate_slot_create(int device, int free_cus, bool verbose) {
    if (device < 0 || hipSetDevice(device) != hipSuccess) return nullptr;
    GpuInflateSlot* s = new GpuInflateSlot(); s->device = device;
    // LOWEST stream priority: an inflate launch runs for tens of milliseconds, and on a hardware queue it shares with the engine's
    // streams the join kernels of the chunks being pushed queued up behind it (first version: the coordinator's push time went from
    // 0.25 to 2.6 s per 100 M records)
    // ... and a CU mask that leaves 32 CUs to the engine: the inflate waves live for tens of milliseconds and hold ~14 KB of LDS
    // each; once a few chunks are in flight they fill every CU's LDS, and a join block (26 KB) found no room until some retired
    // (low stream priority does not evict running waves: push time 1.6 s per 100 M records).  Fallback: a low-priority stream.
    hipDeviceProp_t prop;
    bool masked = false;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 64) {
        const int n_cu = prop.multiProcessorCount, keep_free = std::max(8, std::min(free_cus, n_cu / 2));
        uint32_t mask[16] = {0};
        for (int cu = 0; cu < n_cu - keep_free && cu < 512; cu++) mask[cu >> 5] |= 1u << (cu & 31);
        masked = hipExtStreamCreateWithCUMask(&s->stream, (uint32_t)((n_cu + 31) / 32), mask) == hipSuccess;
        if (!masked) { (void)hipGetLastError(); s->stream = nullptr; }
        if (verbose) fprintf(stderr, "[xck] GPU inflate stream: %d CUs, %d kept free of inflate waves: CU mask %s\n", n_cu, keep_free, masked ? "applied" : "REFUSED (low-priority stream instead)");
    }
    if (!masked) {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&s->stream, hipStreamNonBlocking, least) != hipSuccess) { gpu_inflate_slot_destroy(s); return nullptr; }
    }
    if (hipEventCreateWithFlags(&s->done, hipEventDisableTiming) != hipSuccess) { gpu_inflate_slot_destroy(s); return nullptr; }
    return s;
}
static void slot_free_buffers(GpuInflateSlot* s) {
    if (s->h_in) hipHostFree(s->h_in); if (s->h_out) hipHostFree(s->h_out); if (s->d_out) hipFree(s->d_out);
    if (s->h_bl) hipHostFree(s->h_bl); if (s->h_st) hipHostFree(s->h_st);
    s->h_in = s->a_in = s->h_out = s->a_out = s->d_out = nullptr; s->h_bl = s->a_bl = nullptr; s->h_st = s->a_st = nullptr; s->cap_in = s->cap_out = s->cap_bl = 0;
}
void gpu_inflate_slot_destroy(GpuInflateSlot* s) {
    if (!s) return;
    if (hipSetDevice(s->device) == hipSuccess) {
        if (s->stream) hipStreamSynchronize(s->stream);
        slot_free_buffers(s);
        if (s->done) hipEventDestroy(s->done);
        if (s->stream) hipStreamDestroy(s->stream);
    }
    delete s;
}
bool gpu_inflate_slot_reserve(GpuInflateSlot* s, size_t in_bytes, size_t out_bytes, size_t n_blocks) {
    if (in_bytes <= s->cap_in && out_bytes <= s->cap_out && n_blocks <= s->cap_bl) return true;
    if (hipSetDevice(s->device) != hipSuccess || hipStreamSynchronize(s->stream) != hipSuccess) return false;
    auto grow = [](size_t need, size_t have) { return need > have ? need + need / 4 + 4096 : have; };
    const size_t ci = grow(in_bytes, s->cap_in), co = grow(out_bytes, s->cap_out), cb = grow(n_blocks, s->cap_bl);
    slot_free_buffers(s);
    auto host = [](void** h, void** a, size_t bytes) { return hipHostMalloc(h, bytes, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(a, *h, 0) == hipSuccess; };
    bool ok = host((void**)&s->h_in, (void**)&s->a_in, ci + 16) && host((void**)&s->h_out, (void**)&s->a_out, co + 64) && hipMalloc((void**)&s->d_out, co + 64) == hipSuccess &&
              host((void**)&s->h_bl, (void**)&s->a_bl, cb * sizeof(DevBlock)) && host((void**)&s->h_st, (void**)&s->a_st, cb * sizeof(int32_t));
    if (!ok) { slot_free_buffers(s); (void)hipGetLastError(); return false; }
    s->cap_in = ci; s->cap_out = co; s->cap_bl = cb;
    return true;
}
// No DMA engine on this path: the first version copied in and out with hipMemcpyAsync, and the engine's own copy of every decoded
// chunk (one hipMemcpyAsync per chunk, csrc/engine.hip engine_push_block) queued up behind the 48 MB copy-outs - the coordinator's
// push time rose from 0.25 to 1.6 s per 100 M records.  The kernel is bound by its serial Huffman chain, not by bytes: it takes
// the compressed stream from host memory in 1 KB coalesced windows, every wave stores its finished block to the host block, and the
// statuses go straight back.
int gpu_inflate_slot_launch(GpuInflateSlot* s, size_t in_bytes, size_t out_bytes, size_t n_blocks) {
    (void)in_bytes;
    if (hipSetDevice(s->device) != hipSuccess) return -1;
    for (size_t i = 0; i < n_blocks; i++) s->h_st[i] = -1;                  // (a block the kernel never reaches is left to the host)
    (void)out_bytes;
    if (dev_inflate_launch(s->stream, s->a_in, s->a_bl, (int)n_blocks, s->d_out, s->a_st, s->a_out, s->variant) != 0) { (void)hipGetLastError(); return -1; }
    if (hipEventRecord(s->done, s->stream) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return 0;
}
bool gpu_inflate_slot_done(GpuInflateSlot* s) { return hipEventQuery(s->done) != hipErrorNotReady; }
int gpu_inflate_slot_wait(GpuInflateSlot* s) { return hipEventSynchronize(s->done) == hipSuccess ? 0 : -1; }

}  // namespace xck
