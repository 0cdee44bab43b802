// fold_partition.h - basefc fold WITHOUT a sort (64-bit keys): adaptive two-level partition + one LDS pass per bucket.
// Included by engine.hip inside namespace xck (it uses EngineImpl, Arena, KeyLayout, set_slot, copy_out, res_reserve).
//
// Replaces, for the count matrix, what the reference does with one Python set per (region, cell):
//   xcltk/rdr/fc/mcount.py:34-54 (MCount.add_read: set insert, len() at the end), rdr/fc/core.py:166-178.
//
// The keys (row | cell | umi) leave k_join in 16 shard slices, in tile order of a coordinate-sorted file: at any place of the
// stream only a handful of rows are active.  A radix sort throws that away (4 onesweep passes over (row, cell): 24 GB for
// the 3 GB of keys of configs[2]).  Here:
//
//   row geometry   k_pf_rowhist counts keys per row on a SAMPLE of the stream (every stride-th chunk), k_pf_rowplan gives every row
//                  G_r = 2^l cell groups so that a group holds about C / 2 keys (a cold gene: one group; a hot one: up to 2^lgG_max).
//                  A uniform G does not work: where genes are cold a chunk of the stream touches hundreds of rows, and G groups
//                  each would leave one key per (chunk, group) - nothing to aggregate, every write on its own (first version:
//                  8 ms per pass over the keys instead of 0.7).
//   level-1 cells  z = zbase[row] + (cell >> sg_row).  k_pf_hist<1> counts the keys per z (a chunk of the stream touches few z,
//                  so a block aggregates in an LDS table and flushes ~100 atomics), a scan turns the counts into offsets,
//                  k_pf_part<1> moves every key to its z (again one LDS table per chunk: a run of keys per (chunk, z), one
//                  returning atomic per run).
//   work items     consecutive z are merged into work items of at most CAP = 2C keys that stay inside one block of 2^sb
//                  (row, cell) pairs ("span"): z starts a new item when its first key opens a new page of C keys, when the
//                  span changes, or when it (or its predecessor) holds more than C keys on its own (k_pf_plan / k_pf_emit).
//   level 2        a z with more than CAP keys (a hot gene: thousands of keys per cell) is "big": its keys are partitioned
//                  once more by the low cell bits (k_pf_hist<2> / k_pf_part<2>, the same kernels on the level-1 output),
//                  and its sub-cells are merged into work items by the same page rule.
//   k_pf_bucket    one block per work item: distinct (row, cell, umi) by an LDS hash set, the (row, cell) pairs present by an
//                  LDS bitmap over the span whose prefix popcounts ARE the output order, counts by LDS atomics on the rank.
//   k_pf_write     the per-item results (cell-in-span << 16 | count) -> COO [row | col | val] at the scanned offsets.
//
// Anything this cannot place (a single (row, cell) with more than CAP keys, > 2^32 keys, 128-bit keys, a cell field too
// wide for the span) returns PF_FALLBACK and the caller takes the radix-sort path, which handles every input.
//
// Traffic at configs[2] (n = 380 M keys = 3.04 GB, 98 M non-zeros): hist 3.0 + part 6.1 + level 2 (~40 % of the keys)
// 1.2 + 2.4 + bucket 3.0 + 0.4 + write 0.4 + 1.2 = 17.7 GB (radix path: ~39 GB).  DESIGN.md section 3.2.
#pragma once

constexpr int PF_FALLBACK = 1;
#ifndef XCK_PF_C
#define XCK_PF_C 1024
#endif
constexpr int PF_C_MAX = XCK_PF_C;                     // page size C (keys); a work item holds at most CAP = 2C keys
constexpr int PF_CAP_MAX = 2 * PF_C_MAX;
constexpr int PF_SLOTS = 4 * PF_C_MAX;                 // LDS set of k_pf_bucket: 32 KB at C = 1024, load <= 0.5
#ifndef XCK_PF_THREADS
#define XCK_PF_THREADS 512
#endif
constexpr int PF_THREADS = XCK_PF_THREADS, PF_KPT = PF_CAP_MAX / PF_THREADS;
constexpr int PF_SB_MAX = 16, PF_BM_WORDS_MAX = (1 << PF_SB_MAX) / 32;   // bitmap over the span: at most 65536 (row, cell) pairs
constexpr int PT_THREADS = 512, PT_KPT = 16, PT_CHUNK = PT_THREADS * PT_KPT;
constexpr int PT_TAB_LG = 11, PT_TAB = 1 << PT_TAB_LG; // LDS aggregation table of the histogram / partition kernels
constexpr int PT_WIN = 2048;                           // keys per LDS output window of the partition kernel
constexpr uint32_t PF_BIG = 0x80000000u;               // WorkItem.span: the item is a big z (its keys are handled by level 2)

struct PartGeom {
    int ubits, cbits;                                  // key layout
    int sb;                                            // span bits: a work item stays inside one aligned block of 2^sb (row, cell) pairs
    int lgC;                                           // page size C = 1 << lgC
    uint32_t n_cells;
    const uint32_t* rowtab;                            // per row: (first level-1 cell of the row << 5) | sg;  z = base + (cell >> sg)
};
struct WorkItem { uint32_t off, span, aux, pad; };     // first key, (row, cell) >> sb of every key in it, level 1: big index / level 2: parent big z,
                                                       // level 2: (row, cell) & mask of the item's first sub-cell | PF_CONT
constexpr uint32_t PF_CONT = 0x80000000u;              // WorkItem.pad: that (row, cell) already has keys in earlier items of the same big z

// where the chunks of a histogram / partition launch come from
struct ShardChunks { unsigned long long cap; uint32_t cnt[NSHARD]; uint32_t chunk0[NSHARD + 1]; };   // level 1: the 16 shard slices of the hit buffer
struct BigChunks { const uint32_t* off; const uint32_t* cnt; const uint32_t* chunk0; const uint32_t* z2base; const uint32_t* sg; const uint32_t* eb; uint32_t n_big; };   // level 2: the big z of the level-1 output

#define PF_GADD(p, v) atomicAdd(p, v)
__device__ __forceinline__ uint32_t pf_copy_of_block(int pl) { return blockIdx.x & ((1u << pl) - 1u); }
struct AggTab { uint32_t tag[PT_TAB]; uint32_t cnt[PT_TAB]; uint32_t base[PT_TAB]; };
__device__ __forceinline__ int agg_find(AggTab& t, uint32_t z) {
    uint32_t h = (z * 0x9E3779B1u) >> (32 - PT_TAB_LG);
#pragma unroll 1
    for (int p = 0; p < 24; p++) {
        // a plain read first: nearly every key of a chunk finds its cell already in the table (thousands of keys, a few hundred cells),
        // and LDS atomics - one lane at a time under conflicts - were the longest phase of these kernels (45 k cycles per 8192-key chunk)
        const uint32_t cur = t.tag[h];
        if (cur == z) return (int)h;
        if (cur == 0xffffffffu) { const uint32_t prev = atomicCAS(&t.tag[h], 0xffffffffu, z); if (prev == 0xffffffffu || prev == z) return (int)h; }
        h = (h + 1) & (PT_TAB - 1);
    }
    return -1;                                         // table crowded (unsorted input): the key is handled on its own
}

// Barrier that waits for this wave's LDS traffic only: __syncthreads() also waits for every global load in flight (vmcnt(0)), which
// would turn the bucket kernel's prefetch of the next item's keys into a stall at the first barrier.
#define PF_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
template <int THREADS, bool LDS_ONLY = false>
__device__ __forceinline__ uint32_t block_excl_scan_t(uint32_t v, uint32_t* s_wave, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
    if (lane == 63) s_wave[wave] = inc;
    if (LDS_ONLY) PF_LDS_BARRIER(); else __syncthreads();
    uint32_t base = 0; total = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; w++) { const uint32_t t = s_wave[w]; if (w < wave) base += t; total += t; }
    if (LDS_ONLY) PF_LDS_BARRIER(); else __syncthreads();
    return base + inc - v;
}

// ---- device-wide exclusive scan of uint32 (in place), three launches --------------------------------------------------
constexpr int SC_T = 256, SC_I = 16, SC_TILE = SC_T * SC_I;
__global__ __launch_bounds__(SC_T) void k_scan_reduce(const uint32_t* __restrict__ in, size_t n, uint32_t* __restrict__ bsum) {
    __shared__ uint32_t s_w[SC_T / 64];
    const size_t base = (size_t)blockIdx.x * SC_TILE;
    uint32_t s = 0;
#pragma unroll
    for (int t = 0; t < SC_I; t++) { const size_t i = base + (size_t)t * SC_T + threadIdx.x; if (i < n) s += in[i]; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(1024) void k_scan_top(uint32_t* __restrict__ bsum, size_t nb, uint32_t* __restrict__ total_out) {
    __shared__ uint32_t s_w[16];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (size_t b0 = 0; b0 < nb; b0 += 1024) {
        const size_t i = b0 + threadIdx.x;
        const uint32_t v = i < nb ? bsum[i] : 0u;
        uint32_t inc = v;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        uint32_t wb = 0, tot = 0;
#pragma unroll
        for (int x = 0; x < 16; x++) { if (x < w) wb += s_w[x]; tot += s_w[x]; }
        const uint32_t carry = s_carry;
        if (i < nb) bsum[i] = carry + wb + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total_out) *total_out = s_carry;
}
__global__ __launch_bounds__(SC_T) void k_scan_apply(uint32_t* __restrict__ data, size_t n, const uint32_t* __restrict__ bsum) {
    __shared__ uint32_t s_w[SC_T / 64];
    const size_t i0 = (size_t)blockIdx.x * SC_TILE + (size_t)threadIdx.x * SC_I;       // blocked: a thread owns SC_I consecutive elements
    uint32_t v[SC_I], s = 0;
#pragma unroll
    for (int t = 0; t < SC_I; t++) { v[t] = i0 + t < n ? data[i0 + t] : 0u; s += v[t]; }
    uint32_t total;
    uint32_t run = bsum[blockIdx.x] + block_excl_scan_t<SC_T>(s, s_w, total);
#pragma unroll
    for (int t = 0; t < SC_I; t++) { if (i0 + t < n) data[i0 + t] = run; run += v[t]; }
}

// ---- histogram / partition over the level cells ------------------------------------------------------------------------
// chunk of a block: level 1 = PT_CHUNK consecutive keys of one shard slice, level 2 = PT_CHUNK consecutive keys of one big z
struct ChunkLoc { unsigned long long base; uint32_t n, zbase, sub_mask; int eb, tb; };
__device__ __forceinline__ ChunkLoc pf_locate1(const ShardChunks& sc, uint32_t chunk, uint32_t chunk_keys = PT_CHUNK) {
    int sh = 0;
#pragma unroll
    for (int q = 1; q < NSHARD; q++) sh += (chunk >= sc.chunk0[q]) ? 1 : 0;
    const uint32_t c = chunk - sc.chunk0[sh];
    ChunkLoc L; L.base = (unsigned long long)sh * sc.cap + (unsigned long long)c * chunk_keys;
    L.n = min(chunk_keys, sc.cnt[sh] - c * chunk_keys); L.zbase = 0; L.sub_mask = 0; L.eb = 0; L.tb = 0;
    return L;
}
__device__ __forceinline__ ChunkLoc pf_locate2(const BigChunks& bc, uint32_t chunk, uint32_t* s_b) {
    if (threadIdx.x == 0) {                            // the big z that owns the chunk: last b with chunk0[b] <= chunk
        uint32_t lo = 0, hi = bc.n_big;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (bc.chunk0[mid] <= chunk) lo = mid; else hi = mid; }
        *s_b = lo;
    }
    __syncthreads();
    const uint32_t b = *s_b;
    const uint32_t c = chunk - bc.chunk0[b];
    ChunkLoc L; L.base = (unsigned long long)bc.off[b] + (unsigned long long)c * PT_CHUNK;
    L.n = min((uint32_t)PT_CHUNK, bc.cnt[b] - c * PT_CHUNK); L.zbase = bc.z2base[b]; L.sub_mask = (1u << bc.sg[b]) - 1u; L.eb = (int)(bc.eb[b] & 31u); L.tb = (int)(bc.eb[b] >> 5);
    return L;
}
// Level 1 through a per-block LDS cache of the row table (direct-mapped, row << 32 | entry): a chunk holds a handful of rows, and
// 16 table look-ups per thread, all lanes at once, kept the address unit busy for 21 k cycles per chunk (the phase stamps of
// k_pf_part<1>); now the first key of a thread goes to the table in HBM / L2 and fills the cache, the others mostly hit it.
constexpr int PT_RC = 256;
__device__ __forceinline__ uint32_t pf_cell1_cached(unsigned long long key, const PartGeom& g, unsigned long long* s_rc, bool use_cache) {
    const unsigned long long rc = key >> g.ubits;
    const uint32_t row = (uint32_t)(rc >> g.cbits);
    uint32_t t;
    const unsigned long long e = use_cache ? s_rc[row & (PT_RC - 1)] : ~0ull;
    if ((uint32_t)(e >> 32) == row) t = (uint32_t)e;
    else { t = g.rowtab[row]; s_rc[row & (PT_RC - 1)] = ((unsigned long long)row << 32) | t; }
    return (t >> 5) + (((uint32_t)rc & ((1u << g.cbits) - 1u)) >> (t & 31u));
}
template <int LEVEL>
__device__ __forceinline__ uint32_t pf_cell(unsigned long long key, const PartGeom& g, const ChunkLoc& L) {
    const unsigned long long rc = key >> g.ubits;
    if (LEVEL == 0) return (uint32_t)(rc >> g.cbits);                     // the row itself (pileup hits: one cell per SNP)
    if (LEVEL == 1) {
        const uint32_t t = g.rowtab[(uint32_t)(rc >> g.cbits)];
        return (t >> 5) + (((uint32_t)rc & ((1u << g.cbits) - 1u)) >> (t & 31u));
    }
    // level 2: sub-cell = (low cell bits >> tb, eb bits of a UMI hash).  tb > 0: groups of 2^tb cells (a big z whose cells are
    // shallow); eb > 0 (then tb = 0): one (row, cell) deeper than a work item is cut into 2^eb parts, whose distinct-key counts add
    // up (the parts are disjoint key sets)
    const unsigned long long umi = key & ((1ull << g.ubits) - 1ull);
    const uint32_t h = (uint32_t)(umi ^ (umi >> 13) ^ (umi >> 27));
    return L.zbase + (((((uint32_t)rc & L.sub_mask) >> L.tb) << L.eb) | (h & ((1u << L.eb) - 1u)));
}

// keys per row on a SAMPLE of the stream: pieces of 64 keys, one out of `stride`, so that every fragment a join tile flushed (hundreds
// of keys) is met a few times - sampling whole chunks missed most rows of a few thousand keys (they live in 3 - 4 fragments) and
// grossly over-counted the others.  Decides the cell groups per row, nothing else.  A block covers PT_CHUNK * stride stream keys.
constexpr int PF_PIECE = 64;
__global__ __launch_bounds__(PT_THREADS) void k_pf_rowhist(const unsigned long long* __restrict__ keys, ShardChunks sc, uint32_t stride, int row_shift, uint32_t* __restrict__ rowcnt) {
    __shared__ AggTab t;
    for (int s = threadIdx.x; s < PT_TAB; s += PT_THREADS) { t.tag[s] = 0xffffffffu; t.cnt[s] = 0; }
    const ChunkLoc L = pf_locate1(sc, blockIdx.x, PT_CHUNK * stride);
    unsigned long long k[PT_KPT]; bool ok[PT_KPT];
#pragma unroll
    for (int q = 0; q < PT_KPT; q++) {
        const uint32_t i = q * PT_THREADS + threadIdx.x;
        const uint32_t j = (i / PF_PIECE) * (PF_PIECE * stride) + (i % PF_PIECE);      // piece i / 64 of the block's sample, key i % 64 in it
        ok[q] = j < L.n; k[q] = ok[q] ? keys[L.base + j] : 0ull;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PT_KPT; q++) {
        if (!ok[q]) continue;
        const uint32_t row = (uint32_t)(k[q] >> row_shift);
        const int slot = agg_find(t, row);
        if (slot >= 0) atomicAdd(&t.cnt[slot], 1u); else atomicAdd(&rowcnt[row], 1u);
    }
    __syncthreads();
    for (int s = threadIdx.x; s < PT_TAB; s += PT_THREADS) if (t.tag[s] != 0xffffffffu) atomicAdd(&rowcnt[t.tag[s]], t.cnt[s]);
}
// gsz[row] = cell groups of the row: the smallest 2^l (lg_min <= l <= lg_max) that brings the estimated keys per group to C / 2
__global__ void k_pf_rowplan(const uint32_t* __restrict__ rowcnt, uint32_t n_rows, uint32_t stride, int lg_min, int lg_max, int lgC, uint32_t* __restrict__ gsz) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_rows) return;
    if (r == n_rows) { gsz[r] = 0; return; }
    const unsigned long long est = (unsigned long long)rowcnt[r] * stride;
    int l = lg_min;
    while (l < lg_max && (est >> l) > (1ull << lgC) / 2) l++;
    gsz[r] = 1u << l;
}
// zb = exclusive scan of gsz (n_rows + 1 entries): rowtab[row] = zb << 5 | sg, zrow[z] = row of cell z
__global__ void k_pf_rowtab(const uint32_t* __restrict__ zb, uint32_t n_rows, int cbits, uint32_t* __restrict__ rowtab, uint32_t* __restrict__ zrow) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const uint32_t base = zb[r], G = zb[r + 1] - base;
    const int lg = 31 - __builtin_clz(G);
    rowtab[r] = (base << 5) | (uint32_t)(cbits - lg);
    if (zrow) for (uint32_t q = 0; q < G; q++) zrow[base + q] = r;
}

// The keys of a block.  Level 2: PT_CHUNK consecutive keys of one big z.  Level 0 / 1: the SAME 512-key segment of each of the 16
// shard slices (thread t, key q = slice q, position block * 512 + t): the slices are filled round-robin by consecutive join tiles, so
// equal positions hold keys of the same stretch of the file - a block then meets ALL keys of a narrow position range (a gene with
// 50 k keys used to be 16 runs of 3 k keys in 16 different blocks: 16 x the (block, cell) pairs, i.e. atomics, and runs 16 x
// shorter).  ok = bit q set when key q exists.
static_assert(PT_KPT == NSHARD, "one key per shard slice and thread");
template <int LEVEL>
__device__ __forceinline__ uint32_t pf_load_keys(const unsigned long long* __restrict__ keys, const ShardChunks& sc, const ChunkLoc& L, unsigned long long (&k)[PT_KPT]) {
    uint32_t ok = 0;
#pragma unroll
    for (int q = 0; q < PT_KPT; q++) {
        k[q] = 0ull;
        if (LEVEL == 2) { const uint32_t i = q * PT_THREADS + threadIdx.x; if (i < L.n) { k[q] = keys[L.base + i]; ok |= 1u << q; } }
        else { const uint32_t i = blockIdx.x * PT_THREADS + threadIdx.x; if (i < sc.cnt[q]) { k[q] = keys[(unsigned long long)q * sc.cap + i]; ok |= 1u << q; } }
    }
    return ok;
}
// where key number src (= q * PT_THREADS + thread) of the block came from (the pileup's values sit at the same place of their own array)
template <int LEVEL>
__device__ __forceinline__ unsigned long long pf_src_index(const ShardChunks& sc, const ChunkLoc& L, uint32_t src) {
    if (LEVEL == 2) return L.base + src;
    return (unsigned long long)(src / PT_THREADS) * sc.cap + (unsigned long long)blockIdx.x * PT_THREADS + (src % PT_THREADS);
}

// Level 1 keeps 2^pl copies of every counter / cursor (copy = block index mod 2^pl; the copies of a cell are adjacent, so the scan
// lays a cell's keys out copy after copy): the chunks of a hot gene - thousands in flight - otherwise queue on the same ~150
// addresses, and device-scope atomics on one address serialise at well under 2 per microsecond (first version: 5.4 ms for the
// histogram, 5.9 ms for the partition; the level-2 kernels, same code without the contention, ran at 3.9 TB/s).
template <int LEVEL>
__global__ __launch_bounds__(PT_THREADS) void k_pf_hist(const unsigned long long* __restrict__ keys, ShardChunks sc, BigChunks bc, PartGeom g, int pl, uint32_t* __restrict__ hist) {
    __shared__ AggTab t;
    __shared__ uint32_t s_b;
    for (int s = threadIdx.x; s < PT_TAB; s += PT_THREADS) { t.tag[s] = 0xffffffffu; t.cnt[s] = 0; }
    __shared__ unsigned long long s_rc[LEVEL == 1 ? PT_RC : 1];
    if (LEVEL == 1) for (int s = threadIdx.x; s < PT_RC; s += PT_THREADS) s_rc[s] = ~0ull;
    const uint32_t copy = pf_copy_of_block(pl);
    ChunkLoc L; L.base = 0; L.n = 0; L.zbase = 0; L.sub_mask = 0; L.eb = 0; L.tb = 0;
    if (LEVEL == 2) L = pf_locate2(bc, blockIdx.x, &s_b);
    unsigned long long k[PT_KPT];
    const uint32_t ok = pf_load_keys<LEVEL>(keys, sc, L, k);
    __syncthreads();
    uint32_t zq[PT_KPT];
#pragma unroll
    for (int q = 0; q < PT_KPT; q++) {
        if (!((ok >> q) & 1u)) { zq[q] = 0u; continue; }
        zq[q] = LEVEL == 1 ? pf_cell1_cached(k[q], g, s_rc, q > 0) : pf_cell<LEVEL>(k[q], g, L);
    }
#pragma unroll
    for (int q = 0; q < PT_KPT; q++) {
        if (!((ok >> q) & 1u)) continue;
        const uint32_t z = zq[q];
        const int slot = agg_find(t, z);
        if (slot >= 0) atomicAdd(&t.cnt[slot], 1u); else PF_GADD(&hist[(z << pl) | copy], 1u);
    }
    __syncthreads();
    for (int s = threadIdx.x; s < PT_TAB; s += PT_THREADS) if (t.tag[s] != 0xffffffffu) PF_GADD(&hist[(t.tag[s] << pl) | copy], t.cnt[s]);
}

// cursor[z << pl | copy] = first free index of that copy of z in the output (starts at the exclusive scan of the histogram)
template <int LEVEL>
__global__ __launch_bounds__(PT_THREADS) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_pf_part(const unsigned long long* __restrict__ keys, ShardChunks sc, BigChunks bc, PartGeom g, int pl,
                                                        uint32_t* __restrict__ cursor, unsigned long long* __restrict__ out,
                                                        const uint64_t* __restrict__ vals = nullptr, uint64_t* __restrict__ out_vals = nullptr) {
    __shared__ AggTab t;
    __shared__ uint32_t s_b;
    for (int s = threadIdx.x; s < PT_TAB; s += PT_THREADS) { t.tag[s] = 0xffffffffu; t.cnt[s] = 0; }
    __shared__ unsigned long long s_rc[LEVEL == 1 ? PT_RC : 1];
    if (LEVEL == 1) for (int s = threadIdx.x; s < PT_RC; s += PT_THREADS) s_rc[s] = ~0ull;
    const uint32_t copy = pf_copy_of_block(pl);
    ChunkLoc L; L.base = 0; L.n = 0; L.zbase = 0; L.sub_mask = 0; L.eb = 0; L.tb = 0;
    if (LEVEL == 2) L = pf_locate2(bc, blockIdx.x, &s_b);
    unsigned long long k[PT_KPT];
    const uint32_t ok = pf_load_keys<LEVEL>(keys, sc, L, k);
    __syncthreads();
    uint32_t sr[PT_KPT];                                                  // first the key's cell, then slot << 16 | rank in the slot's run; ~0 = not in the table
    static_assert(PT_CHUNK <= 65536 && PT_TAB <= 32768, "slot and rank share a word");
#pragma unroll
    for (int q = 0; q < PT_KPT; q++) {
        if (!((ok >> q) & 1u)) { sr[q] = 0xffffffffu; continue; }
        sr[q] = LEVEL == 1 ? pf_cell1_cached(k[q], g, s_rc, q > 0) : pf_cell<LEVEL>(k[q], g, L);
    }
#pragma unroll
    for (int q = 0; q < PT_KPT; q++) {
        if (!((ok >> q) & 1u)) continue;
        const int slot = agg_find(t, sr[q]);
        sr[q] = slot >= 0 ? ((uint32_t)slot << 16) | atomicAdd(&t.cnt[slot], 1u) : 0xffffffffu;
    }
    __syncthreads();
    // run bases: one returning atomic per (chunk, cell); the runs' places inside the chunk by a block scan of the slot counts
    __shared__ uint32_t s_wave[PT_THREADS / 64];
    constexpr int SPT = PT_TAB / PT_THREADS;                              // slots per thread (blocked)
    uint32_t c4[SPT], g4[SPT], sum = 0;
#pragma unroll
    for (int i = 0; i < SPT; i++) {
        const int sl = threadIdx.x * SPT + i;
        c4[i] = t.tag[sl] != 0xffffffffu ? t.cnt[sl] : 0u; g4[i] = 0; sum += c4[i];
        if (c4[i])
            g4[i] = PF_GADD(&cursor[(t.tag[sl] << pl) | copy], c4[i]);
    }
    uint32_t n_tab;
    uint32_t run = block_excl_scan_t<PT_THREADS>(sum, s_wave, n_tab);      // (its barriers also end the reads of t.cnt)
#pragma unroll
    for (int i = 0; i < SPT; i++) { const int sl = threadIdx.x * SPT + i; t.cnt[sl] = run; t.base[sl] = g4[i] - run; run += c4[i]; }   // cnt := place in the chunk, base := destination - place
    __syncthreads();
    // the keys leave through an LDS window in run order: consecutive lanes then store to consecutive addresses (a scattered 8-byte
    // store costs the address unit a cycle per lane: 55 k cycles per chunk when every key went out on its own)
    __shared__ unsigned long long s_key[PT_WIN];
    __shared__ uint16_t s_slot[PT_WIN];
    __shared__ uint16_t s_src[PT_WIN];
    for (uint32_t w0 = 0; w0 < n_tab; w0 += PT_WIN) {
#pragma unroll
        for (int q = 0; q < PT_KPT; q++) {
            if (sr[q] == 0xffffffffu) continue;
            const uint32_t pos = t.cnt[sr[q] >> 16] + (sr[q] & 0xffffu) - w0;
            if (pos < (uint32_t)PT_WIN) { s_key[pos] = k[q]; s_slot[pos] = (uint16_t)(sr[q] >> 16); if (vals) s_src[pos] = (uint16_t)(q * PT_THREADS + threadIdx.x); }
        }
        __syncthreads();
        const uint32_t m = min((uint32_t)PT_WIN, n_tab - w0);
        for (uint32_t i = threadIdx.x; i < m; i += PT_THREADS) {
            const uint32_t dst = t.base[s_slot[i]] + w0 + i;
            { out[dst] = s_key[i]; if (vals) out_vals[dst] = vals[pf_src_index<LEVEL>(sc, L, s_src[i])]; }   // (pileup hits: the value travels with its key)
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < PT_KPT; q++) {                                    // keys that found no place in the table (a chunk with too many cells)
        if (!((ok >> q) & 1u) || sr[q] != 0xffffffffu) continue;
        const uint32_t dst = PF_GADD(&cursor[(pf_cell<LEVEL>(k[q], g, L) << pl) | copy], 1u);
        out[dst] = k[q];
        if (vals) out_vals[dst] = vals[pf_src_index<LEVEL>(sc, L, q * PT_THREADS + threadIdx.x)];
    }
}

// ---- work items --------------------------------------------------------------------------------------------------------
// S = exclusive scan of the histogram, Z + 1 entries (S[Z] = number of keys).  Flags per cell z: fs = "starts a work item",
// level 1 also fb = "big" (more than CAP keys: goes through level 2), fc = its number of level-2 chunks, fz = its sub-cells.
// ctr: [3] += keys in big cells, [5] = 1 when some cell cannot be placed (PF_FALLBACK)
struct BigArrays { uint32_t* off; uint32_t* cnt; uint32_t* chunk0; uint32_t* wi; uint32_t* span; uint32_t* first2; uint32_t* z2base; uint32_t* sg; uint32_t* eb; uint32_t* rcl0; uint32_t* need; };
// Level-2 geometry of a big z with c keys over `cells` existing cells (of the 2^sg the z spans; the last group of a row is only
// partly filled).  depth = keys per cell if the cells were even.  Shallow cells: groups of 2^tb cells of about C / 2 keys; cells
// deeper than C / 2: tb = 0 and 2^eb UMI-hash parts per cell, at most 2^16 sub-cells per big z.  Returns eb | tb << 5;
// sub-cells = 2^(sg - tb + eb).  (A cell that is 4x deeper than its neighbours can still exceed a work item: k_pf_plan2 reports it.)
__device__ __forceinline__ uint32_t pf_sub_geom(uint32_t c, int sg, int lgC, uint32_t cells) {
    const uint32_t half = max(1u, (1u << lgC) / 2), depth = (c + cells - 1) / cells;
    if (depth <= half) { int tb = 0; while (tb < sg && (depth << (tb + 1)) <= half) tb++; return (uint32_t)tb << 5; }
    // (parts of a quarter page: the cells of a group are not even - one that is 4x deeper than the group's mean still fits an item)
    int eb = 0;
    while (sg + eb < 16 && ((depth + (1u << eb) - 1) >> eb) > max(1u, half / 2)) eb++;
    return (uint32_t)eb;
}
__device__ __forceinline__ uint32_t pf_sub_cells(uint32_t geom, int sg) { return 1u << (sg - (int)(geom >> 5) + (int)(geom & 31u)); }
__device__ __forceinline__ uint32_t pf_span1(uint32_t z, const PartGeom& g, const uint32_t* __restrict__ zrow, int* sg_out) {
    const uint32_t row = zrow[z], t = g.rowtab[row];
    const int sg = (int)(t & 31u);
    if (sg_out) *sg_out = sg;
    return (uint32_t)((((unsigned long long)row << g.cbits) | ((unsigned long long)(z - (t >> 5)) << sg)) >> g.sb);
}
// cells that exist in level-1 cell z (the row's last group may reach beyond n_cells)
__device__ __forceinline__ uint32_t pf_cells_in(uint32_t z, const PartGeom& g, const uint32_t* __restrict__ zrow) {
    const uint32_t t = g.rowtab[zrow[z]];
    const int sg = (int)(t & 31u);
    const unsigned long long lo = (unsigned long long)(z - (t >> 5)) << sg;
    if (lo >= g.n_cells) return 1u;
    return (uint32_t)min((unsigned long long)g.n_cells - lo, 1ull << sg);
}
__global__ void k_pf_plan1(const uint32_t* __restrict__ Sp, int pl, uint32_t Z, PartGeom g, const uint32_t* __restrict__ zrow,
                           uint32_t* __restrict__ fs, uint32_t* __restrict__ fb, uint32_t* __restrict__ fc, uint32_t* __restrict__ ctr) {
    const uint32_t z = blockIdx.x * blockDim.x + threadIdx.x;
    if (z > Z) return;
    if (z == Z) { fs[z] = 0; fb[z] = 0; fc[z] = 0; return; }                             // sentinel: the scans then end with the totals
    const uint32_t C = 1u << g.lgC, CAP = 2u << g.lgC;
    const uint32_t s = Sp[(size_t)z << pl], c = Sp[(size_t)(z + 1) << pl] - s, sp = z ? Sp[(size_t)(z - 1) << pl] : 0u, cp = z ? s - sp : 0u;
    int sg;
    const uint32_t span = pf_span1(z, g, zrow, &sg), spanp = z ? pf_span1(z - 1, g, zrow, nullptr) : 0u;
    const bool start = z == 0 || (s >> g.lgC) != (sp >> g.lgC) || c > C || cp > C || span != spanp;
    const bool big = c > CAP;
    fs[z] = start ? 1u : 0u; fb[z] = big ? 1u : 0u; fc[z] = big ? (c + PT_CHUNK - 1) / PT_CHUNK : 0u;
    if (big) atomicAdd(&ctr[3], c);
}
// id / bid / ch0 = exclusive scans of fs / fb / fc (Z + 1 entries each)
__global__ void k_pf_emit1(const uint32_t* __restrict__ Sp, int pl, uint32_t Z, PartGeom g, const uint32_t* __restrict__ zrow, const uint32_t* __restrict__ id, const uint32_t* __restrict__ bid,
                           const uint32_t* __restrict__ ch0, WorkItem* __restrict__ wi, BigArrays big) {
    const uint32_t z = blockIdx.x * blockDim.x + threadIdx.x;
    if (z >= Z) return;
    if (z == 0) {                                                                      // sentinels
        wi[id[Z]].off = Sp[(size_t)Z << pl]; wi[id[Z]].span = 0; wi[id[Z]].aux = 0xffffffffu;
        big.chunk0[bid[Z]] = ch0[Z]; big.off[bid[Z]] = Sp[(size_t)Z << pl]; big.cnt[bid[Z]] = 0; big.z2base[bid[Z]] = 0; big.sg[bid[Z]] = 0; big.eb[bid[Z]] = 0; big.rcl0[bid[Z]] = 0; big.need[bid[Z]] = 0;
    }
    if (id[z + 1] == id[z]) return;                                                    // not a start
    int sg;
    WorkItem w; w.off = Sp[(size_t)z << pl]; w.pad = 0; w.aux = 0xffffffffu; w.span = pf_span1(z, g, zrow, &sg);
    if (bid[z + 1] != bid[z]) {
        const uint32_t b = bid[z];
        big.off[b] = Sp[(size_t)z << pl]; big.cnt[b] = Sp[(size_t)(z + 1) << pl] - Sp[(size_t)z << pl]; big.chunk0[b] = ch0[z]; big.wi[b] = id[z]; big.span[b] = w.span; big.z2base[b] = 0; big.sg[b] = (uint32_t)sg; big.need[b] = 0;
        big.eb[b] = pf_sub_geom(Sp[(size_t)(z + 1) << pl] - Sp[(size_t)z << pl], sg, g.lgC, pf_cells_in(z, g, zrow));
        { const uint32_t row = zrow[z], t = g.rowtab[row];                             // (row, cell) of the big z's first cell, inside its span
          big.rcl0[b] = (uint32_t)((((unsigned long long)row << g.cbits) | ((unsigned long long)(z - (t >> 5)) << sg)) & ((1ull << g.sb) - 1ull)); }
        w.span |= PF_BIG; w.aux = b;
    }
    wi[id[z]] = w;
}
// level 2: cells z2 = z2base[b] + sub-cell; the big z of a cell by bisection
__device__ __forceinline__ uint32_t pf_big_of(const uint32_t* __restrict__ z2base, uint32_t n_big, uint32_t z2) {
    uint32_t lo = 0, hi = n_big;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (z2base[mid] <= z2) lo = mid; else hi = mid; }
    return lo;
}
// The geometry of a big z (pf_sub_geom) assumes its cells even.  Where they are not (well-based data: every cell has its own hot genes) a
// sub-cell comes out above CAP: its big z then asks for single cells cut into enough UMI-hash parts for the case that ALL keys of the
// sub-cell belong to one cell (big.need), and the host runs the level-2 histogram again (k_pf_big_refine; ctr[5] = 1).  A z that is
// already at 2^16 sub-cells cannot be refined: ctr[11] = 1 -> PF_FALLBACK.
__global__ void k_pf_big_sub(uint32_t n_big, BigArrays big) {                            // big.z2base[b] := sub-cells of b (scanned next)
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > n_big) return;
    big.z2base[b] = b < n_big ? pf_sub_cells(big.eb[b], (int)big.sg[b]) : 0u;
}
__global__ void k_pf_big_refine(uint32_t n_big, BigArrays big) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_big || !big.need[b]) return;
    big.eb[b] = big.need[b] - 1u;                                                       // tb = 0, eb = need - 1
    big.need[b] = 0;
}
__global__ void k_pf_plan2(const uint32_t* __restrict__ S, uint32_t Z, PartGeom g, BigArrays big, uint32_t n_big, uint32_t* __restrict__ fs, uint32_t* __restrict__ ctr) {
    const uint32_t* __restrict__ z2base = big.z2base;
    const uint32_t z = blockIdx.x * blockDim.x + threadIdx.x;
    if (z > Z) return;
    if (z == Z) { fs[z] = 0; return; }
    const uint32_t C = 1u << g.lgC, CAP = 2u << g.lgC;
    const uint32_t s = S[z], c = S[z + 1] - s, sp = z ? S[z - 1] : 0u, cp = z ? s - sp : 0u;
    const bool first = z2base[pf_big_of(z2base, n_big, z)] == z;                          // first sub-cell of a big z
    fs[z] = (z == 0 || first || (s >> g.lgC) != (sp >> g.lgC) || c > C || cp > C) ? 1u : 0u;
    if (c > CAP) {                                                                    // a sub-cell the geometry could not bring under CAP
        const uint32_t b = pf_big_of(z2base, n_big, z);
        const int sg = (int)big.sg[b], eb = (int)(big.eb[b] & 31u), tb = (int)(big.eb[b] >> 5);
        int more = 1; while ((c >> more) > max(1u, C / 4)) more++;                      // parts that bring c under a quarter page
        const int eb_new = (tb ? 0 : eb) + more;
        ctr[5] = 1u; atomicMax(&ctr[10], c);
        if (sg + eb_new > 16) ctr[11] = 1u; else atomicMax(&big.need[b], (uint32_t)eb_new + 1u);
    }
}
__global__ void k_pf_emit2(const uint32_t* __restrict__ S, uint32_t Z, const uint32_t* __restrict__ id, uint32_t n_big, WorkItem* __restrict__ wi, BigArrays big) {
    const uint32_t z = blockIdx.x * blockDim.x + threadIdx.x;
    if (z >= Z) return;
    if (z == 0) { wi[id[Z]].off = S[Z]; wi[id[Z]].span = 0; wi[id[Z]].aux = 0xffffffffu; big.first2[n_big] = id[Z]; }
    if (id[z + 1] == id[z]) return;
    const uint32_t b = pf_big_of(big.z2base, n_big, z);
    WorkItem w; w.off = S[z]; w.span = big.span[b]; w.aux = b;
    // the (row, cell) of the item's first sub-cell, and whether earlier parts of that cell (earlier items) hold keys: its count then
    // continues the entry those items produced (k_pf_bucket decides, k_pf_write adds)
    const uint32_t local = z - big.z2base[b], eb = big.eb[b] & 31u, tb = big.eb[b] >> 5, cell0 = big.z2base[b] + ((local >> eb) << eb);
    w.pad = (big.rcl0[b] + ((local >> eb) << tb)) | ((S[z] - S[cell0]) ? PF_CONT : 0u);
    if (big.z2base[b] == z) big.first2[b] = id[z];
    wi[id[z]] = w;
}

// ---- the work items: a block takes every gridDim.x-th item -------------------------------------------------------------------
// res[off + i] = (low sb bits of (row, cell)) << 16 | distinct keys, i < nnz, in (row, cell) order; nnz_out[w] = entries that open a
// new (row, cell); cont_out[w] (level 2) = 1 when the item's first entry continues the last entry of the items before it (a
// (row, cell) cut into UMI-hash parts that fell into several items).
// An item lives ~2 us in LDS but its keys take as long to arrive, and a block per item left the CUs waiting for loads (2.9 ms for
// the 424 k items of configs[2]); here the keys of the block's NEXT item (and the descriptor of the one after) are requested
// before the current item is processed.
__global__ __launch_bounds__(PF_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_pf_bucket(const unsigned long long* __restrict__ keys, const WorkItem* __restrict__ wi, uint32_t n_items, int ubits, int sb,
                                                          uint32_t* __restrict__ res, uint32_t* __restrict__ nnz_out, uint32_t* __restrict__ cont_out, uint32_t* __restrict__ ctr) {
    extern __shared__ unsigned long long pf_smem[];
    unsigned long long* set = pf_smem;                                     // phase 1: PF_SLOTS slots
    uint32_t* cnt32 = reinterpret_cast<uint32_t*>(pf_smem);                // later (the set is dead): PF_CAP_MAX 16-bit counters
    uint32_t* stage = cnt32 + PF_CAP_MAX / 2;                              //        the item's output, PF_CAP_MAX entries
    uint32_t* wpre = stage + PF_CAP_MAX;                                   //        prefix popcount of every bitmap word (<= 2048 words)
    uint32_t* bm = reinterpret_cast<uint32_t*>(pf_smem + PF_SLOTS);        // 2^sb bits
    static_assert(PF_CAP_MAX * 2 + PF_CAP_MAX * 4 + PF_BM_WORDS_MAX * 4 <= PF_SLOTS * 8, "the late arrays alias the set");
    __shared__ uint32_t s_wave[PF_THREADS / 64];
    const int tid = threadIdx.x;
    const uint32_t G = gridDim.x;
    const uint32_t bmask = (1u << sb) - 1u;
    const int n_words = 1 << (sb - 5);
    const int wpt = (n_words + PF_THREADS - 1) / PF_THREADS;               // bitmap words per thread (blocked)
    const int w0 = tid * wpt;
    uint32_t item = blockIdx.x;
    if (item >= n_items) return;
    auto usable = [](const WorkItem& w, uint32_t n) { return !(w.span & PF_BIG) && n > 0 && n <= (uint32_t)PF_CAP_MAX; };
    // (descriptors through VECTOR loads: a scalar load shares its counter with LDS, and every LDS wait of the item being processed
    // would then wait for the descriptor of the item after next to arrive from HBM)
    const uint32_t vz = __builtin_amdgcn_mbcnt_lo(0u, 0u);                 // 0 in every lane, unknown to the compiler
    auto load_item = [&](uint32_t i, WorkItem& w, uint32_t& n) {
        const uint4 a = *reinterpret_cast<const uint4*>(wi + i + vz); const uint32_t e = wi[i + 1 + vz].off;
        w.off = __builtin_amdgcn_readfirstlane(a.x); w.span = __builtin_amdgcn_readfirstlane(a.y); w.aux = __builtin_amdgcn_readfirstlane(a.z); w.pad = __builtin_amdgcn_readfirstlane(a.w);
        n = __builtin_amdgcn_readfirstlane(e) - w.off;
    };
    WorkItem cur; uint32_t n_cur; load_item(item, cur, n_cur);
    bool has_nxt = item + G < n_items;
    WorkItem nxt = cur; uint32_t n_nxt = 0;
    if (has_nxt) load_item(item + G, nxt, n_nxt);
    unsigned long long k[PF_KPT];
#pragma unroll
    for (int q = 0; q < PF_KPT; q++) { const uint32_t i = q * PF_THREADS + tid; k[q] = (usable(cur, n_cur) && i < n_cur) ? keys[cur.off + i] : ~0ull; }
    for (;;) {
        // ---- requests for the following items
        unsigned long long kn[PF_KPT];
        const bool ok_nxt = has_nxt && usable(nxt, n_nxt);
#pragma unroll
        for (int q = 0; q < PF_KPT; q++) { const uint32_t i = q * PF_THREADS + tid; kn[q] = (ok_nxt && i < n_nxt) ? keys[nxt.off + i] : ~0ull; }
        const bool has_nn = has_nxt && item + 2 * G < n_items;
        WorkItem nn = nxt; uint32_t n_nn = 0;
        uint4 nn_a = make_uint4(0, 0, 0, 0); uint32_t nn_e = 0;                // (consumed at the end of the iteration)
        if (has_nn) { nn_a = *reinterpret_cast<const uint4*>(wi + item + 2 * G + vz); nn_e = wi[item + 2 * G + 1 + vz].off; }
        // ---- the current item (every condition below is uniform over the block)
        const uint32_t off = cur.off, n = n_cur;
        if (cur.span & PF_BIG) { /* level 1: its keys went through level 2 (k_pf_bignnz fills nnz_out) */ }
        else if (n == 0 || n > (uint32_t)PF_CAP_MAX) {
            if (tid == 0) { if (n) ctr[7] = 1u; nnz_out[item] = 0; if (cont_out) cont_out[item] = 0; }   // n > CAP cannot happen (k_pf_plan); never index out of the LDS arrays
        } else {
            int lg_slots = 8; while ((1u << lg_slots) < 2 * n) lg_slots++;  // load <= 0.5; small items clear and probe a small table
            const uint32_t smask = (1u << lg_slots) - 1u;
            for (uint32_t s = tid; s <= smask; s += PF_THREADS) set[s] = ~0ull;
            for (int s = tid; s < n_words; s += PF_THREADS) bm[s] = 0u;
            PF_LDS_BARRIER();
            uint32_t win = 0;                                              // bit q: this thread's key q is the first occurrence of its (row, cell, umi)
#pragma unroll
            for (int q = 0; q < PF_KPT; q++) {
                if ((uint32_t)(q * PF_THREADS + tid) >= n) continue;
                const unsigned long long key = k[q];
                const uint32_t rcl = (uint32_t)(key >> ubits) & bmask;
                // (an item of a hot gene holds one or two cells: after the first keys every lane finds its bit set, and 64 lanes no
                // longer queue on one LDS word)
                if (!(bm[rcl >> 5] & (1u << (rcl & 31)))) atomicOr(&bm[rcl >> 5], 1u << (rcl & 31));
                uint32_t slot = set_slot<PF_SLOTS>(key) & smask;
                // (double hashing, as in the join's set: the wave waits for its longest probe sequence, and linear probing's clusters
                // make that one long at a load of up to 0.5; any odd stride visits every slot of the power-of-two table)
                const uint32_t stride = ((uint32_t)(key >> 7) ^ (uint32_t)(key >> 41)) | 1u;
                for (;;) {
                    const unsigned long long prev = atomicCAS(&set[slot], ~0ull, key);
                    if (prev == ~0ull) { win |= 1u << q; break; }
                    if (prev == key) break;
                    slot = (slot + stride) & smask;
                }
            }
            PF_LDS_BARRIER();                                               // set dead from here on: cnt32 / stage / wpre alias it
            uint32_t pc = 0;
            for (int i = 0; i < wpt; i++) if (w0 + i < n_words) pc += (uint32_t)__popc(bm[w0 + i]);
            uint32_t nnz;
            uint32_t run = block_excl_scan_t<PF_THREADS, true>(pc, s_wave, nnz);
            for (int i = 0; i < wpt; i++) if (w0 + i < n_words) { wpre[w0 + i] = run; run += (uint32_t)__popc(bm[w0 + i]); }
            for (uint32_t s = tid; s < (nnz + 1) / 2; s += PF_THREADS) cnt32[s] = 0u;
            PF_LDS_BARRIER();
            // count the first occurrences per (row, cell).  Same-address LDS atomics of a wave are served one lane at a time, and in
            // the items of a hot gene all 64 lanes meet on one or two counters: two rounds of "the lanes that share the first
            // active lane's rank add once, together" take those out; whatever is left (many cells: different counters) adds alone.
#pragma unroll
            for (int q = 0; q < PF_KPT; q++) {
                bool mine = (win >> q) & 1u;
                uint32_t rank = 0;
                if (mine) { const uint32_t rcl = (uint32_t)(k[q] >> ubits) & bmask;
                            rank = wpre[rcl >> 5] + (uint32_t)__popc(bm[rcl >> 5] & ((1u << (rcl & 31)) - 1u)); }
#pragma unroll
                for (int round = 0; round < 2; round++) {
                    const unsigned long long act = __ballot(mine);
                    if (!act) break;
                    const uint32_t r0 = __builtin_amdgcn_readlane(rank, (int)__builtin_ctzll(act));
                    const unsigned long long same = __ballot(mine && rank == r0);
                    if (mine && rank == r0) { if ((int)(tid & 63) == (int)__builtin_ctzll(same)) atomicAdd(&cnt32[r0 >> 1], (uint32_t)__popcll(same) << ((r0 & 1u) * 16)); mine = false; }
                }
                if (mine) atomicAdd(&cnt32[rank >> 1], 1u << ((rank & 1u) * 16));
            }
            PF_LDS_BARRIER();
            for (int i = 0; i < wpt; i++) {
                if (w0 + i >= n_words) break;
                uint32_t bits = bm[w0 + i];
                run = wpre[w0 + i];
                while (bits) {
                    const uint32_t b = (uint32_t)__builtin_ctz(bits); bits &= bits - 1u;
                    const uint32_t c = (cnt32[run >> 1] >> ((run & 1u) * 16)) & 0xffffu;
                    stage[run] = ((uint32_t)((w0 + i) * 32 + b) << 16) | c;
                    run++;
                }
            }
            PF_LDS_BARRIER();
            for (uint32_t i = tid; i < nnz; i += PF_THREADS) res[off + i] = stage[i];
            if (tid == 0) {
                const uint32_t cont = (cont_out && (cur.pad & PF_CONT) && (stage[0] >> 16) == (cur.pad & bmask)) ? 1u : 0u;
                nnz_out[item] = nnz - cont;
                if (cont_out) cont_out[item] = cont;
            }
            PF_LDS_BARRIER();                                               // stage is read: the next item may clear the set
        }
        if (!has_nxt) break;
        if (has_nn) { nn.off = __builtin_amdgcn_readfirstlane(nn_a.x); nn.span = __builtin_amdgcn_readfirstlane(nn_a.y); nn.aux = __builtin_amdgcn_readfirstlane(nn_a.z);
                      nn.pad = __builtin_amdgcn_readfirstlane(nn_a.w); n_nn = __builtin_amdgcn_readfirstlane(nn_e) - nn.off; }
        item += G; cur = nxt; n_cur = n_nxt; nxt = nn; n_nxt = n_nn; has_nxt = has_nn;
#pragma unroll
        for (int q = 0; q < PF_KPT; q++) k[q] = kn[q];
    }
}

// non-zeros of a big z = those of its level-2 work items; O2 = exclusive scan of the level-2 counts
__global__ void k_pf_bignnz(uint32_t n_big, BigArrays big, const uint32_t* __restrict__ O2, uint32_t* __restrict__ nnz1) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_big) return;
    nnz1[big.wi[b]] = O2[big.first2[b + 1]] - O2[big.first2[b]];
}

// LIST 1: work items of level 1 (the big ones are skipped), LIST 2: work items of the big z
template <int LIST>
__global__ __launch_bounds__(128) void k_pf_write(const WorkItem* __restrict__ wi, const uint32_t* __restrict__ res, const uint32_t* __restrict__ O1, const uint32_t* __restrict__ O2,
                                                  const uint32_t* __restrict__ cont2, BigArrays big, int cbits, int sb, unsigned long long total, int32_t* __restrict__ out) {
    // LIST 2: val[] is zero on entry and every count is ADDED: an item's first entry may continue the last entry of earlier items
    const WorkItem me = wi[blockIdx.x];
    uint32_t dst, n, cont = 0;
    if (LIST == 1) { if (me.span & PF_BIG) return; dst = O1[blockIdx.x]; n = O1[blockIdx.x + 1] - dst; }
    else { const uint32_t b = me.aux; cont = cont2[blockIdx.x]; dst = O1[big.wi[b]] + O2[blockIdx.x] - O2[big.first2[b]] - cont; n = O2[blockIdx.x + 1] - O2[blockIdx.x] + cont; }
    const unsigned long long hi = (unsigned long long)(me.span & ~PF_BIG) << sb;
    const unsigned long long cmask = (1ull << cbits) - 1;
    for (uint32_t i = threadIdx.x; i < n; i += 128) {
        const uint32_t v = res[me.off + i];
        const unsigned long long rc = hi | (v >> 16);
        if (LIST == 1) { out[dst + i] = (int32_t)(rc >> cbits); out[total + dst + i] = (int32_t)(rc & cmask); out[2 * total + dst + i] = (int32_t)(v & 0xffffu); }
        else {
            if (!(cont && i == 0)) { out[dst + i] = (int32_t)(rc >> cbits); out[total + dst + i] = (int32_t)(rc & cmask); }   // (the continued entry's row / col: written by the item that opened it)
            atomicAdd(&out[2 * total + dst + i], (int32_t)(v & 0xffffu));
        }
    }
}

__global__ void k_pf_publish(const uint32_t* __restrict__ src, unsigned long long* __restrict__ host_alias, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) host_alias[i] = src[i];
}

// ---- host ------------------------------------------------------------------------------------------------------------------
static int pf_scan(EngineImpl* im, uint32_t* data, size_t n, uint32_t* bsum, uint32_t* total_out) {       // in place, exclusive
    const size_t nb = (n + SC_TILE - 1) / SC_TILE;
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(SC_T), 0, im->s_comp, (const uint32_t*)data, n, bsum);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, im->s_comp, bsum, nb, total_out);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(SC_T), 0, im->s_comp, data, n, (const uint32_t*)bsum);
    HIP_TRY(hipGetLastError());
    return 0;
}
static inline int pf_bucket_lds(int sb) { return PF_SLOTS * 8 + (1 << sb) / 8; }

// basefc fold of the n keys in the shard slices of im->d_keys -> result matrix 0.  0 = done, PF_FALLBACK = take the radix path
// (the shard slices are then untouched), < 0 = error.
static int fold_partition(EngineImpl* im, KeyLayout<unsigned long long> kl, size_t n) {
    typedef unsigned long long K;
    if (n >= (size_t(1) << 32) - (size_t(1) << 20)) return PF_FALLBACK;                   // 32-bit offsets
    int lgC = 0;
    { const int c = im->eng->knobs.fold_c > 0 ? im->eng->knobs.fold_c : PF_C_MAX; while ((2 << lgC) <= c && (2 << lgC) <= PF_C_MAX) lgC++; }   // (test knob: small pages reach every path with small inputs)
    PartGeom g; g.ubits = kl.ubits; g.cbits = kl.cbits; g.lgC = lgC; g.n_cells = (uint32_t)im->n_cells;
    g.sb = std::min(PF_SB_MAX, std::max(kl.cbits, 11));                                    // 14 cell bits: one row per span, a 2 KB bitmap
    const int lg_min = std::max(0, kl.cbits - g.sb);                                      // a level-1 cell never straddles two spans
    // at most 2^6 cell groups per row: more groups keep more keys out of level 2 but leave the level-1 kernels more (block, cell) pairs -
    // atomics, short runs (A/B on one box at configs[2]: 8 -> 9.6 ms, 7 -> 8.9, 6 -> 8.7, 5 -> 8.9; profiles/r03_x_fold_variants_ab.log)
    // with at most 512 cells (well-based data: one BAM per cell) a row may get one group per cell: the cells of such data are far from
    // even (every cell has its own hot genes), and a level-1 cell that IS one (row, cell) knows its depth exactly
    const int lg_max = std::max(lg_min, std::min((im->eng->knobs.fold_lgg >= 0 ? im->eng->knobs.fold_lgg : kl.cbits <= 9 ? kl.cbits : 6), kl.cbits));
    const uint32_t n_rows = (uint32_t)im->n_regions;
    ShardChunks sc; sc.cap = im->hit_cap; sc.chunk0[0] = 0;
    for (int sh = 0; sh < NSHARD; sh++) { sc.cnt[sh] = (uint32_t)im->cur[sh]; sc.chunk0[sh + 1] = sc.chunk0[sh] + (uint32_t)((im->cur[sh] + PT_CHUNK - 1) / PT_CHUNK); }
    const unsigned n_chunks1 = sc.chunk0[NSHARD];
    size_t max_cur = 0; for (int sh = 0; sh < NSHARD; sh++) max_cur = std::max<size_t>(max_cur, im->cur[sh]);
    const unsigned n_blocks1 = (unsigned)((max_cur + PT_THREADS - 1) / PT_THREADS);       // level-1 blocks: segment b of every slice
    const uint32_t stride = (uint32_t)std::min<size_t>(16, std::max<size_t>(1, n_chunks1 / 4096));          // the row sample: one piece of 64 keys in `stride`, >= 16 M keys
    ShardChunks scs = sc;                                                                  // its blocks: PT_CHUNK * stride stream keys each
    for (int sh = 0; sh < NSHARD; sh++) scs.chunk0[sh + 1] = scs.chunk0[sh] + (uint32_t)((im->cur[sh] + (size_t)PT_CHUNK * stride - 1) / ((size_t)PT_CHUNK * stride));
    const unsigned n_sample = scs.chunk0[NSHARD];
    const size_t z_cap = ((size_t)n_rows << lg_min) + 4 * ((n + (size_t)NSHARD * stride * PT_CHUNK) >> lgC) + 64;   // sum of 2^l over the rows (k_pf_rowplan)
    if (z_cap > (size_t(1) << 24)) { if (im->eng->knobs.debug_timing) fprintf(stderr, "[xck] partition fold: level-1 cell bound %zu\n", z_cap); return PF_FALLBACK; }
    const size_t zs_cap = z_cap + 1, rs = (size_t)n_rows + 1;
    const size_t sb1 = (std::max(zs_cap, rs) + SC_TILE - 1) / SC_TILE + 8;
    int rc;
    // ---- workspace 1: level-1 output, row geometry, histogram / flags over the cells
    const int pl = std::max(0, std::min(im->eng->knobs.fold_copies_lg, 6));            // 2^pl copies of the level-1 counters / cursors
    const size_t ss_cap = (z_cap << pl) + 1;
    const size_t sbs = (ss_cap + SC_TILE - 1) / SC_TILE + 8;
    if ((rc = arena_begin(im, im->ws1, n * sizeof(K) + 3 * (rs * 4 + 256) + 5 * (zs_cap * 4 + 256) + ss_cap * 4 + (sb1 + sbs) * 4 + 64 * 4 + (1 << 16)))) return rc;
    K* A = im->ws1.get<K>(n);
    uint32_t* rowcnt = im->ws1.get<uint32_t>(rs); uint32_t* zb = im->ws1.get<uint32_t>(rs); uint32_t* rowtab = im->ws1.get<uint32_t>(rs);
    uint32_t* S1 = im->ws1.get<uint32_t>(ss_cap); uint32_t* fs = im->ws1.get<uint32_t>(zs_cap); uint32_t* fb = im->ws1.get<uint32_t>(zs_cap);
    uint32_t* fc = im->ws1.get<uint32_t>(zs_cap); uint32_t* zrow = im->ws1.get<uint32_t>(zs_cap);
    uint32_t* bsum = im->ws1.get<uint32_t>(sb1 + sbs); uint32_t* ctr = im->ws1.get<uint32_t>(64);
    if (!A || !rowcnt || !zb || !rowtab || !S1 || !fs || !fb || !fc || !zrow || !bsum || !ctr) { im->eng->err = "workspace exhausted (partition fold)"; return XCK_E_NOMEM; }
    g.rowtab = rowtab;
    unsigned long long* h_ctr = im->h_ctl + CTL_X0; unsigned long long* d_hctr = im->d_hctl + CTL_X0;   // (the k_expand words: unused in basefc mode)
    BigChunks bc0; memset(&bc0, 0, sizeof bc0);
    const unsigned gr = (unsigned)((rs + 255) / 256);
    // ---- row geometry from a sample of the stream
    HIP_TRY(hipMemsetAsync(rowcnt, 0, rs * 4, im->s_comp));
    HIP_TRY(hipMemsetAsync(ctr, 0, 64 * 4, im->s_comp));
    hipLaunchKernelGGL(k_pf_rowhist, dim3(n_sample), dim3(PT_THREADS), 0, im->s_comp, (const K*)im->d_keys, scs, stride, kl.ubits + kl.cbits, rowcnt);
    hipLaunchKernelGGL(k_pf_rowplan, dim3(gr), dim3(256), 0, im->s_comp, (const uint32_t*)rowcnt, n_rows, stride, lg_min, lg_max, lgC, zb);
    HIP_TRY(hipGetLastError());
    if ((rc = pf_scan(im, zb, rs, bsum, ctr + 8))) return rc;                              // ctr[8] = level-1 cells
    hipLaunchKernelGGL(k_pf_publish, dim3(1), dim3(64), 0, im->s_comp, (const uint32_t*)ctr, d_hctr, 16);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(im->s_comp));
    const uint32_t Z = (uint32_t)h_ctr[8];
    if ((size_t)Z > z_cap) { im->eng->err = "internal: level-1 cells exceed their bound"; return XCK_E_STATE; }
    const size_t zs = (size_t)Z + 1;
    const unsigned gz = (unsigned)((zs + 255) / 256);
    hipLaunchKernelGGL(k_pf_rowtab, dim3(gr), dim3(256), 0, im->s_comp, (const uint32_t*)zb, n_rows, kl.cbits, rowtab, zrow);
    // ---- level 1: histogram over the cells, work items
    const size_t ss = ((size_t)Z << pl) + 1;
    HIP_TRY(hipMemsetAsync(S1, 0, ss * 4, im->s_comp));
    hipLaunchKernelGGL((k_pf_hist<1>), dim3(n_blocks1), dim3(PT_THREADS), 0, im->s_comp, (const K*)im->d_keys, sc, bc0, g, pl, S1);
    HIP_TRY(hipGetLastError());
    if ((rc = pf_scan(im, S1, ss, bsum, nullptr))) return rc;
    hipLaunchKernelGGL(k_pf_plan1, dim3(gz), dim3(256), 0, im->s_comp, (const uint32_t*)S1, pl, Z, g, (const uint32_t*)zrow, fs, fb, fc, ctr);
    HIP_TRY(hipGetLastError());
    if ((rc = pf_scan(im, fs, zs, bsum, ctr + 0))) return rc;                              // ctr[0] = level-1 work items
    if ((rc = pf_scan(im, fb, zs, bsum, ctr + 1))) return rc;                              // ctr[1] = big cells
    if ((rc = pf_scan(im, fc, zs, bsum, ctr + 2))) return rc;                              // ctr[2] = level-2 chunks
    hipLaunchKernelGGL(k_pf_publish, dim3(1), dim3(64), 0, im->s_comp, (const uint32_t*)ctr, d_hctr, 16);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(im->s_comp));
    const size_t n_wi1 = h_ctr[0], n_big = h_ctr[1], n_chunks2 = h_ctr[2], n_bigkeys = h_ctr[3];
    // level-2 cells: about 4 per page of big keys; room for 16x that (refined geometries of uneven cells), beyond it the radix fold
    const size_t z2_cap = std::min<size_t>(size_t(1) << 26, 64 * (n_bigkeys >> lgC) + 64 * n_big + 65536);
    const size_t zs2 = z2_cap + 1;
    const size_t wi2_cap = n_big + 3 * (n_bigkeys >> lgC) + 8;
    const size_t sb2 = (std::max(std::max(zs2, wi2_cap + 1), n_wi1 + 1) + SC_TILE - 1) / SC_TILE + 8;
    // ---- workspace 2: everything whose size is known now
    if ((rc = arena_begin(im, im->ws2, (n_wi1 + 1) * (sizeof(WorkItem) + 4) + (n_big + 2) * 4 * 12 + n * 4 + 2 * (zs2 * 4 + 256) + (wi2_cap + 1) * (sizeof(WorkItem) + 8)
                                       + n_bigkeys * 4 + sb2 * 4 + n * 12 + (1 << 16)))) return rc;
    WorkItem* wi1 = im->ws2.get<WorkItem>(n_wi1 + 1); uint32_t* nnz1 = im->ws2.get<uint32_t>(n_wi1 + 1);
    BigArrays big; big.off = im->ws2.get<uint32_t>(n_big + 2); big.cnt = im->ws2.get<uint32_t>(n_big + 2); big.chunk0 = im->ws2.get<uint32_t>(n_big + 2);
    big.wi = im->ws2.get<uint32_t>(n_big + 2); big.span = im->ws2.get<uint32_t>(n_big + 2); big.first2 = im->ws2.get<uint32_t>(n_big + 2);
    big.z2base = im->ws2.get<uint32_t>(n_big + 2); big.sg = im->ws2.get<uint32_t>(n_big + 2); big.eb = im->ws2.get<uint32_t>(n_big + 2); big.rcl0 = im->ws2.get<uint32_t>(n_big + 2); big.need = im->ws2.get<uint32_t>(n_big + 2);
    uint32_t* res1 = im->ws2.get<uint32_t>(n);
    uint32_t* S2 = im->ws2.get<uint32_t>(zs2); uint32_t* fs2 = im->ws2.get<uint32_t>(zs2);
    WorkItem* wi2 = im->ws2.get<WorkItem>(wi2_cap + 1); uint32_t* nnz2 = im->ws2.get<uint32_t>(wi2_cap + 1); uint32_t* cont2 = im->ws2.get<uint32_t>(wi2_cap + 1);
    uint32_t* res2 = im->ws2.get<uint32_t>(n_bigkeys); uint32_t* bsum2 = im->ws2.get<uint32_t>(sb2);
    if (!wi1 || !nnz1 || !big.need || !res1 || !S2 || !fs2 || !wi2 || !nnz2 || !cont2 || !res2 || !bsum2) { im->eng->err = "workspace exhausted (partition fold, stage 2)"; return XCK_E_NOMEM; }
    hipLaunchKernelGGL(k_pf_emit1, dim3(gz), dim3(256), 0, im->s_comp, (const uint32_t*)S1, pl, Z, g, (const uint32_t*)zrow, (const uint32_t*)fs, (const uint32_t*)fb, (const uint32_t*)fc,
                       wi1, big);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL((k_pf_part<1>), dim3(n_blocks1), dim3(PT_THREADS), 0, im->s_comp, (const K*)im->d_keys, sc, bc0, g, pl, S1, A);   // (S1 is the cursor array from here on)
    HIP_TRY(hipGetLastError());
    const int lds = pf_bucket_lds(g.sb);
    const size_t bucket_grid = (size_t)std::max(1, im->eng->knobs.fold_bucket_blocks);     // resident blocks (4 per CU by LDS) x 2: the tail evens out
    size_t n_wi2 = 0;
    K* B = (K*)im->d_keys;                                                                 // level-2 output: the shard slices are dead once level 1 has moved the keys
    // The level-1 items do not depend on level 2: their bucket pass runs on the copy stream (idle until the copy-out) beside the
    // level-2 histogram / partition - two latency-bound kernels share the CUs better than either fills them (XCK_FOLD_OVERLAP=0: serial).
    const bool overlap = n_big && im->eng->knobs.fold_overlap != 0;
    if (overlap) {
        if (!im->ev_f1) { HIP_TRY(hipEventCreateWithFlags(&im->ev_f1, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&im->ev_f2, hipEventDisableTiming)); }
        HIP_TRY(hipEventRecord(im->ev_f1, im->s_comp));
        HIP_TRY(hipStreamWaitEvent(im->s_copy, im->ev_f1, 0));
        // (half of the CUs' wave slots - 2 blocks of 8 waves per CU: a persistent grid that fills the device would leave the level-2
        // kernels waiting for a slot until it retires)
        const size_t side_grid = (size_t)std::max(1, im->eng->knobs.fold_overlap_blocks);
        hipLaunchKernelGGL(k_pf_bucket, dim3((unsigned)std::min<size_t>(n_wi1, side_grid)), dim3(PF_THREADS), lds, im->s_copy, (const K*)A, (const WorkItem*)wi1, (uint32_t)n_wi1, kl.ubits, g.sb, res1, nnz1, (uint32_t*)nullptr, ctr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(im->ev_f2, im->s_copy));
    }
    size_t Z2 = 0;
    if (n_big) {
        BigChunks bc; bc.off = big.off; bc.cnt = big.cnt; bc.chunk0 = big.chunk0; bc.z2base = big.z2base; bc.sg = big.sg; bc.eb = big.eb; bc.n_big = (uint32_t)n_big;
        const unsigned gb = (unsigned)((n_big + 1 + 255) / 256);
        im->fold_refinements = 0;
        for (int attempt = 0;; attempt++) {
            // sub-cells of every big z from its geometry -> first level-2 cell of every big z
            hipLaunchKernelGGL(k_pf_big_sub, dim3(gb), dim3(256), 0, im->s_comp, (uint32_t)n_big, big);
            if ((rc = pf_scan(im, big.z2base, n_big + 1, bsum2, ctr + 9))) return rc;       // ctr[9] = level-2 cells
            hipLaunchKernelGGL(k_pf_publish, dim3(1), dim3(64), 0, im->s_comp, (const uint32_t*)ctr, d_hctr, 16);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(im->s_comp));
            Z2 = h_ctr[9];
            if (Z2 > z2_cap) { if (im->eng->knobs.debug_timing) fprintf(stderr, "[xck] partition fold: %zu level-2 cells (room for %zu)\n", Z2, z2_cap);
                               if (overlap) HIP_TRY(hipStreamSynchronize(im->s_copy));      // (the level-1 bucket pass still reads this workspace)
                               return PF_FALLBACK; }
            const size_t zs2n = Z2 + 1;
            const unsigned gz2 = (unsigned)((zs2n + 255) / 256);
            HIP_TRY(hipMemsetAsync(S2, 0, zs2n * 4, im->s_comp));
            HIP_TRY(hipMemsetAsync(ctr + 5, 0, 4, im->s_comp)); HIP_TRY(hipMemsetAsync(ctr + 10, 0, 8, im->s_comp));
            hipLaunchKernelGGL((k_pf_hist<2>), dim3((unsigned)n_chunks2), dim3(PT_THREADS), 0, im->s_comp, (const K*)A, sc, bc, g, 0, S2);
            HIP_TRY(hipGetLastError());
            if ((rc = pf_scan(im, S2, zs2n, bsum2, nullptr))) return rc;
            hipLaunchKernelGGL(k_pf_plan2, dim3(gz2), dim3(256), 0, im->s_comp, (const uint32_t*)S2, (uint32_t)Z2, g, big, (uint32_t)n_big, fs2, ctr);
            HIP_TRY(hipGetLastError());
            if ((rc = pf_scan(im, fs2, zs2n, bsum2, ctr + 4))) return rc;                   // ctr[4] = level-2 work items
            hipLaunchKernelGGL(k_pf_publish, dim3(1), dim3(64), 0, im->s_comp, (const uint32_t*)ctr, d_hctr, 16);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(im->s_comp));
            if (!h_ctr[5]) break;
            // (the shard slices still hold the keys: level 2 has not written yet)
            if (im->eng->knobs.debug_timing) fprintf(stderr, "[xck] partition fold: a level-2 sub-cell holds %llu keys (more than %d) - %s (n=%zu cells=%u big=%zu, %zu keys, %zu sub-cells)\n",
                                                    h_ctr[10], 2 << lgC, h_ctr[11] || attempt >= 3 ? "radix fold" : "finer geometry for its big cell", n, Z, n_big, n_bigkeys, Z2);
            if (h_ctr[11] || attempt >= 3) { if (overlap) HIP_TRY(hipStreamSynchronize(im->s_copy)); return PF_FALLBACK; }
            im->fold_refinements++;
            hipLaunchKernelGGL(k_pf_big_refine, dim3(gb), dim3(256), 0, im->s_comp, (uint32_t)n_big, big);
            HIP_TRY(hipGetLastError());
        }
        const unsigned gz2 = (unsigned)((Z2 + 1 + 255) / 256);
        n_wi2 = h_ctr[4];
        if (n_wi2 > wi2_cap) { im->eng->err = "internal: level-2 work items exceed their bound"; return XCK_E_STATE; }
        hipLaunchKernelGGL(k_pf_emit2, dim3(gz2), dim3(256), 0, im->s_comp, (const uint32_t*)S2, (uint32_t)Z2, (const uint32_t*)fs2, (uint32_t)n_big, wi2, big);
        hipLaunchKernelGGL((k_pf_part<2>), dim3((unsigned)n_chunks2), dim3(PT_THREADS), 0, im->s_comp, (const K*)A, sc, bc, g, 0, S2, B);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_pf_bucket, dim3((unsigned)std::min<size_t>(n_wi2, bucket_grid)), dim3(PF_THREADS), lds, im->s_comp, (const K*)B, (const WorkItem*)wi2, (uint32_t)n_wi2, kl.ubits, g.sb, res2, nnz2, cont2, ctr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemsetAsync(nnz2 + n_wi2, 0, 4, im->s_comp));
        if ((rc = pf_scan(im, nnz2, n_wi2 + 1, bsum2, nullptr))) return rc;                // nnz2 -> O2
        hipLaunchKernelGGL(k_pf_bignnz, dim3((unsigned)((n_big + 255) / 256)), dim3(256), 0, im->s_comp, (uint32_t)n_big, big, (const uint32_t*)nnz2, nnz1);
        HIP_TRY(hipGetLastError());
    }
    if (overlap) HIP_TRY(hipStreamWaitEvent(im->s_comp, im->ev_f2, 0));
    else {
        hipLaunchKernelGGL(k_pf_bucket, dim3((unsigned)std::min<size_t>(n_wi1, bucket_grid)), dim3(PF_THREADS), lds, im->s_comp, (const K*)A, (const WorkItem*)wi1, (uint32_t)n_wi1, kl.ubits, g.sb, res1, nnz1, (uint32_t*)nullptr, ctr);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemsetAsync(nnz1 + n_wi1, 0, 4, im->s_comp));
    if ((rc = pf_scan(im, nnz1, n_wi1 + 1, bsum2, ctr + 6))) return rc;                    // nnz1 -> O1, ctr[6] = non-zeros of the matrix
    hipLaunchKernelGGL(k_pf_publish, dim3(1), dim3(64), 0, im->s_comp, (const uint32_t*)ctr, d_hctr, 16);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(im->s_comp));
    if (h_ctr[7]) { im->eng->err = "internal: a work item of the partition fold exceeds its capacity"; return XCK_E_STATE; }
    const size_t total = h_ctr[6];
    if (im->eng->knobs.debug_timing) fprintf(stderr, "[xck] partition fold: n=%zu rows=%u cells=%u (sample stride %u) items=%zu big=%zu (%zu keys, %zu sub-cells, %zu items) nnz=%zu C=%d sb=%d\n",
                                            n, n_rows, Z, stride, n_wi1, n_big, n_bigkeys, Z2, n_wi2, total, 1 << lgC, g.sb);
    im->res_nnz[0] = total; im->d_res[0] = nullptr;
    if (!total) return 0;
    if ((rc = res_reserve(im, 0, total))) return rc;
    int32_t* d_o = im->ws2.get<int32_t>(total * 3);
    if (!d_o) { im->eng->err = "workspace exhausted (COO)"; return XCK_E_NOMEM; }
    if (n_wi2) HIP_TRY(hipMemsetAsync(d_o + 2 * total, 0, total * sizeof(int32_t), im->s_comp));      // level-2 counts are added
    hipLaunchKernelGGL((k_pf_write<1>), dim3((unsigned)n_wi1), dim3(128), 0, im->s_comp, (const WorkItem*)wi1, (const uint32_t*)res1, (const uint32_t*)nnz1, (const uint32_t*)nnz2, (const uint32_t*)nullptr, big,
                       kl.cbits, g.sb, (unsigned long long)total, d_o);
    if (n_wi2) hipLaunchKernelGGL((k_pf_write<2>), dim3((unsigned)n_wi2), dim3(128), 0, im->s_comp, (const WorkItem*)wi2, (const uint32_t*)res2, (const uint32_t*)nnz1, (const uint32_t*)nnz2, (const uint32_t*)cont2, big,
                                  kl.cbits, g.sb, (unsigned long long)total, d_o);
    HIP_TRY(hipGetLastError());
    return copy_out(im, 0, d_o, total);
}


// ---- pileup: the (key, value) hits sorted WITHOUT a radix sort -------------------------------------------------------------------
// The hits with a base leave k_join<pileup> in position order and a key's top field is the SNP's index in position order: the
// stream is almost sorted by row already.  The radix sort took 8 passes over the 25 M pairs of configs[2] (4.0 of the pileup
// fold's 11 ms); here the pairs are partitioned by row (one pass, the kernels above with z = row) into items of whole SNPs of at
// most 2C pairs, and every item is sorted in LDS (k_pf_radix_items below; the bitonic network k_pf_sort_items was the first form and
// stays selectable).  A (SNP, cell group) deeper than an item (UMI-less deep pileups) returns PF_FALLBACK: the radix sort handles it.  Reference semantics: first read per (SNP, cell, UMI) in fetch order,
// xcltk/baf/fc/mcount.py:109-127 - the order inside a key run is irrelevant to what follows (k_first_base takes the minimum).
constexpr int PS_CAP = PF_CAP_MAX, PS_THREADS = 256;
__global__ void k_pf_plan0(const uint32_t* __restrict__ S, uint32_t Z, int lgC, uint32_t* __restrict__ fs, uint32_t* __restrict__ ctr) {
    const uint32_t z = blockIdx.x * blockDim.x + threadIdx.x;
    if (z > Z) return;
    if (z == Z) { fs[z] = 0; return; }
    const uint32_t C = 1u << lgC, CAP = 2u << lgC;
    const uint32_t s = S[z], c = S[z + 1] - s, sp = z ? S[z - 1] : 0u, cp = z ? s - sp : 0u;
    fs[z] = (z == 0 || (s >> lgC) != (sp >> lgC) || c > C || cp > C) ? 1u : 0u;
    if (c > CAP) ctr[5] = 1u;
}
__global__ void k_pf_emit0(const uint32_t* __restrict__ S, uint32_t Z, const uint32_t* __restrict__ id, uint32_t* __restrict__ item_off) {
    const uint32_t z = blockIdx.x * blockDim.x + threadIdx.x;
    if (z >= Z) return;
    if (z == 0) item_off[id[Z]] = S[Z];
    if (id[z + 1] != id[z]) item_off[id[z]] = S[z];
}
__global__ __launch_bounds__(PS_THREADS) void k_pf_sort_items(unsigned long long* __restrict__ keys, uint64_t* __restrict__ vals, const uint32_t* __restrict__ item_off, uint32_t* __restrict__ ctr) {
    // Bitonic network on (key, value) in LDS.  Every wave owns a quarter of the array: the stages whose partner distance stays inside
    // a quarter are run by that wave alone, in lock step, without block barriers (63 of the 66 stages of a 2048-pair item); only
    // the three stages that pair elements of different quarters meet at a barrier.  (A barrier per stage: 67 us per item, 1.9 ms
    // for the 29 k items of configs[2].)
    __shared__ unsigned long long sk[PS_CAP];
    __shared__ unsigned long long sv[PS_CAP];
    const uint32_t off = item_off[blockIdx.x], n = item_off[blockIdx.x + 1] - off;
    if (n < 2) return;
    if (n > (uint32_t)PS_CAP) { if (threadIdx.x == 0) ctr[7] = 1u; return; }
    uint32_t N2 = 2; while (N2 < n) N2 <<= 1;
    for (uint32_t i = threadIdx.x; i < N2; i += PS_THREADS) { sk[i] = i < n ? keys[off + i] : ~0ull; sv[i] = i < n ? vals[off + i] : ~0ull; }
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t SEG = max(N2 / (PS_THREADS / 64), 128u);                // elements of a wave's segment (a power of two, >= 2 per lane)
    const bool wave_on = wave * SEG < N2;
    auto cmpx = [&](uint32_t p, uint32_t j, uint32_t k) {                 // pair number p of stage (k, j)
        const uint32_t i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), x = i | j;
        const unsigned long long ka = sk[i], kb = sk[x], va = sv[i], vb = sv[x];
        const bool gt = ka > kb || (ka == kb && va > vb);
        if (gt == ((i & k) == 0)) { sk[i] = kb; sk[x] = ka; sv[i] = vb; sv[x] = va; }
    };
    __syncthreads();
    for (uint32_t k = 2; k <= N2; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            if (j >= SEG) {                                               // partners in different segments: the whole block, between barriers
                __syncthreads();
                for (uint32_t p = threadIdx.x; p < N2 / 2; p += PS_THREADS) cmpx(p, j, k);
                __syncthreads();
            } else {
                if (wave_on) for (uint32_t p = lane; p < min(SEG, N2) / 2; p += 64) cmpx(wave * (SEG / 2) + p, j, k);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this wave's exchanges have landed before its next stage reads
            }
        }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += PS_THREADS) { keys[off + i] = sk[i]; vals[off + i] = sv[i]; }
}

// The same items sorted by an LSD radix sort in LDS (default; XCK_PILEUP_ITEM_SORT=bitonic keeps the network above).  512 threads, four
// pairs per thread held in REGISTERS between the passes; element e = wave * 256 + round * 64 + lane.  One pass per 8-bit window of key
// bits in which the item's keys differ at all (4 windows where an item is one hot SNP, ~6 where it is many cold ones): every wave ranks its 256 elements round by
// round with ballots (lanes of equal digit: eight ballots; rank = the wave's running count of the digit + lanes of the group below me),
// the 8 x 256 wave counts are scanned digit-major, and the pairs go through ONE LDS buffer to their new places and back into
// registers.  Stable, so equal keys keep their order and the padding (~0 keys) stays behind.  ~0.5 MB of LDS traffic per item
// against the network's 4.3 MB.
constexpr int RS_THREADS = 512, RS_WAVES = RS_THREADS / 64, RS_EPT = PS_CAP / RS_THREADS, RS_SEG = PS_CAP / RS_WAVES;
static_assert(RS_EPT * RS_THREADS == PS_CAP && RS_SEG == RS_EPT * 64, "item capacity = threads x elements per thread");
struct RsShared {
    unsigned long long sk[PS_CAP];
    unsigned long long sv[PS_CAP];                                        // (sk + sv = the 4096-slot key table of k_hap_items)
    uint16_t hist[RS_WAVES][256];                                         // (counts and places of an item: <= 2048.  16-bit: 36.9 KB per block, four blocks per CU)
    uint32_t s_wtot[RS_WAVES];
    unsigned long long s_diff;
};
// the item's pairs into registers (padding: ~0), OR of (key ^ first key) over the block -> which digits differ at all
template <bool HAS_VAL>
__device__ __forceinline__ unsigned long long rs_load(RsShared& sm, const unsigned long long* __restrict__ keys, const uint64_t* __restrict__ vals, uint32_t off, uint32_t n,
                                                      unsigned long long (&key)[RS_EPT], unsigned long long (&val)[RS_EPT]) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, e0 = wave * RS_SEG + lane;
    if (threadIdx.x == 0) sm.s_diff = 0;
    const unsigned long long first = keys[off];
    unsigned long long diff = 0;
#pragma unroll
    for (int r = 0; r < RS_EPT; r++) {
        const uint32_t e = e0 + r * 64;
        key[r] = ~0ull; val[r] = ~0ull;
        if (e < n) { key[r] = keys[off + e]; if (HAS_VAL) val[r] = vals[off + e]; diff |= key[r] ^ first; }
    }
    for (int d = 32; d; d >>= 1) diff |= __shfl_xor(diff, d);
    __syncthreads();
    if (lane == 0 && diff) atomicOr(&sm.s_diff, diff);
    __syncthreads();
    return sm.s_diff;
}
// one stable counting pass on the digit (key >> shift) & 255; the pairs come back in registers in their new order
template <bool HAS_VAL>
__device__ __forceinline__ void rs_pass(RsShared& sm, unsigned long long (&key)[RS_EPT], unsigned long long (&val)[RS_EPT], int shift, int n_rounds) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, e0 = wave * RS_SEG + lane;
#pragma unroll
    for (int q = 0; q < 256 / 64; q++) sm.hist[wave][q * 64 + lane] = 0;
    uint32_t rank[RS_EPT];
#pragma unroll
    for (int r = 0; r < RS_EPT; r++) {
        if (r < n_rounds) {
            const uint32_t dg = (uint32_t)(key[r] >> shift) & 0xffu;
            unsigned long long m = ~0ull;
#pragma unroll
            for (int b = 0; b < 8; b++) { const bool on = (dg >> b) & 1u; const unsigned long long bal = __ballot(on); m &= on ? bal : ~bal; }
            const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            const uint32_t prev = sm.hist[wave][dg];                      // (the lanes of a group read one word; LDS operations of a wave stay in order)
            if (below == 0) sm.hist[wave][dg] = (uint16_t)(prev + (uint32_t)__popcll(m));
            rank[r] = prev + below;
        }
    }
    __syncthreads();
    uint32_t c[RS_WAVES], tot = 0, inc = 0;
    if (threadIdx.x < 256) {
#pragma unroll
        for (int w = 0; w < RS_WAVES; w++) { c[w] = sm.hist[w][threadIdx.x]; tot += c[w]; }
        inc = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d, 64); if (lane >= (uint32_t)d) inc += t; }
        if (lane == 63) sm.s_wtot[wave] = inc;
    }
    __syncthreads();
    if (threadIdx.x < 256) {
        uint32_t base = inc - tot;
        for (uint32_t w = 0; w < wave; w++) base += sm.s_wtot[w];
#pragma unroll
        for (int w = 0; w < RS_WAVES; w++) { sm.hist[w][threadIdx.x] = (uint16_t)base; base += c[w]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_EPT; r++) {
        if (r < n_rounds) {
            const uint32_t pos = sm.hist[wave][(uint32_t)(key[r] >> shift) & 0xffu] + rank[r];
            sm.sk[pos] = key[r]; if (HAS_VAL) sm.sv[pos] = val[r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_EPT; r++) {
        const uint32_t e = e0 + r * 64;
        if (r < n_rounds) { key[r] = sm.sk[e]; if (HAS_VAL) val[r] = sm.sv[e]; }   // (places >= n hold this pass's padding: ~0 again)
    }
}
__global__ __launch_bounds__(RS_THREADS) void k_pf_radix_items(unsigned long long* __restrict__ keys, uint64_t* __restrict__ vals, const uint32_t* __restrict__ item_off, uint32_t* __restrict__ ctr) {
    __shared__ RsShared sm;
    const uint32_t off = item_off[blockIdx.x], n = item_off[blockIdx.x + 1] - off;
    if (n < 2) return;
    if (n > (uint32_t)PS_CAP) { if (threadIdx.x == 0) ctr[7] = 1u; return; }
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, e0 = wave * RS_SEG + lane;
    const int n_rounds = (int)min((uint32_t)RS_EPT, wave * RS_SEG >= n ? 0u : (n - wave * RS_SEG + 63) / 64);   // rounds of this wave that hold anything (wave-uniform)
    unsigned long long key[RS_EPT], val[RS_EPT];
    const unsigned long long diff = rs_load<true>(sm, keys, vals, off, n, key, val);
    if (!diff) return;                                                    // one key: sorted as it is
    // 8-bit windows that start at the lowest differing bit not yet sorted on (block-uniform): an item of one hot SNP differs in the
    // 24 UMI bits in use and in 4 cell bits six bits further up - 4 windows, where windows at multiples of 8 take 5
    for (unsigned long long rem = diff; rem; ) {
        const int shift = __builtin_ctzll(rem);
        rs_pass<true>(sm, key, val, shift, n_rounds);
        rem = shift >= 56 ? 0ull : rem & ~(0xffull << shift);
    }
#pragma unroll
    for (int r = 0; r < RS_EPT; r++) { const uint32_t e = e0 + r * 64; if (e < n) { keys[off + e] = key[r]; vals[off + e] = val[r]; } }
}

// Region-level hits of the pileup (key = region | cell | UMI, value = haplotype bits 1 REF / 2 ALT / 4 other): the items of the
// partition are NOT sorted by UMI.  Per item: (1) the radix passes above on the (region, cell) digits only (one pass where an item is
// one hot region, three where it is many cold ones) - the (region, cell) runs are now contiguous and in order; (2) run numbers from
// head flags; (3) every key into an LDS hash set (sized to the item, load <= 0.5) whose slot payload collects run number | OR of the
// key's haplotype bits - the set algebra of baf/fc/core.py:173-192 per molecule, however many SNPs it meets; (4) a sweep over the
// slots adds each molecule's class to its run's four packed counters (REF-hap, ALT-hap, either, other-only keys); (5) the run's
// head key and its four sums (one packed word) go to run_key / sums at [item offset + run number] - a staging area with holes (an item has fewer runs
// than keys), zero where nothing is written; k_hap_count / k_hap_scatter skip the holes.  Replaces the UMI digits of the sort,
// k_hap_class (+ _long), k_fold_heads and k_hap_sum of the sorted path.
// per staging entry: 4 x 16-bit sums in one word (zero = hole), the run's head key.  pack_shift >= 0: there are no values - the haplotype
// class (0 REF-hap, 1 ALT-hap, 2 other) rides in two unused bits of the key's UMI field, at pack_shift (k_expand put it there)
struct HapItemsOut { unsigned long long* sums; unsigned long long* run_key; int pack_shift; };
// what a finished partition sort leaves behind for look-ups by key: cell z of (row, cell) = (rowtab[row] >> 5) + (cell >> (rowtab[row] & 31)),
// its entries in the sorted output = [z ? end[z - 1] : 0, end[z]) (k_pf_part turned the scanned counts into end offsets).  Lives in the
// sort's scratch arena: valid until that arena is begun again.
struct PartIndex { const uint32_t* rowtab; const uint32_t* end; };
template <bool PACKED>
__global__ __launch_bounds__(RS_THREADS) void k_hap_items(const unsigned long long* __restrict__ keys, const uint64_t* __restrict__ vals, const uint32_t* __restrict__ item_off,
                                                          int ubits, HapItemsOut out, uint32_t* __restrict__ ctr) {
    __shared__ RsShared sm;
    __shared__ uint32_t pay[PS_CAP];                                      // 16 bits per slot of the 4096-slot table: run number << 3 | haplotype bits (0 = empty)
    __shared__ uint32_t s_wheads[RS_WAVES];
    const uint32_t off = item_off[blockIdx.x], n = item_off[blockIdx.x + 1] - off;
    if (n == 0) return;
    if (n > (uint32_t)PS_CAP) { if (threadIdx.x == 0) ctr[7] = 1u; return; }
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63, e0 = wave * RS_SEG + lane;
    const int n_rounds = (int)min((uint32_t)RS_EPT, wave * RS_SEG >= n ? 0u : (n - wave * RS_SEG + 63) / 64);
    unsigned long long key[RS_EPT], val[RS_EPT];
    const unsigned long long diff = rs_load<!PACKED>(sm, keys, vals, off, n, key, val);
    for (unsigned long long rem = (diff >> ubits) << ubits; rem; ) {      // windows from the lowest differing (region, cell) bit upwards
        const int shift = __builtin_ctzll(rem);
        rs_pass<!PACKED>(sm, key, val, shift, n_rounds);
        rem = shift >= 56 ? 0ull : rem & ~(0xffull << shift);
    }
    if (PACKED) {                                                         // the class leaves the key: from here on as with values
#pragma unroll
        for (int r = 0; r < RS_EPT; r++) { val[r] = 1ull << ((key[r] >> out.pack_shift) & 3ull); if (e0 + r * 64 < n) key[r] &= ~(3ull << out.pack_shift); }
    }
    // (2) heads of the (region, cell) runs, run numbers in element order
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_EPT; r++) { const uint32_t e = e0 + r * 64; if (e < n) sm.sk[e] = key[r]; }
    __syncthreads();
    uint32_t runid[RS_EPT], heads_w = 0;
    bool head[RS_EPT];
#pragma unroll
    for (int r = 0; r < RS_EPT; r++) {
        const uint32_t e = e0 + r * 64;
        head[r] = e < n && (e == 0 || (sm.sk[e - 1] >> ubits) != (key[r] >> ubits));
        const unsigned long long bal = __ballot(head[r]);
        runid[r] = heads_w + (uint32_t)__popcll(bal & ((2ull << lane) - 1ull));      // heads up to and including me, inside this wave
        heads_w += (uint32_t)__popcll(bal);
    }
    if (lane == 0) s_wheads[wave] = heads_w;
    __syncthreads();
    uint32_t wbase = 0, n_runs = 0;
#pragma unroll
    for (int w = 0; w < RS_WAVES; w++) { const uint32_t t = s_wheads[w]; if ((uint32_t)w < wave) wbase += t; n_runs += t; }
    int lg_slots = 8; while ((1u << lg_slots) < 2 * n) lg_slots++;        // <= 4096 = sk + sv
    const uint32_t smask = (1u << lg_slots) - 1u;
    unsigned long long* __restrict__ tab = sm.sk;
    for (uint32_t q = threadIdx.x; q <= smask; q += RS_THREADS) tab[q] = ~0ull;
    for (uint32_t q = threadIdx.x; q <= (smask >> 1); q += RS_THREADS) pay[q] = 0u;
#pragma unroll
    for (int r = 0; r < RS_EPT; r++) {
        runid[r] = wbase + runid[r] - 1u;
        if (head[r]) out.run_key[(size_t)off + runid[r]] = key[r];
    }
    __syncthreads();
    // (3) molecules: hash set on the whole key, payload OR (no key equals the empty word ~0: key_layout() sizes the row field so that it is never all ones)
#pragma unroll
    for (int r = 0; r < RS_EPT; r++) {
        const uint32_t e = e0 + r * 64;
        if (e >= n) continue;
        const uint32_t p = (runid[r] << 3) | ((uint32_t)val[r] & 7u);
        uint32_t slot = set_slot<PF_SLOTS>(key[r]) & smask;
        const uint32_t stride = ((uint32_t)(key[r] >> 7) ^ (uint32_t)(key[r] >> 41)) | 1u;     // double hashing (see k_pf_bucket)
        for (;;) {
            const unsigned long long prev = atomicCAS(&tab[slot], ~0ull, key[r]);
            if (prev == ~0ull || prev == key[r]) break;
            slot = (slot + stride) & smask;
        }
        atomicOr(&pay[slot >> 1], p << ((slot & 1u) * 16));
    }
    __syncthreads();
    // (4) classes -> run counters: 4 x 16 bits per run (an item holds at most 2048 keys), in the memory of the key table - the keys are
    // dead, the sweep reads the payloads only (44 KB of LDS per block instead of 60: three blocks per CU)
    unsigned long long* __restrict__ cnt = sm.sk;
    for (uint32_t q = threadIdx.x; q < n_runs; q += RS_THREADS) cnt[q] = 0ull;
    __syncthreads();
    auto add_class = [&](uint32_t p) {
        const uint32_t bits = p & 7u, run = (p >> 3) & 0x7ffu;
        const unsigned long long c = (unsigned long long)(bits & 1u) | ((unsigned long long)((bits >> 1) & 1u) << 16) | ((unsigned long long)((bits & 3u) ? 1u : 0u) << 32)
                                   | ((unsigned long long)((!(bits & 3u) && (bits & 4u)) ? 1u : 0u) << 48);
        atomicAdd(&cnt[run], c);
    };
    for (uint32_t q = threadIdx.x; q <= smask; q += RS_THREADS) {
        const uint32_t p = (pay[q >> 1] >> ((q & 1u) * 16)) & 0xffffu;
        if (p) add_class(p);
    }
    __syncthreads();
    // (5) the runs of this item
    for (uint32_t q = threadIdx.x; q < n_runs; q += RS_THREADS) {
        out.sums[(size_t)off + q] = cnt[q];
    }
}

// (key, value) pairs in 16 slices (slice sh = src[sh * cap .. + cnt[sh]); the slices filled round-robin by the blocks of a kernel that walks
// a position-ordered stream) -> out_keys / out_vals (n entries each) sorted by key; equal keys in any order.  Used for the pileup's hits
// with a base (rows = SNPs; slices = the join's shard slices) and for its region-level hits (rows = regions; slices = k_expand's).
// Cells as in the basefc fold: every row gets 2^l cell groups so that a group holds about C / 2 hits (a SNP in a hot gene is tens of
// thousands of reads deep); the groups of a row are in cell order, so cell order = key order.  0 = done, PF_FALLBACK = use the radix
// sort (the source slices are untouched either way), < 0 = error.  Scratch: `ar`, begun here when `own_arena`, else reserved by the
// caller (partition_sort_scratch()).  With `hap` the items are not sorted: k_hap_items classifies them in place (region-level hits).
struct PartSortSizes { size_t rs, z_cap, zs_cap, wi_cap, sb, bytes; };
static PartSortSizes partition_sort_sizes(size_t n, size_t n_rows, int lgC) {
    PartSortSizes q;
    q.rs = n_rows + 1;
    q.z_cap = n_rows + 4 * ((n + (size_t)NSHARD * 16 * PT_CHUNK) >> lgC) + 64;      // (16 = the largest sampling stride of the row histogram)
    q.zs_cap = q.z_cap + 1; q.wi_cap = 3 * (n >> lgC) + 8;
    q.sb = (std::max(std::max(q.zs_cap, q.rs), q.wi_cap + 1) + SC_TILE - 1) / SC_TILE + 8;
    q.bytes = 3 * (q.rs * 4 + 256) + 2 * (q.zs_cap * 4 + 256) + (q.wi_cap + 1) * 4 + q.sb * 4 + 64 * 4 + (1 << 16);
    return q;
}
static int partition_sort_lgC(const EngineImpl* im) { int lgC = 0; const int c = im->eng->knobs.fold_c > 0 ? im->eng->knobs.fold_c : PF_C_MAX; while ((2 << lgC) <= c && (2 << lgC) <= PF_C_MAX) lgC++; return lgC; }
static size_t partition_sort_scratch(const EngineImpl* im, size_t n, size_t n_rows) { return partition_sort_sizes(n, n_rows, partition_sort_lgC(im)).bytes; }
static int pileup_partition_sort(EngineImpl* im, Arena& ar, bool own_arena, KeyLayout<unsigned long long> kl, const unsigned long long* src_keys, const uint64_t* src_vals,
                                 size_t src_cap, const unsigned long long* src_cnt, uint32_t n_rows, size_t n, unsigned long long* out_keys, uint64_t* out_vals,
                                 const HapItemsOut* hap = nullptr, PartIndex* index = nullptr) {
    typedef unsigned long long K;
    if (n >= (size_t(1) << 32) - (size_t(1) << 20)) return PF_FALLBACK;
    const int lgC = partition_sort_lgC(im);
    PartGeom g; memset(&g, 0, sizeof g); g.ubits = kl.ubits; g.cbits = kl.cbits; g.lgC = lgC; g.sb = PF_SB_MAX; g.n_cells = (uint32_t)im->n_cells;
    // (there is no second level here: a SNP of a hot gene - 100 k hits at configs[2] - must come apart in the first one, so up to 2^10 groups per SNP)
    const int lg_max = std::max(0, std::min(im->eng->knobs.pileup_lgg, kl.cbits));
    ShardChunks sc; sc.cap = src_cap; sc.chunk0[0] = 0;
    for (int sh = 0; sh < NSHARD; sh++) { sc.cnt[sh] = (uint32_t)src_cnt[sh]; sc.chunk0[sh + 1] = sc.chunk0[sh] + (uint32_t)((src_cnt[sh] + PT_CHUNK - 1) / PT_CHUNK); }
    const unsigned n_chunks = sc.chunk0[NSHARD];
    size_t max_cur = 0; for (int sh = 0; sh < NSHARD; sh++) max_cur = std::max<size_t>(max_cur, src_cnt[sh]);
    const unsigned n_blocks1 = (unsigned)((max_cur + PT_THREADS - 1) / PT_THREADS);
    const uint32_t stride = (uint32_t)std::min<size_t>(16, std::max<size_t>(1, n_chunks / 4096));
    ShardChunks scs = sc;
    for (int sh = 0; sh < NSHARD; sh++) scs.chunk0[sh + 1] = scs.chunk0[sh] + (uint32_t)((src_cnt[sh] + (size_t)PT_CHUNK * stride - 1) / ((size_t)PT_CHUNK * stride));
    BigChunks bc0; memset(&bc0, 0, sizeof bc0);
    const PartSortSizes q = partition_sort_sizes(n, n_rows, lgC);
    const size_t rs = q.rs, z_cap = q.z_cap, zs_cap = q.zs_cap, wi_cap = q.wi_cap, sb = q.sb;
    if (z_cap > (size_t(1) << 26)) return PF_FALLBACK;
    int rc;
    if (own_arena && (rc = arena_begin(im, ar, q.bytes))) return rc;
    uint32_t* rowcnt = ar.get<uint32_t>(rs); uint32_t* zb = ar.get<uint32_t>(rs); uint32_t* rowtab = ar.get<uint32_t>(rs);
    uint32_t* S = ar.get<uint32_t>(zs_cap); uint32_t* fs = ar.get<uint32_t>(zs_cap); uint32_t* item_off = ar.get<uint32_t>(wi_cap + 1);
    uint32_t* bsum = ar.get<uint32_t>(sb); uint32_t* ctr = ar.get<uint32_t>(64);
    if (!rowcnt || !zb || !rowtab || !S || !fs || !item_off || !bsum || !ctr) { im->eng->err = "workspace exhausted (pileup partition)"; return XCK_E_NOMEM; }
    g.rowtab = rowtab;
    unsigned long long* h_ctr = im->h_ctl + CTL_X0; unsigned long long* d_hctr = im->d_hctl + CTL_X0;   // (the k_expand words: not in use yet)
    const unsigned gr = (unsigned)((rs + 255) / 256);
    HIP_TRY(hipMemsetAsync(rowcnt, 0, rs * 4, im->s_comp));
    HIP_TRY(hipMemsetAsync(ctr, 0, 64 * 4, im->s_comp));
    hipLaunchKernelGGL(k_pf_rowhist, dim3(scs.chunk0[NSHARD]), dim3(PT_THREADS), 0, im->s_comp, (const K*)src_keys, scs, stride, kl.ubits + kl.cbits, rowcnt);
    hipLaunchKernelGGL(k_pf_rowplan, dim3(gr), dim3(256), 0, im->s_comp, (const uint32_t*)rowcnt, n_rows, stride, 0, lg_max, lgC, zb);
    HIP_TRY(hipGetLastError());
    if ((rc = pf_scan(im, zb, rs, bsum, ctr + 8))) return rc;
    hipLaunchKernelGGL(k_pf_publish, dim3(1), dim3(64), 0, im->s_comp, (const uint32_t*)ctr, d_hctr, 16);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(im->s_comp));
    const uint32_t Z = (uint32_t)h_ctr[8];
    if ((size_t)Z > z_cap) { im->eng->err = "internal: pileup cells exceed their bound"; return XCK_E_STATE; }
    const size_t zs = (size_t)Z + 1;
    const unsigned gz = (unsigned)((zs + 255) / 256);
    hipLaunchKernelGGL(k_pf_rowtab, dim3(gr), dim3(256), 0, im->s_comp, (const uint32_t*)zb, n_rows, kl.cbits, rowtab, (uint32_t*)nullptr);
    HIP_TRY(hipMemsetAsync(S, 0, zs * 4, im->s_comp));
    hipLaunchKernelGGL((k_pf_hist<1>), dim3(n_blocks1), dim3(PT_THREADS), 0, im->s_comp, (const K*)src_keys, sc, bc0, g, 0, S);
    HIP_TRY(hipGetLastError());
    if ((rc = pf_scan(im, S, zs, bsum, nullptr))) return rc;
    hipLaunchKernelGGL(k_pf_plan0, dim3(gz), dim3(256), 0, im->s_comp, (const uint32_t*)S, Z, lgC, fs, ctr);
    HIP_TRY(hipGetLastError());
    if ((rc = pf_scan(im, fs, zs, bsum, ctr + 0))) return rc;
    hipLaunchKernelGGL(k_pf_publish, dim3(1), dim3(64), 0, im->s_comp, (const uint32_t*)ctr, d_hctr, 16);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(im->s_comp));
    if (h_ctr[5]) {                                                                       // a (SNP, cell group) with more hits than an item holds
        if (im->eng->knobs.debug_timing) fprintf(stderr, "[xck] pileup partition sort: a cell group exceeds an item (n=%zu cells=%u): radix sort\n", n, Z);
        return PF_FALLBACK;
    }
    const size_t n_items = h_ctr[0];
    if (n_items > wi_cap) { im->eng->err = "internal: pileup items exceed their bound"; return XCK_E_STATE; }
    hipLaunchKernelGGL(k_pf_emit0, dim3(gz), dim3(256), 0, im->s_comp, (const uint32_t*)S, Z, (const uint32_t*)fs, item_off);
    hipLaunchKernelGGL((k_pf_part<1>), dim3(n_blocks1), dim3(PT_THREADS), 0, im->s_comp, (const K*)src_keys, sc, bc0, g, 0, S, out_keys, src_vals, out_vals);
    const bool bitonic = im->eng->knobs.pileup_bitonic;
    if (hap && hap->pack_shift >= 0) hipLaunchKernelGGL((k_hap_items<true>), dim3((unsigned)n_items), dim3(RS_THREADS), 0, im->s_comp, (const unsigned long long*)out_keys, (const uint64_t*)nullptr, (const uint32_t*)item_off, kl.ubits, *hap, ctr);
    else if (hap) hipLaunchKernelGGL((k_hap_items<false>), dim3((unsigned)n_items), dim3(RS_THREADS), 0, im->s_comp, (const unsigned long long*)out_keys, (const uint64_t*)out_vals, (const uint32_t*)item_off, kl.ubits, *hap, ctr);
    else if (bitonic) hipLaunchKernelGGL(k_pf_sort_items, dim3((unsigned)n_items), dim3(PS_THREADS), 0, im->s_comp, out_keys, out_vals, (const uint32_t*)item_off, ctr);
    else hipLaunchKernelGGL(k_pf_radix_items, dim3((unsigned)n_items), dim3(RS_THREADS), 0, im->s_comp, out_keys, out_vals, (const uint32_t*)item_off, ctr);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_pf_publish, dim3(1), dim3(64), 0, im->s_comp, (const uint32_t*)ctr, d_hctr, 16);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(im->s_comp));
    if (h_ctr[7]) { im->eng->err = "internal: a pileup item exceeds its capacity"; return XCK_E_STATE; }
    if (index) { index->rowtab = rowtab; index->end = S; }
    if (im->eng->knobs.debug_timing) fprintf(stderr, "[xck] pileup partition sort: n=%zu snps=%u cells=%u items=%zu C=%d\n", n, n_rows, Z, n_items, 1 << lgC);
    return 0;
}
