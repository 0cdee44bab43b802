// engine.hip - MI355X (gfx950) counting engine: tables, HIP kernels, per-GPU pipeline.
//
// Replaces the reference's per-region / per-SNP fetch loops
//   xcltk/rdr/fc/core.py:96-178  (fc_features -> fc_fet1 -> check_read / include test / MCount)
//   xcltk/baf/fc/core.py:70-247  (fc_features -> fc_fet1 -> plp_snp -> MCount/SCount/UCount)
// with ONE streaming pass over coordinate-sorted record batches:
//
//   k_tile_meta : one thread per 1024-read tile: extent, first candidate region (bisection on the running maximum of the
//                 region ends) or staged SNP windows (48-byte record).
//   k_join      : one block per tile; CIGAR run and the next regions / SNPs staged in LDS.
//                 basefc: read x region interval join, region-major and wave-uniform (the 64 reads of a wave walk the same
//                 candidate list once) + CIGAR-walk include test, accepted keys de-duplicated in an LDS hash set.
//                 pileup: single-block reads walk the SNPs under them SNP-major; spliced / indel reads are set aside and
//                 processed together after the sweeps ((read, SNP) pairs dealt evenly over each wave's lanes); hits with
//                 a base and "gap records" (SNPs inside N / D gaps) leave in two streams.
//                 Fragments are appended through sharded cursors.
//   finish      : basefc: the partition fold of fold_partition.h (no sort); fallback: radix sort (rocPRIM) over the (row, cell) bits,
//                 k_fold_heads / k_fold_emit_unsorted (distinct UMIs of a run told apart by an LDS hash set) straight into COO.
//                 pileup: hits with a base sorted by row partition + LDS radix sort per item (fold_partition.h; fallback: rocPRIM),
//                 k_first_base (first read per key, SNP-mask filter laid out along the sorted stream; k_first_long for runs longer
//                 than 64), k_claim (gap records that hold a key earlier in fetch order), k_tally_rows, k_expand (per-SNP filters,
//                 SNP -> region fan-out), region-level hits partitioned again and classified per item by an LDS hash set
//                 (k_hap_items; fallback: sort + k_hap_class / k_hap_sum), k_hap_count / k_hap_scatter (AD / DP / OTH -> COO);
//                 128-bit keys: k_first_read and the sorted path.
//                 Copy-out on the copy stream (xck_finish_async).
//
// Integer / byte work only - HBM-bound, no MFMA.  See DESIGN.md for layouts, byte counts and measurements.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <chrono>
#include <type_traits>
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include "xck_internal.h"

namespace xck {

typedef unsigned __int128 u128;
#ifndef XCK_WS_SNP
#define XCK_WS_SNP 10           // 1 kb: the first probe lands within a SNP or two of the read (32 kb windows needed a binary search per read)
#endif
constexpr int WSS = XCK_WS_SNP;        // window shift of the SNP index (window -> first SNP)
constexpr int JOIN_BLOCK = 256;

// hipGetLastError() after a launch also returns (and clears) an error that some EARLIER, unchecked runtime call of this thread
// left behind; the launch sites clear it first, and XCK_DEBUG_TIMING reports what was there.
static inline void clear_stale_error(const char* where, bool report = false) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess && report) fprintf(stderr, "[xck] %s: cleared a stale HIP error left by an earlier call: %s [%d]\n", where, hipGetErrorString(e), (int)e);
}
#define HIP_TRY(expr)                                                                      \
    do { hipError_t e_ = (expr); if (e_ != hipSuccess) {                                   \
        char b_[512]; snprintf(b_, sizeof b_, "%s failed: %s [%d] (%s:%d)", #expr,         \
                               hipGetErrorString(e_), (int)e_, __FILE__, __LINE__);        \
        im->eng->err = b_; return XCK_E_DEVICE; } } while (0)


// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ bool op_aligned(uint32_t op) { return (0x181u >> op) & 1u; }   // M,=,X
__device__ __forceinline__ bool op_ref(uint32_t op)     { return (0x18Du >> op) & 1u; }   // M,D,N,=,X
__device__ __forceinline__ bool op_query(uint32_t op)   { return (0x193u >> op) & 1u; }   // M,I,S,=,X

template <class K> struct KeyLayout {
    int ubits, cbits;
    __host__ __device__ K make(uint32_t row, uint32_t cell, uint64_t umi) const {
        return (K(row) << (cbits + ubits)) | (K(cell) << ubits) | K(umi);
    }
    __host__ __device__ K rc(K k) const { return k >> ubits; }
    __host__ __device__ uint32_t row(K k) const { return uint32_t(k >> (cbits + ubits)); }
    __host__ __device__ uint32_t cell(K k) const { return uint32_t((k >> ubits) & ((K(1) << cbits) - 1)); }
    __host__ __device__ uint64_t umi(K k) const { return ubits >= 64 ? uint64_t(k) : uint64_t(k & ((K(1) << ubits) - 1)); }
};

struct ReadFilter {                      // check_read(), rdr/fc/core.py:46-62
    int32_t min_mapq, min_len;
    uint32_t incl_flag, excl_flag;
    int32_t no_orphan;
    int32_t frac_mode;                   // rdr/fc/core.py:160-165
    double  min_inc_frac;
    int32_t min_inc_len;
};

#define XCK_GLOBAL __attribute__((address_space(1)))
template <class T> __device__ __forceinline__ const XCK_GLOBAL T* as_global(const T* p) { return (const XCK_GLOBAL T*)p; }

// One queued record batch (device pointers) inside a fused launch.
struct BatchDesc {
    int32_t n, tile0;                       // reads, first tile of this batch in the fused grid
    const int32_t* pos; const uint16_t* flag; const uint8_t* mapq; const int32_t* cell;
    const uint64_t* umi; const uint32_t* cig_off; const uint32_t* cigar;
    const uint32_t* seq_off; const uint8_t* seq;
    uint64_t ordinal_base;
    int32_t reg_lo, reg_hi;                                 // regions of the batch's contig: [reg_lo, reg_hi) of the start-sorted arrays
    const int32_t* snp_win; int32_t n_swin; int32_t snp_end; // SNP window table of the batch's contig
};
constexpr int MAX_FUSE = 24;              // batches per fused launch (the table travels in the kernel arguments)

// Batch table, passed BY VALUE: it lives in the kernarg segment and is read with scalar loads
// through the constant cache instead of costing a dependent round trip to HBM per tile.
struct BatchTable { int32_t n_batches; int32_t n_tiles; BatchDesc desc[MAX_FUSE]; };

// Per-tile facts computed once by k_tile_meta (one thread per tile, all tiles in parallel) so that
// the join kernel's prologue is ONE load of this record plus ONE round of independent staging loads.
struct TileMeta {
    uint32_t c_lo, cg_n;                  // CIGAR words of the tile: [c_lo, c_lo + cg_n) are staged
    int32_t  w0, nw, e0, n_ent;           // basefc: staged regions [e0, e0 + n_ent) of the start-sorted arrays; pileup: staged SNP windows [w0, w0 + nw)
    int32_t  k0, nk;                      // pileup: staged SNPs [k0, k0+nk); basefc: k0 = position of the tile's first read
    int32_t  b, r0, r1, pad;              // batch index, first / one-past-last read of the tile
};

template <class K> struct JoinArgs {
    BatchTable bt;
    const TileMeta* meta;                 // [n_tiles]
    ReadFilter f;
    const int32_t* reg_s0; const int32_t* reg_e0; const int32_t* reg_row;   // regions per contig sorted by start (0-based half-open + output row)
    const int32_t* reg_pmax;                                                 // running maximum of reg_e0 inside the contig
    const int32_t* snp_p0;
    KeyLayout<K> kl;
    K* keys; uint64_t* vals; unsigned long long cap;   // cap = capacity of ONE shard
    K* nkeys; uint64_t* nvals;                          // pileup split mode: the stream of hits without a base (same shard capacity)
    unsigned long long* ctl;             // control block, see CTL_* below
};

static_assert(sizeof(JoinArgs<unsigned __int128>) <= 4000, "kernel arguments must stay under the 4 KiB kernarg limit");

// The append cursor is sharded: a returning atomicAdd on one word tops out near 88 ops/us on gfx950
// (one L2 channel), which bounded the first two versions of this kernel.  Tiles use shard
// blockIdx % NSHARD; every shard owns its own cursor word (128 B apart -> different channels) and
// its own slice [shard*cap, (shard+1)*cap) of the hit buffer; finish() packs the slices.
constexpr int NSHARD = 16;
constexpr int CTL_OVERFLOW = 1, CTL_GIANT = 2, CTL_SCRATCH = 3, CTL_SHARD0 = 16, CTL_STRIDE = 16;
constexpr int XSHARD = NSHARD;            // k_expand: sharded totals / cursors (one shared word serialises at ~90 atomics/us); = NSHARD: its output slices feed the partition sort
constexpr int CTL_X0 = CTL_SHARD0 + 2 * NSHARD * CTL_STRIDE;
constexpr int CTL_WORDS = CTL_X0 + XSHARD * CTL_STRIDE;
struct XBases { unsigned long long base[XSHARD]; };
struct ShardSpan { unsigned long long start[NSHARD + 1]; };               // first packed index of every shard slice
__host__ __device__ inline int ctl_cursor(int shard) { return CTL_SHARD0 + shard * CTL_STRIDE; }
__host__ __device__ inline int ctl_umi_or(int shard) { return CTL_SHARD0 + shard * CTL_STRIDE + 1; }   // OR of the UMI codes seen
__host__ __device__ inline int ctl_ncursor(int shard) { return CTL_SHARD0 + shard * CTL_STRIDE + 2; }  // cursor of the no-base stream
__host__ __device__ inline int ctl_accepted(int shard) { return CTL_SHARD0 + (NSHARD + shard) * CTL_STRIDE; }

struct ReadInfo { int32_t pos, endpos, n_al; uint32_t c0, c1; int32_t cell; uint64_t umi; bool ok;
                  bool span_is_cigar;        // endpos - pos is the CIGAR's reference length (false: unmapped flag / no CIGAR: 1)
                  int32_t m_rej, m_acc; };   // fraction mode: m < m_rej fails, m >= m_acc passes, between: divide

// `m / float(n) < min_include` (rdr/fc/core.py:160-165) is an IEEE double comparison of a ROUNDED quotient.  The
// fp64 divide costs ~40 VALU per (read, region) pair, so each read carries two integer bounds instead: with
// p = RN(f * n), any m < p(1 - 2^-50) has RN(m/n) < f and any m > p(1 + 2^-50) has RN(m/n) >= f (three roundings
// of 2^-53 each stay inside the 2^-50 margin); only an m between the bounds (f * n within 2^-50 of an integer)
// takes the exact divide.  tests/test_host_logic.py::test_fraction_bounds checks the equivalence exhaustively.
__device__ __forceinline__ void frac_bounds(ReadInfo& r, double f) {
    const double p = f * (double)r.n_al;
    r.m_rej = (int32_t)ceil(p * (1.0 - 0x1p-50));
    r.m_acc = (int32_t)floor(p * (1.0 + 0x1p-50)) + 1;
}
__device__ __forceinline__ bool frac_below(int32_t m, const ReadInfo& r0, double f) {
    ReadInfo r = r0;
    asm volatile("" : "+v"(r.n_al));                                  // (keeps the fp64 bounds here: hoisted, they cost every read of every sweep ~14 instructions)
    frac_bounds(r, f);                                                // only (read, region) pairs with a partial overlap get here
    if (m < r.m_rej) return true;
    if (m >= r.m_acc) return false;
    return (double)m / (double)r.n_al < f;
}

// ---- join kernel: one 256-thread block per tile of 1024 consecutive reads ---------------------
// Reads are coordinate sorted, so a tile touches a handful of index windows, regions / SNPs and
// one contiguous run of CIGAR words: all of that is staged in LDS once per tile (with a global
// fallback for anything outside the staged range - the staging is a cache, never a correctness
// assumption).  Accepted (read, region) keys go into an LDS hash set (64-bit keys), which removes
// the PCR/UMI duplicates that sit next to each other in a sorted BAM before they ever reach HBM;
// pileup hits and 128-bit keys go through an LDS queue.  The set / queue is flushed as one
// contiguous COO fragment: ONE atomicAdd on the global cursor per flush (a cursor word saturates
// at ~88 returning atomics/us on gfx950 - one per 256 reads was the bottleneck of the first
// version), wave ballot + mbcnt prefix compaction, coalesced 8/16-byte stores.
#ifndef XCK_STAMPS
#define XCK_STAMPS 0
#endif
// which of the 16 cursor shards a join block appends to: block index modulo 16 (XCK_SHARD_SHIFT 0), or runs of 2^shift
// consecutive tiles per shard (experiment for a cross-tile duplicate filter)
#ifndef XCK_SHARD_SHIFT
#define XCK_SHARD_SHIFT 0
#endif
#define JOIN_SHARD ((int)((blockIdx.x >> XCK_SHARD_SHIFT) & (NSHARD - 1)))
#ifndef XCK_TILE_ITEMS
#define XCK_TILE_ITEMS 4
#endif
#ifndef XCK_BAF_SPLIT
#define XCK_BAF_SPLIT 1           // pileup, 64-bit keys: hits without a base go to a second stream that is never sorted
#endif
#ifndef XCK_BAF_NQUEUE_BYTES
#define XCK_BAF_NQUEUE_BYTES 3072   // split mode: queue of the gap records (16 B each, about one per spliced read), flushed at the tile end
#endif
#ifndef XCK_BAF_BQUEUE_BYTES
#define XCK_BAF_BQUEUE_BYTES 3072   // split mode: queue of the hits with a base (~1 in 10), flushed at the tile end
#endif
#ifndef XCK_CX_CAP
#define XCK_CX_CAP 128            // pileup: spliced / indel reads of a tile set aside for the joint walk (beyond it they are walked in place)
#endif
#ifndef XCK_HS_BYTES
#define XCK_HS_BYTES 16384
#endif
#ifndef XCK_CG_CAP
#define XCK_CG_CAP 1536
#endif
#ifndef XCK_ST_CAP
#define XCK_ST_CAP 256
#endif
constexpr int TILE_ITEMS = XCK_TILE_ITEMS;
constexpr int TILE = JOIN_BLOCK * TILE_ITEMS;
constexpr int HS_BYTES = XCK_HS_BYTES;   // LDS set / queue storage per block
constexpr int HS_SLOTS = HS_BYTES / 8;   // slots of the 64-bit key set
constexpr int CG_CAP = XCK_CG_CAP;       // staged CIGAR words (basefc)
#ifndef XCK_CG_CAP_BAF
#define XCK_CG_CAP_BAF 1408       // pileup: with the 3 KB queues and 128 set-aside reads below its block fits 26.5 KB of LDS, i.e. 6 blocks per CU (6.14 -> 5.94 ms)
#endif
template <int MODE> struct CigCap { static constexpr int value = MODE == XCK_MODE_BAF ? XCK_CG_CAP_BAF : CG_CAP; };
constexpr int ST_CAP = XCK_ST_CAP;       // staged regions / SNPs
constexpr int ST_WIN = 64;               // staged index windows

template <class K, int MODE> struct JoinSmem {
    // basefc, 64-bit keys: accepted keys go through an LDS hash SET (a duplicate (region, cell, UMI) is dropped): the PCR /
    // UMI duplicates that sit next to each other in a sorted BAM never reach HBM.  Everything else uses plain queues.
    static constexpr bool USE_SET = sizeof(K) == 8 && MODE == XCK_MODE_BASEFC;
    static constexpr bool HAS_VAL = MODE == XCK_MODE_BAF;
    static constexpr int  SLOTS = HS_SLOTS;
    // pileup split mode (64-bit keys): a hit whose read shows NO base at the SNP (the SNP sits in an N / D gap - the
    // bulk of the hits of spliced reads) only matters if a read of the same (SNP, cell, UMI) WITH a base comes later in
    // fetch order (baf/fc/mcount.py:118-119: the earlier read holds the key).  Those hits go to their own queue / HBM
    // stream, which finish() never sorts: it is only looked up against the (small) sorted stream of hits with a base.
    static constexpr bool SPLIT = MODE == XCK_MODE_BAF && sizeof(K) == 8 && XCK_BAF_SPLIT;
    static constexpr int  STORE_BYTES = USE_SET ? SLOTS * 8 : (HAS_VAL ? (SPLIT ? XCK_BAF_BQUEUE_BYTES : 6144) : HS_BYTES);
    static constexpr int  QCAP = STORE_BYTES / (int)(sizeof(K) + (HAS_VAL ? 8 : 0));
    alignas(16) unsigned char store[STORE_BYTES];
    uint32_t cig[CigCap<MODE>::value];
    static constexpr int ST_BC = MODE == XCK_MODE_BASEFC ? ST_CAP : 1;    // region ends / rows: basefc only
    int32_t  st_a[ST_CAP], st_b[ST_BC], st_c[ST_BC];
    int32_t  st_w[ST_WIN + 1];
    uint32_t cg_lo, cg_n;                // staged CIGAR range [cg_lo, cg_lo + cg_n)
    int32_t  cg_all;                     // 1: that is the tile's whole CIGAR run
    int32_t  w0, nw;                     // basefc: staged regions [w0, w0 + nw) of the start-sorted arrays; pileup: staged SNP windows
    int32_t  k0, nk;                     // pileup: staged SNPs [k0, k0 + nk); basefc: k0 = position of the tile's first read
    uint32_t count;                      // entries currently in the queue
    uint32_t wuor[2 * (JOIN_BLOCK / 64)];  // per-wave OR of the UMI codes
    uint32_t wcnt[JOIN_BLOCK / 64];
    unsigned long long base;
    static constexpr int  NQCAP = SPLIT ? XCK_BAF_NQUEUE_BYTES / 16 : 1;
    uint64_t nq_key[NQCAP], nq_val[NQCAP];
    uint32_t ncount;
    unsigned long long nbase;
    // pileup: (read, SNP) pairs under aligned blocks, parked per wave (64 slots each) until their bases are fetched together
    static constexpr int PR = MODE == XCK_MODE_BAF ? JOIN_BLOCK : 1;
    uint64_t pk_umi[PR];
    // pileup: reads with N / D gaps (or without a CIGAR span) of the whole tile, set aside for pileup_complex()
    static constexpr int CXCAP = MODE == XCK_MODE_BAF ? XCK_CX_CAP : 1;
    uint64_t cx_umi[CXCAP]; int32_t cx_pos[CXCAP], cx_end[CXCAP], cx_cell[CXCAP], cx_idx[CXCAP]; uint32_t cx_c0[CXCAP], cx_c1[CXCAP], cx_s0[CXCAP], cx_sl[CXCAP];
    uint32_t cx_n;
    int32_t  pk_k[PR], pk_qi[PR], pk_cell[PR], pk_idx[PR];
    uint32_t pk_s0[PR], pk_sl[PR];
    __device__ K* keys() { return reinterpret_cast<K*>(store); }
    __device__ uint64_t* vals() { return reinterpret_cast<uint64_t*>(store + (size_t)QCAP * sizeof(K)); }
    __device__ unsigned long long* hkeys() { return reinterpret_cast<unsigned long long*>(store); }            // set mode
};

// (sm.cg_all - block-uniform, from k_tile_meta - says that the tile's whole CIGAR run is staged: practically always at the capacities
// above, and the two-path form costs a handful of exec-mask instructions per word)
template <class K, int MODE>
__device__ __forceinline__ uint32_t cig_at(const JoinArgs<K>& a, const BatchDesc& d, const JoinSmem<K, MODE>& sm, uint32_t c) {
    uint32_t rel = c - sm.cg_lo;
    if (__builtin_amdgcn_readfirstlane(sm.cg_all)) return sm.cig[rel];
    return rel < sm.cg_n ? sm.cig[rel] : as_global(d.cigar)[c];
}

// the seven SoA fields of one read, fetched one sweep ahead of their use (software prefetch)
struct RawRead { int32_t pos, cell; uint64_t umi; uint32_t c0, c1; uint32_t flag; int32_t mapq; uint32_t s0, s1; bool valid; };
// The arrays are addressed as (scalar base + tile start) + a per-lane offset below 8 KB: the loads then take the base from SGPRs
// and ONE 32-bit offset register per element size, where base + 64-bit index arithmetic cost ~12 VALU instructions per sweep -
// in a kernel that is bound by VALU issue.  tile0 is wave-uniform; k = read of the tile (< TILE).
template <bool WITH_SEQ>
__device__ __forceinline__ RawRead fetch_read(const BatchDesc& d, int tile0, uint32_t k, bool valid) {
    RawRead w; w.valid = valid; w.pos = 0; w.cell = -1; w.umi = 0; w.c0 = w.c1 = 0; w.flag = 0; w.mapq = 0; w.s0 = w.s1 = 0;
    if (valid) { w.flag = (as_global(d.flag) + tile0)[k]; w.mapq = (as_global(d.mapq) + tile0)[k]; w.cell = (as_global(d.cell) + tile0)[k]; w.umi = (as_global(d.umi) + tile0)[k];
                 w.pos = (as_global(d.pos) + tile0)[k]; w.c0 = (as_global(d.cig_off) + tile0)[k]; w.c1 = (as_global(d.cig_off) + tile0)[k + 1];
                 if (WITH_SEQ) { w.s0 = (as_global(d.seq_off) + tile0)[k]; w.s1 = (as_global(d.seq_off) + tile0)[k + 1]; } }
    return w;
}

// filter + CIGAR summary of a read (endpos = htslib bam_endpos, n_al = len(read.positions))
template <class K, int MODE>
__device__ __forceinline__ ReadInfo load_read(const JoinArgs<K>& a, const BatchDesc& d, const JoinSmem<K, MODE>& sm, const RawRead& w) {
    ReadInfo r; r.ok = false; r.span_is_cigar = false; r.pos = 0; r.endpos = 0; r.n_al = 0; r.c0 = r.c1 = 0; r.cell = -1; r.umi = 0; r.m_rej = 0; r.m_acc = 0;
    if (!w.valid) return r;
    uint32_t flag = w.flag;
    int32_t mapq = w.mapq;
    r.cell = w.cell;
    r.umi = w.umi;
    r.pos = w.pos;
    r.c0 = w.c0; r.c1 = w.c1;
    bool ok = mapq >= a.f.min_mapq;
    ok = ok && !(a.f.excl_flag && (flag & a.f.excl_flag));
    ok = ok && !(a.f.incl_flag && !(flag & a.f.incl_flag));
    ok = ok && !(a.f.no_orphan && (flag & BAM_FPAIRED) && !(flag & BAM_FPROPER_PAIR));
    ok = ok && r.cell >= 0 && r.umi != XCK_UMI_NONE;                  // (a negative pos is kept: fetch() only asks pos < stop && endpos > start)
    if (!ok) return r;
    int32_t rlen = 0, n_al = 0;
    for (uint32_t c = r.c0; c < r.c1; c++) {
        uint32_t w = cig_at(a, d, sm, c); uint32_t op = w & 15u; int32_t l = int32_t(w >> 4);
        if (op_ref(op)) rlen += l;
        if (op_aligned(op)) n_al += l;
    }
    // htslib bam_endpos(): an unmapped-flagged read, or one without reference-consuming CIGAR, spans one base for fetch();
    // read.positions (the include test) still follows the CIGAR
    r.span_is_cigar = !((flag & BAM_FUNMAP) || r.c1 == r.c0 || rlen == 0);
    if (!r.span_is_cigar) rlen = 1;
    r.endpos = r.pos + rlen;
    r.n_al = n_al;
    r.ok = n_al >= a.f.min_len;
    return r;
}

// __get_include_len(): aligned bases with s0 <= p < e0
template <class K, int MODE>
__device__ __forceinline__ int32_t included_len(const JoinArgs<K>& a, const BatchDesc& d, const JoinSmem<K, MODE>& sm, const ReadInfo& r, int32_t s0, int32_t e0) {
    // both shortcuts need endpos to be the CIGAR's own end (an unmapped-flagged read with a CIGAR is fetched by its first base
    // only, yet its aligned positions are counted over the whole CIGAR)
    if (r.span_is_cigar && r.pos >= s0 && r.endpos <= e0) return r.n_al;
    // no D / N in the CIGAR (reference span == aligned length): the aligned bases are one block, no walk needed
    if (r.span_is_cigar && r.endpos - r.pos == r.n_al) return max(min(r.endpos, e0) - max(r.pos, s0), 0);
    int32_t p = r.pos, m = 0;
    for (uint32_t c = r.c0; c < r.c1; c++) {
        uint32_t w = cig_at(a, d, sm, c); uint32_t op = w & 15u; int32_t l = int32_t(w >> 4);
        if (op_aligned(op)) {
            int32_t lo = max(p, s0), hi = min(p + l, e0);
            if (hi > lo) m += hi - lo;
            p += l;
        } else if (op_ref(op)) p += l;
    }
    return m;
}


// append straight to HBM (slow path: LDS set/queue saturated)
template <class K, int MODE>
__device__ __forceinline__ void emit_global(const JoinArgs<K>& a, K key, uint64_t val) {
    const int shard = JOIN_SHARD;
    unsigned long long idx = atomicAdd(&a.ctl[ctl_cursor(shard)], 1ull);
    if (idx < a.cap) { idx += (unsigned long long)shard * a.cap; a.keys[idx] = key; if (MODE == XCK_MODE_BAF) a.vals[idx] = val; }
    else atomicExch(&a.ctl[CTL_OVERFLOW], 1ull);
}

// slot of a 64-bit key: full-rate VALU only (a 64-bit multiply is four quarter-rate v_mul ops on CDNA)
template <int SLOTS>
__device__ __forceinline__ uint32_t set_slot(unsigned long long kk) {
    const uint32_t lo = (uint32_t)kk, hi = (uint32_t)(kk >> 32);
    uint32_t x = lo ^ ((hi << 9) | (hi >> 23));
    x ^= x >> 15;
    // (written as asm: only bits 12.. of the product are used, so the compiler narrows __umul24 to a plain 32-bit multiply -
    // v_mul_lo_u32, a quarter-rate instruction on gfx9 - where the 24-bit form issues at full rate)
    uint32_t p;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(p) : "v"(x), "v"(0x9E3779u));
    return (p >> 12) & (SLOTS - 1);
}

template <class K, int MODE>
__device__ __forceinline__ void emit(const JoinArgs<K>& a, JoinSmem<K, MODE>& sm, K key, uint64_t val) {
    if constexpr (JoinSmem<K, MODE>::USE_SET) {
        constexpr int SLOTS = JoinSmem<K, MODE>::SLOTS;
        unsigned long long* set = sm.hkeys();
        const unsigned long long kk = (unsigned long long)key;
        uint32_t slot = set_slot<SLOTS>(kk);
        // double hashing: the 64 lanes of a wave wait for the LONGEST probe sequence among their keys, and linear probing's
        // clusters make that 8 - 10 probes at the fill a tile reaches; an odd, key-dependent stride (any odd stride visits every
        // slot of a power-of-two table) keeps the sequences geometric
        const uint32_t stride = (((uint32_t)(kk >> 7) ^ (uint32_t)(kk >> 41)) | 1u) & (SLOTS - 1);
        for (int probe = 0; probe < 24; probe++) {
            unsigned long long prev = atomicCAS(&set[slot], ~0ull, kk);
            if (prev == ~0ull || prev == kk) return;                 // new key, or a duplicate (same region, cell, UMI); no shared counter: flush points are static
            slot = (slot + stride) & (SLOTS - 1);
        }
        emit_global<K, MODE>(a, key, val);
    } else {
        uint32_t idx = atomicAdd(&sm.count, 1u);
        if (idx < (uint32_t)JoinSmem<K, MODE>::QCAP) { sm.keys()[idx] = key; if (MODE == XCK_MODE_BAF) sm.vals()[idx] = val; }
        else emit_global<K, MODE>(a, key, val);
    }
}

// no-base pileup hit (split mode): second LDS queue, spill straight to the second HBM stream
template <class K, int MODE>
__device__ __forceinline__ void emit_nobase(const JoinArgs<K>& a, JoinSmem<K, MODE>& sm, K key, uint64_t val) {
    const uint32_t idx = atomicAdd(&sm.ncount, 1u);
    if (idx < (uint32_t)JoinSmem<K, MODE>::NQCAP) { sm.nq_key[idx] = (uint64_t)key; sm.nq_val[idx] = val; }
    else {
        const int shard = JOIN_SHARD;
        unsigned long long g = atomicAdd(&a.ctl[ctl_ncursor(shard)], 1ull);
        if (g < a.cap) { g += (unsigned long long)shard * a.cap; a.nkeys[g] = key; a.nvals[g] = val; }
        else atomicExch(&a.ctl[CTL_OVERFLOW], 1ull);
    }
}
// phase stamps of the join kernel (build with -DXCK_STAMPS=1): cycles of wave 0 of every block between two stamps, kept in
// registers and stored over the block's own TileMeta record (12 words, read in the prologue and dead since) - no atomics, no
// extra traffic worth the name (the first version added 12 global atomics per block: 48 ms instead of 7.5).  The host sums
// the records after the launch; finish_t() prints the table.  Slots: 0 tile record, 1 prologue loads + staging, 2 staging barrier,
// 3-6 the four sweeps, 7 barrier before the flush, 8 flush: count, 9 flush: cursor atomic, 10 flush: stores, 11 tail.
struct StampRec { long long t; uint32_t d[12]; };
#if XCK_STAMPS
#define XCK_STAMP(sr, slot) do { const long long t_ = clock64(); (sr)->d[slot] += (uint32_t)(t_ - (sr)->t); (sr)->t = t_; } while (0)
#else
#define XCK_STAMP(sr, slot) do {} while (0)
#endif
// split mode: both queues leave in ONE round (two cursor atomics in flight together, one set of barriers)
template <class K, int MODE>
__device__ __forceinline__ void flush_split(const JoinArgs<K>& a, JoinSmem<K, MODE>& sm, StampRec* ts = nullptr) {   // block-wide; all inserts are complete (barrier before)
    const uint32_t tb = min(sm.count, (uint32_t)JoinSmem<K, MODE>::QCAP), tn = min(sm.ncount, (uint32_t)JoinSmem<K, MODE>::NQCAP);
    if (threadIdx.x < 2) {
        const uint32_t total = threadIdx.x ? tn : tb;
        unsigned long long b = 0;
        if (total) {
            const int shard = JOIN_SHARD;
            b = atomicAdd(&a.ctl[threadIdx.x ? ctl_ncursor(shard) : ctl_cursor(shard)], (unsigned long long)total);
            if (b + total > a.cap) { atomicExch(&a.ctl[CTL_OVERFLOW], 1ull); b = ~0ull; }
            else b += (unsigned long long)shard * a.cap;
        }
        if (threadIdx.x) sm.nbase = b; else sm.base = b;
    }
    __syncthreads();
    XCK_STAMP(ts, 9);
    const unsigned long long db = sm.base, dn = sm.nbase;
    if (db != ~0ull) for (uint32_t t = threadIdx.x; t < tb; t += JOIN_BLOCK) { a.keys[db + t] = sm.keys()[t]; a.vals[db + t] = sm.vals()[t]; }
    if (dn != ~0ull) for (uint32_t t = threadIdx.x; t < tn; t += JOIN_BLOCK) { a.nkeys[dn + t] = (K)sm.nq_key[t]; a.nvals[dn + t] = sm.nq_val[t]; }
    __syncthreads();
    if (threadIdx.x == 0) { sm.count = 0; sm.ncount = 0; }
    __syncthreads();
    XCK_STAMP(ts, 10);
}

// write the LDS set / queue to HBM as one contiguous fragment; block-wide call
template <class K, int MODE>
__device__ __forceinline__ void flush(const JoinArgs<K>& a, JoinSmem<K, MODE>& sm, StampRec* ts = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if constexpr (JoinSmem<K, MODE>::USE_SET) {
        unsigned long long* set = sm.hkeys();
        constexpr int PER_WAVE = JoinSmem<K, MODE>::SLOTS / (JOIN_BLOCK / 64);
        uint32_t c = 0;
        for (int s = wave * PER_WAVE + lane; s < (wave + 1) * PER_WAVE; s += 64) c += (set[s] != ~0ull) ? 1u : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
        if (lane == 0) sm.wcnt[wave] = c;
        __syncthreads();
        XCK_STAMP(ts, 8);
        if (threadIdx.x == 0) {
            uint32_t total = sm.wcnt[0] + sm.wcnt[1] + sm.wcnt[2] + sm.wcnt[3];
            unsigned long long b = 0;
            if (total) {
                const int shard = JOIN_SHARD;
                b = atomicAdd(&a.ctl[ctl_cursor(shard)], (unsigned long long)total);
                if (b + total > a.cap) { atomicExch(&a.ctl[CTL_OVERFLOW], 1ull); b = ~0ull; }
                else b += (unsigned long long)shard * a.cap;
            }
            sm.base = b; sm.count = 0;
        }
        __syncthreads();
        XCK_STAMP(ts, 9);
        unsigned long long dst = sm.base;
        for (int w = 0; w < wave; w++) dst += sm.wcnt[w];
        // (wave-uniform: kept in SGPRs, so that the stores take base + fragment start from scalars and one 32-bit lane offset)
        dst = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(dst >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)dst);
        const bool fits = sm.base != ~0ull;
        for (int s = wave * PER_WAVE + lane; s < (wave + 1) * PER_WAVE; s += 64) {
            unsigned long long v = set[s];
            bool valid = v != ~0ull;
            unsigned long long m = __ballot(valid);
            if (valid) {
                uint32_t pre = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                if (fits) (a.keys + dst)[pre] = (K)v;
                set[s] = ~0ull;
            }
            dst += __popcll(m);
        }
        __syncthreads();
        XCK_STAMP(ts, 10);
    } else {
        __syncthreads();
        const uint32_t total = min(sm.count, (uint32_t)JoinSmem<K, MODE>::QCAP);
        if (threadIdx.x == 0) {
            unsigned long long b = 0;
            if (total) {
                const int shard = JOIN_SHARD;
                b = atomicAdd(&a.ctl[ctl_cursor(shard)], (unsigned long long)total);
                if (b + total > a.cap) { atomicExch(&a.ctl[CTL_OVERFLOW], 1ull); b = ~0ull; }
                else b += (unsigned long long)shard * a.cap;
            }
            sm.base = b;
        }
        __syncthreads();
        const unsigned long long dst = sm.base;
        if (dst != ~0ull)
            for (uint32_t t = threadIdx.x; t < total; t += JOIN_BLOCK) {
                a.keys[dst + t] = sm.keys()[t];
                if (MODE == XCK_MODE_BAF) a.vals[dst + t] = sm.vals()[t];
            }
        __syncthreads();
        if (threadIdx.x == 0) sm.count = 0;
        __syncthreads();
    }
}

// basefc: read x region interval join, region-major.  Regions are sorted by start inside the contig, the reads of a wave are
// (in a sorted BAM) a narrow position range, so the whole wave walks ONE short list together: from the tile's first
// candidate (first region whose running-maximum end lies beyond the tile's first position, k_tile_meta) up to the first
// region that starts at or after the end of every read of the wave (one ballot per step, no reduction).  Start, end and row of a region are wave-uniform (LDS
// broadcast reads of the staged slice); each lane only compares its own read against them - no per-lane index lookups,
// no divergent loop counts.  A wave that holds a read left of the tile's first read (unsorted input) scans from the
// contig's first region, so sortedness is a speed assumption, never a correctness one.
// minimum / maximum of one int32 per lane over the 64 lanes of a wave (all lanes active): four row_shr steps inside the rows
// of 16 lanes, row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3 - lane 63 then holds the result.  Six DPP
// VALU ops and one v_readlane; no LDS traffic (a __shfl_xor butterfly is six ds_bpermute round trips).
template <bool MAX>
__device__ __forceinline__ int32_t wave_minmax(int32_t v) {
    constexpr int32_t ID = MAX ? std::numeric_limits<int32_t>::min() : std::numeric_limits<int32_t>::max();
#define XCK_DPP_STEP(ctrl, rows) { const int32_t t_ = __builtin_amdgcn_update_dpp(ID, v, ctrl, rows, 0xf, false); v = MAX ? max(v, t_) : min(v, t_); }
    XCK_DPP_STEP(0x111, 0xf) XCK_DPP_STEP(0x112, 0xf) XCK_DPP_STEP(0x114, 0xf) XCK_DPP_STEP(0x118, 0xf)   // row_shr:1,2,4,8
    XCK_DPP_STEP(0x142, 0xa) XCK_DPP_STEP(0x143, 0xc)                                                       // row_bcast:15, row_bcast:31
#undef XCK_DPP_STEP
    return __builtin_amdgcn_readlane(v, 63);
}

// The walk is done 64 regions at a time with the REGIONS in the lanes: lane l loads start / end of region kb + l (one
// conflict-free LDS read per array) and asks whether the region can meet any read of the wave at all - start below the
// largest read end, end beyond the smallest read position (two DPP reductions per sweep).  One ballot gives the candidate
// regions of the chunk; only those are visited, their start / end / row taken from the lanes by v_readlane (no memory
// round trip, scalar operands for the per-read compare).  The first version visited every region from the tile's first
// candidate on, each visit two dependent LDS reads + readfirstlane (7.76 -> 7.44 ms at configs[2]; the walk it replaces is in
// the history: commit 97658ab).  What bounds the kernel now is VALU issue: 1756 VALU instructions per wave x 4 cycles x 6
// waves per SIMD = 88 % of a wave's 48 k-cycle life; by ablation (profiles/experiments/join_r04/) 676 of them are the read
// summary + prologue + flush, 583 the walk, ~500 the set inserts.
template <class K, int MODE>
__device__ __forceinline__ uint32_t join_regions(const JoinArgs<K>& a, const BatchDesc& d, JoinSmem<K, MODE>& sm, const ReadInfo& r,
                                                 const int32_t lb, const int32_t n_st, const int32_t p_first, StampRec* ts = nullptr, int sweep = 0) {   // staged slice: regions [lb, lb + n_st); scalars
    uint32_t n_acc = 0;
    if (!__ballot(r.ok)) return 0;                                   // no read of this wave passed the filter (wave-uniform)
    const int lane = threadIdx.x & 63;
    const int32_t reg_lo = __builtin_amdgcn_readfirstlane(d.reg_lo), reg_hi = __builtin_amdgcn_readfirstlane(d.reg_hi);
    const int32_t wmin = wave_minmax<false>(r.ok ? r.pos : std::numeric_limits<int32_t>::max());
    const int32_t wmax = wave_minmax<true>(r.ok ? r.endpos : std::numeric_limits<int32_t>::min());
    // A region that CONTAINS the whole wave (start <= smallest position, end >= largest read end - two scalar compares; genes are
    // kilobases long, the 64 reads of a wave span a few hundred bases at most, so this is the usual case) takes every read of the
    // wave that passed the filter: all its aligned bases are inside (included_len's first shortcut), m == n, the fraction is 1.
    // What is left per read is the key and the set insert; the overlap / include arithmetic runs only for regions whose
    // boundary falls inside the wave.  (A read whose fetch span is not its CIGAR's - unmapped flag - rules the shortcut out.)
    const bool all_span = !__ballot(r.ok && !r.span_is_cigar);
    const bool full_ok = r.ok && (a.f.frac_mode ? r.n_al > 0 : r.n_al >= a.f.min_inc_len);
    const K kbase = a.kl.make(0u, (uint32_t)r.cell, r.umi);
    // p_first = position of the tile's first read: a read left of it means unsorted input, the list is then walked from its start
    for (int32_t kb = __ballot(r.ok && r.pos < p_first) ? reg_lo : lb; kb < reg_hi; kb += 64) {
        const int32_t k = kb + lane;
        const uint32_t rel = (uint32_t)(k - lb);
        // (scalar: the whole chunk is staged and inside the contig - the usual case; the per-lane form below is the general one)
        const bool whole = (uint32_t)(kb - lb) + 64u <= (uint32_t)n_st && kb + 64 <= reg_hi;
        bool in = true, staged = true;
        int32_t s0 = std::numeric_limits<int32_t>::max(), e0 = std::numeric_limits<int32_t>::min(), row = 0;
        if (whole) { s0 = sm.st_a[rel]; e0 = sm.st_b[rel]; }
        else {
            in = k < reg_hi; staged = rel < (uint32_t)n_st;
            if (in) { s0 = staged ? sm.st_a[rel] : as_global(a.reg_s0)[k]; e0 = staged ? sm.st_b[rel] : as_global(a.reg_e0)[k]; }
        }
        const bool last = __ballot(!in || s0 >= wmax) != 0;          // sorted by start: no read of the wave reaches a region after this chunk
        unsigned long long cand = __ballot(in && s0 < wmax && e0 > wmin);
        if (cand) { if (whole) row = sm.st_c[rel]; else if (in) row = staged ? sm.st_c[rel] : as_global(a.reg_row)[k]; }
#if XCK_STAMPS == 2
        if (sweep == 0) { XCK_STAMP(ts, 4); ts->d[11] += (uint32_t)__popcll(cand); }      // (slot 11: candidate regions of sweep 0)
#endif
        while (cand) {
            const int b = (int)__builtin_ctzll(cand); cand &= cand - 1;
            const int32_t rs0 = __builtin_amdgcn_readlane(s0, b), re0 = __builtin_amdgcn_readlane(e0, b), rrow = __builtin_amdgcn_readlane(row, b);
            const K krow = K((uint32_t)rrow) << (a.kl.cbits + a.kl.ubits);      // wave-uniform: scalar shift
            if (all_span && wmin >= rs0 && wmax <= re0) {             // the region contains every read of the wave
                if (full_ok) { emit<K, MODE>(a, sm, kbase | krow, 0); n_acc++; }
                continue;
            }
            if (!(r.ok && r.pos < re0 && r.endpos > rs0)) continue;  // htslib fetch overlap
            const int32_t m = included_len(a, d, sm, r, rs0, re0);
            if (a.f.frac_mode) {
                if (r.n_al <= 0) continue;
                // m == n gives exactly 1.0, never below a threshold in (0,1): skip the fp64 divide
                if (m != r.n_al && frac_below(m, r, a.f.min_inc_frac)) continue;   // IEEE double, as m / float(n)
            } else if (m < a.f.min_inc_len) continue;
            emit<K, MODE>(a, sm, kbase | krow, 0);
            n_acc++;
        }
        if (last) break;
    }
    return n_acc;
}


// position of SNP k (staged slice first)
template <class K, int MODE>
__device__ __forceinline__ int32_t snp_p0(const JoinArgs<K>& a, const JoinSmem<K, MODE>& sm, int32_t k) {
    const uint32_t dl = (uint32_t)(k - sm.k0);
    return dl < (uint32_t)sm.nk ? sm.st_a[dl] : as_global(a.snp_p0)[k];
}

template <class K, int MODE>
__device__ __forceinline__ int32_t lower_snp_tail(const JoinArgs<K>& a, const BatchDesc& d, int32_t k, int32_t x) {
    while (k < d.snp_end && as_global(a.snp_p0)[k] < x) k++;
    return k;
}
// first SNP k >= k_from of the contig with position >= x: binary search in the staged slice, linear beyond it
template <class K, int MODE>
__device__ __forceinline__ int32_t lower_snp(const JoinArgs<K>& a, const BatchDesc& d, const JoinSmem<K, MODE>& sm, int32_t k_from, int32_t x) {
    int32_t k = k_from;
    const uint32_t dl = (uint32_t)(k - sm.k0);
    if (dl < (uint32_t)sm.nk) {
        int32_t lo = (int32_t)dl, hi = sm.nk;
#pragma unroll
        for (int q = 0; q < 3; q++) if (lo < hi && sm.st_a[lo] < x) lo++;       // the answer is usually 0-2 SNPs away
        if (lo < hi && sm.st_a[lo] >= x) return sm.k0 + lo;
        while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (sm.st_a[mid] < x) lo = mid + 1; else hi = mid; }
        k = sm.k0 + lo;
        if (lo < sm.nk) return k;
    }
    return lower_snp_tail<K, MODE>(a, d, k, x);
}
// One read whose reference span is not one aligned block (N / D gaps, no CIGAR span, unmapped flag): the pileup of
// baf/fc/mcount.py:109-127 + utils/sam.py:4-40 over its CIGAR.  SNPs under aligned blocks are hits with a base (fetched
// here: these reads are few); the SNPs inside a gap - where the read holds the key but shows no base - leave as ONE range
// record per gap, (first SNP, cell, UMI | ordinal, count - 1), in pieces of 32 SNPs: a spliced read over 20 SNPs costs one
// 16-byte record instead of 20 hits.  Such reads are collected per tile and walked together (k_join), so that the waves
// that walk them are full and the other 85 % of the reads never wait for them.
template <class K, int MODE>
__device__ __forceinline__ uint32_t pileup_complex(const JoinArgs<K>& a, const BatchDesc& d, JoinSmem<K, MODE>& sm, int32_t pos, int32_t endpos,
                                                   uint32_t c0, uint32_t c1, int32_t cell, uint64_t umi, uint32_t s0, uint32_t sl, int32_t idx) {
    const int32_t w_lo = max(pos, 0) >> WSS;
    if (w_lo >= d.n_swin) return 0;
    constexpr uint64_t AL_MASK = (uint64_t)((1u << ALLELE_BITS) - 1);
    const int32_t k_w = (uint32_t)(w_lo - sm.w0) < (uint32_t)sm.nw ? sm.st_w[w_lo - sm.w0] : as_global(d.snp_win)[w_lo];
    int32_t k = lower_snp<K, MODE>(a, d, sm, k_w, pos);
    const uint64_t ordv = (d.ordinal_base + (uint64_t)idx) << ALLELE_BITS;
    uint32_t n = 0;
    auto gap = [&](int32_t ka, int32_t kb) {
        if constexpr (JoinSmem<K, MODE>::SPLIT) {
            for (int32_t ks = ka; ks < kb; ks += 32)
                emit_nobase<K, MODE>(a, sm, a.kl.make((uint32_t)ks, (uint32_t)cell, umi), ordv | (uint64_t)(min(kb - ks, 32) - 1));
        } else for (int32_t ks = ka; ks < kb; ks++) emit<K, MODE>(a, sm, a.kl.make((uint32_t)ks, (uint32_t)cell, umi), ordv);
        n += (uint32_t)(kb - ka);
    };
    int32_t rp = pos, q = 0;
    for (uint32_t cc = c0; cc < c1 && rp < endpos; cc++) {
        const uint32_t w = cig_at(a, d, sm, cc); const uint32_t op = w & 15u; const int32_t l = int32_t(w >> 4);
        if (op_ref(op) && l != 0) {
            const int32_t k2 = lower_snp<K, MODE>(a, d, sm, k, min(rp + l, endpos));
            if (op_aligned(op)) {
                for (int32_t kk = k; kk < k2; kk++) {
                    const int32_t qi = q + (snp_p0<K, MODE>(a, sm, kk) - rp);
                    int al = -1;
                    if ((uint32_t)(qi >> 1) < sl) { const uint32_t by = as_global(d.seq)[s0 + (uint32_t)(qi >> 1)]; al = (qi & 1) ? int(by & 15u) : int(by >> 4); }
                    const K key = a.kl.make((uint32_t)kk, (uint32_t)cell, umi);
                    if (JoinSmem<K, MODE>::SPLIT && al < 0) emit_nobase<K, MODE>(a, sm, key, ordv);
                    else emit<K, MODE>(a, sm, key, ordv | (uint64_t)(al + 1));
                }
                n += (uint32_t)(k2 - k);
            } else if (k2 > k) gap(k, k2);
            k = k2; rp += l;
        }
        if (op_aligned(op) || op == 1u || op == 4u) q += l;
    }
    if (rp < endpos) {                                                        // no CIGAR / zero reference length: the position itself, without a base
        const int32_t k2 = lower_snp<K, MODE>(a, d, sm, k, endpos);
        if (k2 > k) gap(k, k2);
    }
    (void)AL_MASK;
    return n;
}

// one thread per tile: locate the batch, read the tile's extent, size the LDS staging
template <int MODE>
__global__ __launch_bounds__(256) void k_tile_meta(BatchTable bt, TileMeta* __restrict__ out, const int32_t* __restrict__ rpmax) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= bt.n_tiles) return;
    int lo = 0, hi = bt.n_batches - 1;
    while (lo < hi) { int mid = (lo + hi + 1) >> 1; if (bt.desc[mid].tile0 <= t) lo = mid; else hi = mid - 1; }
    const BatchDesc& d = bt.desc[lo];
    TileMeta m;
    m.b = lo; m.r0 = (t - d.tile0) * TILE; m.r1 = min(m.r0 + TILE, d.n); m.pad = 0;
    const uint32_t c_lo = as_global(d.cig_off)[m.r0], c_hi = as_global(d.cig_off)[m.r1];
    const int32_t p_first = max(as_global(d.pos)[m.r0], 0), p_last = max(as_global(d.pos)[m.r1 - 1], 0);
    m.c_lo = c_lo; m.cg_n = min(c_hi - c_lo, (uint32_t)CigCap<MODE>::value); m.pad = c_hi - c_lo <= (uint32_t)CigCap<MODE>::value ? 1 : 0;
    m.w0 = p_first >> WSS; m.nw = 0; m.e0 = 0; m.n_ent = 0; m.k0 = 0; m.nk = 0;
    if (MODE == XCK_MODE_BASEFC) {
        // first candidate region of the tile: the first one whose running-maximum end lies beyond the first read's position
        // (every region before it ends at or before that position; reads further right cannot reach them either)
        const int32_t p0 = as_global(d.pos)[m.r0];
        int32_t lo = d.reg_lo, hi = d.reg_hi;
        while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (as_global(rpmax)[mid] > p0) hi = mid; else lo = mid + 1; }
        m.e0 = lo; m.n_ent = min(d.reg_hi - lo, ST_CAP); m.k0 = p0;
        (void)p_last;
    } else {
        if (m.w0 < d.n_swin) { m.k0 = as_global(d.snp_win)[m.w0]; m.nk = min(d.snp_end - m.k0, ST_CAP); m.nw = min(d.n_swin - m.w0, ST_WIN); }
    }
    out[t] = m;
}

// (64-bit pileup: asked to stay at 80 VGPRs, i.e. 6 waves per SIMD beside its 26.5 KB of LDS; the others have room anyway, and the
// 128-bit pileup kernel would spill under that bound)
template <class K, int MODE>
__global__ __launch_bounds__(JOIN_BLOCK) __attribute__((amdgpu_waves_per_eu((sizeof(K) == 8 && MODE == XCK_MODE_BAF) ? 6 : 4, 8))) void k_join(JoinArgs<K> a) {
    __shared__ JoinSmem<K, MODE> sm;
    const int tid = threadIdx.x, lane = tid & 63;
    StampRec t_s0;
#if XCK_STAMPS
    for (int q = 0; q < 12; q++) t_s0.d[q] = 0;
    t_s0.t = clock64();
#endif
#define STAMP(slot) XCK_STAMP(&t_s0, slot)
    // ---- prologue: one record from k_tile_meta, then ONE round of independent loads ----
    const XCK_GLOBAL TileMeta* mp = as_global(a.meta) + blockIdx.x;
    const uint32_t c_lo = mp->c_lo, cg_n = mp->cg_n;
    const int32_t w0 = mp->w0, nw = mp->nw, e0 = mp->e0, n_ent = mp->n_ent, k0 = mp->k0, nk = mp->nk;
    const int b = __builtin_amdgcn_readfirstlane(mp->b);
    const int tile0 = __builtin_amdgcn_readfirstlane(mp->r0);
    const BatchDesc& d = a.bt.desc[b];                                // kernarg: scalar loads through the constant cache
    const int32_t u_lb = __builtin_amdgcn_readfirstlane(e0), u_nst = __builtin_amdgcn_readfirstlane(n_ent), u_p0 = __builtin_amdgcn_readfirstlane(k0);   // basefc: the tile's region slice as scalars
    unsigned long long uor = 0;
    STAMP(0);
    RawRead W[TILE_ITEMS];
    if (__builtin_amdgcn_readfirstlane(tile0 + TILE <= d.n)) {        // a full tile (all but the last of a batch): no per-lane bounds
#pragma unroll
        for (int j = 0; j < TILE_ITEMS; j++) W[j] = fetch_read<MODE == XCK_MODE_BAF>(d, tile0, (uint32_t)(j * JOIN_BLOCK + tid), true);
    } else {
#pragma unroll
        for (int j = 0; j < TILE_ITEMS; j++) W[j] = fetch_read<MODE == XCK_MODE_BAF>(d, tile0, (uint32_t)(j * JOIN_BLOCK + tid), tile0 + j * JOIN_BLOCK + tid < d.n);
    }                                                                 // the whole tile's loads fly during the staging
    if constexpr (JoinSmem<K, MODE>::USE_SET) {
        unsigned long long* set = sm.hkeys();                         // all ones = empty
        for (int s = tid; s < JoinSmem<K, MODE>::SLOTS; s += JOIN_BLOCK) set[s] = ~0ull;
    }
    if (tid == 0) { sm.count = 0; sm.ncount = 0; sm.cx_n = 0; sm.cg_lo = c_lo; sm.cg_n = cg_n; sm.cg_all = mp->pad; sm.k0 = k0; sm.nk = nk;
                    if (MODE == XCK_MODE_BASEFC) { sm.w0 = e0; sm.nw = n_ent; } else { sm.w0 = w0; sm.nw = nw; } }
    // Every global load of the prologue is issued BEFORE the first LDS store: written as load/store loops the
    // compiler waits (s_waitcnt vmcnt(0)) inside each iteration, which serialised ~7 HBM round trips per tile.
    static_assert(ST_CAP <= JOIN_BLOCK && ST_WIN + 1 <= JOIN_BLOCK, "staging assumes one element per thread");
    constexpr int CG_IT = (CigCap<MODE>::value + JOIN_BLOCK - 1) / JOIN_BLOCK;
    uint32_t cw[CG_IT];
    const uint32_t u_clo = __builtin_amdgcn_readfirstlane(c_lo);      // (scalar starts: the loads below take base + start from SGPRs, like fetch_read)
#pragma unroll
    for (int q = 0; q < CG_IT; q++) { const uint32_t c = tid + q * JOIN_BLOCK; cw[q] = c < cg_n ? (as_global(d.cigar) + u_clo)[c] : 0u; }
    int32_t g_a = 0, g_b = 0, g_c = 0, g_w = 0;
    if (MODE == XCK_MODE_BASEFC) {
        if (tid < n_ent) { g_a = (as_global(a.reg_s0) + u_lb)[(uint32_t)tid]; g_b = (as_global(a.reg_e0) + u_lb)[(uint32_t)tid]; g_c = (as_global(a.reg_row) + u_lb)[(uint32_t)tid]; }
    } else {
        const int32_t u_k0 = __builtin_amdgcn_readfirstlane(k0), u_w0 = __builtin_amdgcn_readfirstlane(w0);
        if (tid < nk) g_a = (as_global(a.snp_p0) + u_k0)[(uint32_t)tid];
        if (tid < nw) g_w = (as_global(d.snp_win) + u_w0)[(uint32_t)tid];   // first SNP of each window
    }
#pragma unroll
    for (int q = 0; q < CG_IT; q++) { const uint32_t c = tid + q * JOIN_BLOCK; if (c < cg_n) sm.cig[c] = cw[q]; }
    if (MODE == XCK_MODE_BASEFC) {
        if (tid < n_ent) { sm.st_a[tid] = g_a; sm.st_b[tid] = g_b; sm.st_c[tid] = g_c; }
    } else {
        if (tid < nk) sm.st_a[tid] = g_a;
        if (tid < nw) sm.st_w[tid] = g_w;
    }
    STAMP(1);
    __syncthreads();
    STAMP(2);
    // ---- TILE_ITEMS coalesced sweeps over the tile (the reads were requested in the prologue) ----
    uint32_t acc = 0;
    int pr_n = 0;                                                     // pileup: parked pairs of this wave (wave-uniform)
    // sweeps between two flushes: keep the expected fill (256 reads x ~2 pairs per sweep) under half the set / queue
    constexpr int CAP_ENTRIES = JoinSmem<K, MODE>::USE_SET ? JoinSmem<K, MODE>::SLOTS : JoinSmem<K, MODE>::QCAP;
    // set mode: the de-duplicated fill of a 1024-read tile is a few hundred keys, so flush once, at the end
    // (better de-duplication, half the cursor atomics); saturation still spills correctly through emit_global()
#ifndef XCK_SET_FLUSH_EVERY
#define XCK_SET_FLUSH_EVERY TILE_ITEMS
#endif
    constexpr int FLUSH_EVERY = JoinSmem<K, MODE>::USE_SET ? XCK_SET_FLUSH_EVERY : JoinSmem<K, MODE>::SPLIT ? TILE_ITEMS
                              : ((CAP_ENTRIES / 2 / (JOIN_BLOCK * 2)) < 1 ? 1 : (CAP_ENTRIES / 2 / (JOIN_BLOCK * 2)));
#pragma unroll
    for (int j = 0; j < TILE_ITEMS; j++) {
        const int i = tile0 + j * JOIN_BLOCK + tid;
        const RawRead cur = W[j];
        ReadInfo r = load_read<K, MODE>(a, d, sm, cur);
#if XCK_STAMPS == 2
        if (j == 0) { uint32_t x_ = (uint32_t)r.endpos ^ (uint32_t)r.n_al; asm volatile("" :: "v"(x_)); STAMP(3); }   // (the summary is complete)
#endif
        if constexpr (MODE == XCK_MODE_BAF) {
            // read x SNP join, SNP-major: the 64 reads of a wave are (in a sorted BAM) a narrow position range, so the wave
            // walks the few SNPs of that range TOGETHER - position of SNP k is wave-uniform, every lane only asks "inside my
            // read?" - instead of every read searching the SNP table for itself (two or three binary searches per read, most of
            // them to learn that a 91-base read covers no SNP).  A hit parks its (SNP, query offset) pair in the wave's LDS
            // segment; the bases of the parked pairs are fetched together later (one HBM latency per batch, not per hit).
            uint32_t c = 0, n_gap = 0;
            const int wb = tid & ~63;                                       // first slot of this wave's segment
            constexpr uint64_t AL_MASK = (uint64_t)((1u << ALLELE_BITS) - 1);
            auto drain = [&]() {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
                if (lane < pr_n) {
                    const int u = wb + lane;
                    const int32_t qi = sm.pk_qi[u];
                    int al = -1;                                            // no sequence stored for that offset: key held, no base
                    if ((uint32_t)(qi >> 1) < sm.pk_sl[u]) { const uint32_t by = as_global(d.seq)[sm.pk_s0[u] + (uint32_t)(qi >> 1)]; al = (qi & 1) ? int(by & 15u) : int(by >> 4); }
                    const K key = a.kl.make((uint32_t)sm.pk_k[u], (uint32_t)sm.pk_cell[u], sm.pk_umi[u]);
                    const uint64_t val = ((d.ordinal_base + (uint64_t)sm.pk_idx[u]) << ALLELE_BITS) | (uint64_t)(al + 1);
                    if (JoinSmem<K, MODE>::SPLIT && al < 0) emit_nobase<K, MODE>(a, sm, key, val & ~AL_MASK);   // a record of one SNP
                    else emit<K, MODE>(a, sm, key, val);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();                            // the segment is free again
                pr_n = 0;
            };
            // reads whose reference span is ONE aligned block (no N / D; 85 % of a 10x run) take the wave-uniform walk below; the
            // others are set aside in LDS and walked together after the last sweep (pileup_complex)
            const bool simple = r.ok && r.span_is_cigar && r.endpos - r.pos == r.n_al;
            const unsigned long long cxm = __ballot(r.ok && !simple);
            if (cxm) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&sm.cx_n, (uint32_t)__popcll(cxm));
                base = __builtin_amdgcn_readfirstlane(base);
                if (r.ok && !simple) {
                    const uint32_t u = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(cxm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cxm, 0u));
                    if (u < (uint32_t)JoinSmem<K, MODE>::CXCAP) {
                        sm.cx_pos[u] = r.pos; sm.cx_end[u] = r.endpos; sm.cx_c0[u] = r.c0; sm.cx_c1[u] = r.c1; sm.cx_cell[u] = r.cell; sm.cx_umi[u] = r.umi;
                        sm.cx_s0[u] = cur.s0; sm.cx_sl[u] = cur.s1 - cur.s0; sm.cx_idx[u] = i;
                    } else n_gap += pileup_complex<K, MODE>(a, d, sm, r.pos, r.endpos, r.c0, r.c1, r.cell, r.umi, cur.s0, cur.s1 - cur.s0, i);   // list full: walk it here
                }
            }
            if (__ballot(simple)) {                                         // wave-uniform
                // first SNP to look at: the 1 kb window of the wave's first read (lane 0 exists whenever any lane does); a read
                // left of it means unsorted input - then the contig's SNPs are walked from the start (speed, never correctness)
                const int32_t p_w = __builtin_amdgcn_readfirstlane(cur.pos);
                int32_t k;
                if (__ballot(simple && r.pos < p_w)) k = d.n_swin > 0 ? as_global(d.snp_win)[0] : d.snp_end;
                else { const int32_t w = max(p_w, 0) >> WSS;
                       k = w >= d.n_swin ? d.snp_end : ((uint32_t)(w - sm.w0) < (uint32_t)sm.nw ? sm.st_w[w - sm.w0] : as_global(d.snp_win)[w]); }
                k = __builtin_amdgcn_readfirstlane(k);
                for (; k < d.snp_end; k++) {
                    const int32_t p = __builtin_amdgcn_readfirstlane(snp_p0<K, MODE>(a, sm, k));
                    const bool reach = simple && p < r.endpos;
                    if (!__ballot(reach)) break;                            // SNPs are sorted: no read of the wave reaches this or any later one
                    const bool hit = reach && p >= r.pos;
                    const unsigned long long am = __ballot(hit);
                    if (!am) continue;
                    int32_t qi = p - r.pos;                                 // one aligned op: query offset = reference offset
                    if (hit && r.c1 - r.c0 != 1) {                          // I / S / H / P around the aligned blocks shift the query offset
                        int32_t rp = r.pos, q = 0;
                        for (uint32_t cc = r.c0; cc < r.c1; cc++) {
                            const uint32_t w = cig_at(a, d, sm, cc); const uint32_t op = w & 15u; const int32_t l = int32_t(w >> 4);
                            if (op_aligned(op)) { if (p < rp + l) { qi = q + (p - rp); break; } rp += l; q += l; }
                            else if (op == 1u || op == 4u) q += l;
                        }
                    }
                    const int n_new = __popcll(am);
                    if (pr_n + n_new > 64) drain();
                    if (hit) {
                        const int u = wb + pr_n + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u));
                        sm.pk_k[u] = k; sm.pk_qi[u] = qi; sm.pk_cell[u] = r.cell; sm.pk_umi[u] = r.umi; sm.pk_s0[u] = cur.s0; sm.pk_sl[u] = cur.s1 - cur.s0; sm.pk_idx[u] = i;
                        c++;
                    }
                    pr_n += n_new;
                }
            }
            if (j + 1 == TILE_ITEMS) {
                if (pr_n) drain();
                // the set-aside reads of the whole tile, one per thread: full waves of long walks instead of one long walk per wave and sweep
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                const uint32_t cx_n = min(sm.cx_n, (uint32_t)JoinSmem<K, MODE>::CXCAP);
                for (uint32_t u = tid; u < cx_n; u += JOIN_BLOCK)
                    n_gap += pileup_complex<K, MODE>(a, d, sm, sm.cx_pos[u], sm.cx_end[u], sm.cx_c0[u], sm.cx_c1[u], sm.cx_cell[u], sm.cx_umi[u], sm.cx_s0[u], sm.cx_sl[u], sm.cx_idx[u]);
            }
            if (r.ok) uor |= r.umi;                                   // (finish() puts the haplotype class of the region-level keys into UMI-field bits no code uses)
            acc += c + n_gap;
        } else {
            if (r.ok) uor |= r.umi;
            acc += join_regions<K, MODE>(a, d, sm, r, u_lb, u_nst, u_p0, &t_s0, j);   // wave-uniform call: the sweep over the regions is shared by all 64 lanes
        }
        // Flush points are fixed at compile time, never decided from sm.count: a count-based decision read
        // after the barrier races with the next sweep's inserts (threads could disagree and split at the
        // barriers inside flush()).  A set / queue that saturates between two flush points spills through
        // emit_global(), which is always correct.
#if XCK_STAMPS == 2
        if (j == 0) STAMP(5); else STAMP(6);                          // sweep 0 in three parts (3 read summary, 4 chunk scan, 5 candidates), 6 = sweeps 1-3
#else
        STAMP(3 + j);
#endif
        if ((j + 1) % FLUSH_EVERY == 0 || j + 1 == TILE_ITEMS) {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS only: global prefetches stay in flight
            STAMP(7);
            if constexpr (JoinSmem<K, MODE>::SPLIT) flush_split<K, MODE>(a, sm, &t_s0); else flush<K, MODE>(a, sm, &t_s0);
        }
    }
    // accepted (read, region|SNP) pairs before the LDS de-duplication: the algorithmic unit of the join
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) acc += __shfl_xor(acc, dd, 64);
    {
        // highest UMI-code bit in use: the radix-sort fold drops the dead bits between the UMI codes and the cell field before the sort
        uint32_t ulo = (uint32_t)uor, uhi = (uint32_t)(uor >> 32);
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) { ulo |= __shfl_xor(ulo, dd, 64); uhi |= __shfl_xor(uhi, dd, 64); }
        if (lane == 0) { sm.wuor[2 * (tid >> 6)] = ulo; sm.wuor[2 * (tid >> 6) + 1] = uhi; }
    }
    if (lane == 0) sm.wcnt[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        uor = 0;
#pragma unroll
        for (int w = 0; w < JOIN_BLOCK / 64; w++) uor |= ((unsigned long long)sm.wuor[2 * w + 1] << 32) | sm.wuor[2 * w];
        // one word per shard, on the shard cursor's cache line (a single shared word serialises at ~90 atomics/us)
        if (uor) atomicOr(&a.ctl[ctl_umi_or(JOIN_SHARD)], uor);
    }
    if (tid == 0) { acc = sm.wcnt[0] + sm.wcnt[1] + sm.wcnt[2] + sm.wcnt[3];
                    if (acc) atomicAdd(&a.ctl[ctl_accepted(JOIN_SHARD)], (unsigned long long)acc); }
#if XCK_STAMPS != 2
    STAMP(11);
#endif
#if XCK_STAMPS
    if (tid == 0) { uint32_t* o = (uint32_t*)(a.meta + blockIdx.x); for (int q = 0; q < 12; q++) o[q] = t_s0.d[q]; }
    static_assert(sizeof(TileMeta) == 48, "the stamp record reuses the tile record");
#endif
}

// exclusive scan of one uint32 per thread over a 256-thread block; returns block total in `total`
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* s_wave, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t base = 0; total = 0;
#pragma unroll
    for (int w = 0; w < JOIN_BLOCK / 64; w++) { uint32_t t = s_wave[w]; if (w < wave) base += t; total += t; }
    __syncthreads();
    return base + inc - v;
}

// ------------------------------------------------------------------------------------------
// finish kernels
// ------------------------------------------------------------------------------------------
// basefc fold without a dense intermediate: pass A counts the (row, cell) run heads of every 2048-key tile,
// a one-block scan turns that into output offsets, pass B writes (row, col, #distinct keys of the run)
// straight into the COO arrays.  Heads walk their run (runs average ~2 keys; the walk stays in L2).
constexpr int FD_BLOCK = 256, FD_ITEMS = 8, FD_TILE = FD_BLOCK * FD_ITEMS;

template <class K>
__global__ __launch_bounds__(FD_BLOCK) void k_fold_heads(const K* __restrict__ k, long long n, KeyLayout<K> kl, uint32_t* __restrict__ blk) {
    __shared__ uint32_t s_wave[FD_BLOCK / 64];
    const long long base = (long long)blockIdx.x * FD_TILE;
    uint32_t c = 0;
#pragma unroll
    for (int t = 0; t < FD_ITEMS; t++) {                                  // striped: lane-contiguous, fully coalesced
        const long long i = base + t * FD_BLOCK + threadIdx.x;
        if (i < n) { if (i == 0 || kl.rc(k[i]) != kl.rc(k[i - 1])) c++; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) blk[blockIdx.x] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
}

template <class K>
__global__ __launch_bounds__(FD_BLOCK) void k_fold_emit(const K* __restrict__ k, long long n, KeyLayout<K> kl, const unsigned long long* __restrict__ off,
                                                        int32_t* __restrict__ row, int32_t* __restrict__ col, int32_t* __restrict__ val) {
    // val[] is zero on entry.  A tile knows the distinct keys of every (row, cell) run that STARTS in it only up to
    // the tile end; what a later tile holds of that run (its "lead": distinct keys before its first head) is added
    // with one atomicAdd to the slot of the last head before it.  No thread ever walks a run, so a hot (gene, cell)
    // pair with thousands of UMIs costs the same per key as a cold one.
    __shared__ uint32_t s_wave[FD_BLOCK / 64];
    __shared__ K tile[FD_TILE + 1];                                       // tile[0] = key before the tile (halo)
    __shared__ uint16_t s_hx[FD_TILE + 1], s_he[FD_TILE];                 // per head: distinct keys before it / its element
    const long long base = (long long)blockIdx.x * FD_TILE;
    const int n_loc = (int)min((long long)FD_TILE, n - base);
#pragma unroll
    for (int t = 0; t < FD_ITEMS; t++) { const int e = t * FD_BLOCK + threadIdx.x; if (e < n_loc) tile[1 + e] = k[base + e]; }
    if (threadIdx.x == 0) tile[0] = base > 0 ? k[base - 1] : K(0);
    __syncthreads();
    // blocked arrangement: thread t owns elements [t*FD_ITEMS, (t+1)*FD_ITEMS)
    const int e0 = threadIdx.x * FD_ITEMS;
    uint32_t hmask = 0, dmask = 0;
    K prev = tile[e0];
#pragma unroll
    for (int q = 0; q < FD_ITEMS; q++) {
        const int e = e0 + q;
        if (e < n_loc) {
            const K me = tile[1 + e];
            const bool first = base + e == 0;
            if (first || kl.rc(me) != kl.rc(prev)) hmask |= 1u << q;
            if (first || me != prev) dmask |= 1u << q;
            prev = me;
        }
    }
    uint32_t total;
    const uint32_t excl = block_excl_scan((uint32_t)__popc(hmask) | ((uint32_t)__popc(dmask) << 16), s_wave, total);
    const uint32_t n_heads = total & 0xffffu, n_dist = total >> 16;
    uint32_t r = excl & 0xffffu, xd = excl >> 16;
#pragma unroll
    for (int q = 0; q < FD_ITEMS; q++) {
        if (hmask & (1u << q)) { s_hx[r] = (uint16_t)xd; s_he[r] = (uint16_t)(e0 + q); r++; }
        if (dmask & (1u << q)) xd++;
    }
    if (threadIdx.x == 0) s_hx[n_heads] = (uint16_t)n_dist;
    __syncthreads();
    const unsigned long long out = off[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < n_heads; i += FD_BLOCK) {
        const K me = tile[1 + s_he[i]];
        const int32_t cnt = (int32_t)s_hx[i + 1] - (int32_t)s_hx[i];
        const unsigned long long d = out + i;
        row[d] = (int32_t)kl.row(me); col[d] = (int32_t)kl.cell(me);
        if (i + 1 == n_heads) atomicAdd(&val[d], cnt); else val[d] = cnt;  // the last run may continue in later tiles
    }
    if (threadIdx.x == 0) {
        const uint32_t lead = n_heads ? s_hx[0] : n_dist;
        if (lead && out > 0) atomicAdd(&val[out - 1], (int32_t)lead);
    }
}

// Same fold for keys that are sorted by (row, cell) ONLY (the UMI bits were left out of the radix sort: 4 passes instead
// of 7).  Inside a run the UMIs are in arbitrary order, so "distinct" is decided by an LDS hash set: the tile first
// inserts the part of its leading run that lies in earlier tiles (its lead-in, at most FU_LEAD keys), then its own keys -
// a key is counted by the one thread whose compare-and-swap claims the slot.  A key therefore counts in the tile that
// holds its first occurrence and nowhere else.  A run with a longer lead-in raises *giant and the host redoes the fold on
// fully sorted keys.  64-bit keys only.
constexpr int FU_LEAD = 4096, FU_SLOTS = 8192;                           // <= 2048 + 4096 keys in 8192 slots (64 KB, dynamic LDS)
// 1024 threads per tile: the 72 KB of LDS allow two blocks per CU, and with 256-thread blocks the two waves per SIMD could not
// hide the chains of LDS atomics (4.3 ms for 380 M keys; 3.0 ms with 512 threads, 2.8 ms with 1024)
constexpr int FU_BLOCK = 1024, FU_ITEMS = FD_TILE / FU_BLOCK, FU_STRIDE = FU_LEAD / FU_BLOCK;
static_assert(FU_ITEMS * FU_BLOCK == FD_TILE && FU_STRIDE * FU_BLOCK == FU_LEAD && FU_STRIDE <= 64, "lead-in search / sweep layout");
__global__ __launch_bounds__(FU_BLOCK) void k_fold_emit_unsorted(const unsigned long long* __restrict__ k, long long n, KeyLayout<unsigned long long> kl,
                                                                 const unsigned long long* __restrict__ off,
                                                                 int32_t* __restrict__ row, int32_t* __restrict__ col, int32_t* __restrict__ val,
                                                                 unsigned long long* __restrict__ giant, int sub_shift) {
    // sub_shift = lowest key bit the radix sort covered (<= kl.ubits): the sort's whole digits usually reach a few bits into
    // the UMI field, so a (row, cell) run is a sequence of sub-runs by UMI prefix - and only the sub-run of the tile's first
    // key can have keys (hence duplicates) in earlier tiles.  That is what the lead-in covers.
    typedef unsigned long long K;
    extern __shared__ K set[];                                            // FU_SLOTS slots (a tile with a short lead-in uses half)
    __shared__ uint16_t s_hx[FD_TILE + 1], s_he[FD_TILE];                 // per head: distinct keys before it / its element
    __shared__ uint32_t s_wh[FU_BLOCK / 64], s_wd[FU_BLOCK / 64];
    __shared__ long long s_lead;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long base = (long long)blockIdx.x * FD_TILE;
    const int n_loc = (int)min((long long)FD_TILE, n - base);
    // the tile's own keys (and each key's predecessor) are requested first: the loads fly during the lead-in search and
    // the table set-up.  Wave w owns FD_TILE / 8 consecutive elements in FU_ITEMS coalesced sweeps of 64; element order = (wave, sweep, lane).
    K me[FU_ITEMS], pv[FU_ITEMS];
    const int e_w = wave * (FD_TILE / (FU_BLOCK / 64));
#pragma unroll
    for (int q = 0; q < FU_ITEMS; q++) {
        const int e = e_w + q * 64 + lane;
        me[q] = ~0ull; pv[q] = ~0ull;
        if (e < n_loc) { me[q] = k[base + e]; if (base + e > 0) pv[q] = k[base + e - 1]; }
    }
    const unsigned long long out = off[blockIdx.x];
    // ---- lead-in of the first run: keys of earlier tiles with the (row, cell) of the tile's first key (a suffix of
    //      what precedes the tile).  One coarse probe per thread, FU_STRIDE keys apart, then FU_STRIDE fine ones: two L2 round trips.
    const K rc0 = k[base] >> sub_shift;
    const long long w0 = max(0ll, base - FU_LEAD);
    long long lead0 = base;
    if (base > 0 && (k[base - 1] >> sub_shift) == rc0) {                      // block-uniform
        if (tid == 0) s_lead = base - 1;
        __syncthreads();
        const long long g = base - 1 - (long long)FU_STRIDE * tid;
        if (g >= w0 && (k[g] >> sub_shift) == rc0) atomicMin((unsigned long long*)&s_lead, (unsigned long long)g);
        __syncthreads();
        const long long c = s_lead;                                       // smallest coarse match: the run starts in (c - FU_STRIDE, c]
        __syncthreads();
        if (tid < FU_STRIDE) { const long long g2 = c - tid; if (g2 >= w0 && (k[g2] >> sub_shift) == rc0) atomicMin((unsigned long long*)&s_lead, (unsigned long long)g2); }
        __syncthreads();
        lead0 = s_lead;
        if (tid == 0 && lead0 == w0 && w0 > 0 && (k[w0 - 1] >> sub_shift) == rc0) *giant = 1ull;   // the sub-run starts before the window
    }
    const int slots = ((int)(base - lead0) + n_loc <= 3072) ? FU_SLOTS / 2 : FU_SLOTS;      // block-uniform; load factor <= 0.75
    const uint32_t smask = (uint32_t)slots - 1;
    for (int t = tid; t < slots; t += FU_BLOCK) set[t] = ~0ull;
    __syncthreads();
    auto probe_on = [&](K key, uint32_t slot) -> bool {                  // continue after a first probe that hit another key
        for (;;) {
            slot = (slot + 1) & smask;
            const K prev = atomicCAS(&set[slot], ~0ull, key);
            if (prev == ~0ull) return true;
            if (prev == key) return false;
        }
    };
    for (long long g = lead0 + tid; g < base; g += FU_BLOCK) {
        const K key = k[g]; const uint32_t sl = set_slot<FU_SLOTS>(key) & smask;
        const K prev = atomicCAS(&set[sl], ~0ull, key);
        if (prev != ~0ull && prev != key) (void)probe_on(key, sl);
    }
    __syncthreads();
    // ---- own keys
    K got[FU_ITEMS]; uint32_t sl[FU_ITEMS]; bool hd[FU_ITEMS];
#pragma unroll
    for (int q = 0; q < FU_ITEMS; q++) {
        const int e = e_w + q * 64 + lane;
        hd[q] = false; sl[q] = 0; got[q] = 0;
        if (e < n_loc) {
            hd[q] = base + e == 0 || kl.rc(me[q]) != kl.rc(pv[q]);
            sl[q] = set_slot<FU_SLOTS>(me[q]) & smask;
        }
    }
#pragma unroll
    for (int q = 0; q < FU_ITEMS; q++)                                    // independent LDS atomics in flight per lane
        if (e_w + q * 64 + lane < n_loc) got[q] = atomicCAS(&set[sl[q]], ~0ull, me[q]);
    unsigned long long hb[FU_ITEMS], db[FU_ITEMS];
    uint32_t th = 0, td = 0;
#pragma unroll
    for (int q = 0; q < FU_ITEMS; q++) {
        bool dist = false;
        if (e_w + q * 64 + lane < n_loc) dist = got[q] == ~0ull ? true : (got[q] == me[q] ? false : probe_on(me[q], sl[q]));   // first occurrence of this (row, cell, UMI)
        hb[q] = __ballot(hd[q]); db[q] = __ballot(dist);
        th += (uint32_t)__popcll(hb[q]); td += (uint32_t)__popcll(db[q]);
    }
    if (lane == 0) { s_wh[wave] = th; s_wd[wave] = td; }
    __syncthreads();
    uint32_t r = 0, xd = 0, n_heads = 0, n_dist = 0;
#pragma unroll
    for (int w = 0; w < FU_BLOCK / 64; w++) { if (w < wave) { r += s_wh[w]; xd += s_wd[w]; } n_heads += s_wh[w]; n_dist += s_wd[w]; }
    const unsigned long long lt = (1ull << lane) - 1;
#pragma unroll
    for (int q = 0; q < FU_ITEMS; q++) {
        if (hd[q]) { const uint32_t rr = r + (uint32_t)__popcll(hb[q] & lt); s_hx[rr] = (uint16_t)(xd + (uint32_t)__popcll(db[q] & lt)); s_he[rr] = (uint16_t)(e_w + q * 64 + lane); }
        r += (uint32_t)__popcll(hb[q]); xd += (uint32_t)__popcll(db[q]);
    }
    if (tid == 0) s_hx[n_heads] = (uint16_t)n_dist;
    __syncthreads();
    for (uint32_t i = tid; i < n_heads; i += FU_BLOCK) {
        const K key = k[base + s_he[i]];
        const int32_t cnt = (int32_t)s_hx[i + 1] - (int32_t)s_hx[i];
        const unsigned long long d = out + i;
        row[d] = (int32_t)kl.row(key); col[d] = (int32_t)kl.cell(key);
        if (i + 1 == n_heads) atomicAdd(&val[d], cnt); else val[d] = cnt;  // the last run may continue in later tiles
    }
    if (tid == 0) {
        const uint32_t lead = n_heads ? s_hx[0] : n_dist;
        if (lead && out > 0) atomicAdd(&val[out - 1], (int32_t)lead);
    }
}

constexpr long long RUN_WALK = 64;     // a run head walks at most this many followers itself; longer runs go to k_first_long
__device__ __forceinline__ int nib_bucket(int nib) { return nib == 1 ? 0 : nib == 2 ? 1 : nib == 4 ? 2 : nib == 8 ? 3 : 4; }

// BAF step 1: per (snp, cell, umi) run keep the value with the smallest ordinal (first read in
// fetch order, baf/fc/mcount.py:118-119); tally its allele per SNP (mcount.py:140-150).
template <class K>
__global__ void k_first_read(const K* __restrict__ k, const uint64_t* __restrict__ v, long long n, KeyLayout<K> kl,
                             uint8_t* __restrict__ al_out, uint32_t* __restrict__ tally, unsigned long long* __restrict__ long_runs) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    K me = k[i];
    if (i > 0 && k[i - 1] == me) { al_out[i] = 0; return; }
    uint64_t best = v[i];
    long long j = i + 1;
    for (; j < n && j <= i + RUN_WALK && k[j] == me; j++) { uint64_t x = v[j]; if (x < best) best = x; }
    if (j < n && j > i + RUN_WALK && k[j] == me) { long_runs[1 + atomicAdd(&long_runs[0], 1ull)] = (unsigned long long)i; return; }   // finished by k_first_long
    uint32_t code = uint32_t(best & ((1u << ALLELE_BITS) - 1));      // nibble + 1, 0 = no base
    al_out[i] = (uint8_t)code;
    if (code) atomicAdd(&tally[(size_t)kl.row(me) * 5 + nib_bucket(int(code) - 1)], 1u);
}

// A key run longer than RUN_WALK (a molecule with many reads over one SNP: constant UMI tags, UMI-less deep pileups) is not
// walked by its head lane: the head only queues it, and here ONE BLOCK per run finds the run's end by bisection on the sorted
// keys and takes the minimum (ordinal, allele) value in parallel.  SPLIT: k_first_base's outputs, else k_first_read's.
template <class K, bool SPLIT>
__global__ void __launch_bounds__(256) k_first_long(const K* __restrict__ k, const uint64_t* __restrict__ v, long long n, KeyLayout<K> kl,
                                                      const unsigned long long* __restrict__ long_runs, uint8_t* __restrict__ al_out,
                                                      uint64_t* __restrict__ ord_out, uint32_t* __restrict__ tally) {
    __shared__ unsigned long long s_min[4];
    const unsigned long long n_long = long_runs[0];
    for (unsigned long long r = blockIdx.x; r < n_long; r += gridDim.x) {
        const long long h = (long long)long_runs[1 + r];
        const K me = k[h];
        long long lo = h + 1, hi = n;                                     // first index past the run (every lane bisects: same loads, broadcast by the cache)
        while (lo < hi) { const long long mid = lo + ((hi - lo) >> 1); if (k[mid] == me) lo = mid + 1; else hi = mid; }
        unsigned long long best = ~0ull;
        for (long long j = h + threadIdx.x; j < lo; j += blockDim.x) { const unsigned long long x = v[j]; if (x < best) best = x; }
        for (int d = 32; d; d >>= 1) { const unsigned long long o = __shfl_xor(best, d); if (o < best) best = o; }
        if ((threadIdx.x & 63) == 0) s_min[threadIdx.x >> 6] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; w++) if (s_min[w] < best) best = s_min[w];
            const uint32_t code = uint32_t(best & ((1u << ALLELE_BITS) - 1));
            al_out[h] = (uint8_t)code;
            if (SPLIT) ord_out[h] = best >> ALLELE_BITS;
            else if (code) atomicAdd(&tally[(size_t)kl.row(me) * 5 + nib_bucket(int(code) - 1)], 1u);
        }
        __syncthreads();
    }
}

// ---- split mode (64-bit keys): the sorted stream holds only hits WITH a base; the hits without one are looked up ----
// Filter in front of the exact lookups: per (molecule = cell | UMI, block of 32 SNPs) ONE hashed 64-bit word that holds the
// block's SNPs at which the molecule shows a base twice - SNP offset o as bit (o + r1) & 31 of the low half and bit (o + r2) & 31 of
// the high half, r1 / r2 two rotations hashed from the molecule.  A gap record loads the word, rotates the halves back and ANDs
// them: a foreign molecule in the same word must hit the same offset under both of ITS rotations (~0.4 % per SNP), and only the
// SNPs left in the mask are looked up exactly.  One atomic per key run, one load per (record, block) whatever the gap's length.
// The table is laid out ALONG THE SORTED STREAM: the entries of SNP block b go to the words [lo_b, lo_b + len_b), lo_b / len_b =
// the block's index range in the sorted keys (k_blk_bounds) - one word per key, whatever the depth of the block.  The inserts of
// k_first_base therefore land next to the keys being read, and the gap records - which arrive in position order - ask a window of
// the table that moves along with them and stays in L2 (a table hashed over all of its 230 MB cost one HBM round trip per record).
__device__ __forceinline__ void bloom_slot(unsigned long long cellumi, uint32_t blk, unsigned long long lo, unsigned long long len, unsigned long long& word, uint32_t& r1, uint32_t& r2) {
    unsigned long long x = cellumi ^ ((unsigned long long)blk * 0x9E3779B97F4A7C15ull);
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    word = lo + (unsigned long long)(((x & 0xffffffffull) * (len & 0xffffffffull)) >> 32);   // len < 2^32 (checked on the host)
    r1 = (uint32_t)(x >> 32) & 31u; r2 = (uint32_t)(x >> 40) & 31u;
}
// first index of every block of 32 SNPs in the sorted keys (blk_lo[n_blk] = n): one bisection per block
template <class K>
__global__ void k_blk_bounds(const K* __restrict__ k, long long n, KeyLayout<K> kl, uint32_t n_blk, unsigned long long* __restrict__ blk_lo) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b > n_blk) return;
    if (b == n_blk) { blk_lo[b] = (unsigned long long)n; return; }
    long long lo = 0, hi = n;
    const uint32_t row0 = b << 5;
    while (lo < hi) { const long long mid = lo + ((hi - lo) >> 1); if (kl.row(k[mid]) < row0) lo = mid + 1; else hi = mid; }
    blk_lo[b] = (unsigned long long)lo;
}
// per key run: allele code + ordinal of its first read with a base, at the run head; first / one-past-last index of every SNP
template <class K>
__global__ void k_first_base(const K* __restrict__ k, const uint64_t* __restrict__ v, long long n, KeyLayout<K> kl,
                             uint8_t* __restrict__ al_out, uint64_t* __restrict__ ord_out, unsigned long long* __restrict__ row_lo, unsigned long long* __restrict__ row_hi,
                             unsigned long long* __restrict__ bloom, const unsigned long long* __restrict__ blk_lo, unsigned long long* __restrict__ long_runs) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // The kernel waits for memory 91 % of its wave cycles (profiles/r03_X_sq_counters_resident.txt): a wave's time is the number of
    // DEPENDENT load stages.  Everything a short run needs - the neighbours' keys and the next two values - is therefore loaded up front,
    // without looking at the keys first (the lines are the neighbouring lanes' own), and the filter block's bounds as soon as the row is known.
    const long long ip = i > 0 ? i - 1 : 0, i1 = i + 1 < n ? i + 1 : n - 1, i2 = i + 2 < n ? i + 2 : n - 1;
    const K kp = k[ip], me = k[i], kn1 = k[i1], kn2 = k[i2];
    const uint64_t v0 = v[i], v1 = v[i1], v2 = v[i2];
    const uint32_t row = kl.row(me);
    const unsigned long long b_lo = blk_lo[row >> 5], b_len = blk_lo[(row >> 5) + 1] - b_lo;   // (this key is inside: b_len >= 1)
    if (i == 0 || kl.row(kp) != row) row_lo[row] = (unsigned long long)i;
    if (i + 1 == n || kl.row(kn1) != row) row_hi[row] = (unsigned long long)(i + 1);
    if (i > 0 && kp == me) { al_out[i] = 0; return; }
    uint64_t best = v0;
    long long j = i + 1;
    if (j < n && kn1 == me) {
        if (v1 < best) best = v1;
        j = i + 2;
        if (j < n && kn2 == me) {
            if (v2 < best) best = v2;
            for (j = i + 3; j < n && j <= i + RUN_WALK && k[j] == me; j++) { uint64_t x = v[j]; if (x < best) best = x; }
        }
    }
    if (j < n && j > i + RUN_WALK && k[j] == me) long_runs[1 + atomicAdd(&long_runs[0], 1ull)] = (unsigned long long)i;   // al / ord of this head: k_first_long
    else {
        al_out[i] = (uint8_t)(best & ((1u << ALLELE_BITS) - 1));      // nibble + 1 (never 0 here)
        ord_out[i] = best >> ALLELE_BITS;
    }
    unsigned long long word; uint32_t r1, r2;
    const unsigned long long cu = (unsigned long long)(me & ((K(1) << (kl.cbits + kl.ubits)) - 1));
    bloom_slot(cu, row >> 5, b_lo, b_len, word, r1, r2);
    atomicOr(&bloom[word], (1ull << ((row + r1) & 31u)) | (1ull << (32u + ((row + r2) & 31u))));
}
// every gap record (first SNP, cell, UMI | ordinal, count - 1): for each of its SNPs, if (SNP, cell, UMI) has a run and
// this read comes EARLIER in fetch order than the run's first read with a base, the key belongs to this read
// (baf/fc/mcount.py:118-119) and the run contributes nothing.  The Bloom filter answers "no" for almost every record;
// the exact lookups that remain are binary searches inside one SNP's run (the stream is in tile order, so
// neighbouring threads search the same few SNPs and stay in L2).
constexpr int CL_U = 4;                    // gap records per thread
template <class K>
__global__ void __launch_bounds__(256) k_claim(const K* __restrict__ nk, const uint64_t* __restrict__ nv, unsigned long long cap, ShardSpan sp, unsigned long long n_units,
                                                 const K* __restrict__ keys, KeyLayout<K> kl, const unsigned long long* __restrict__ row_lo, const unsigned long long* __restrict__ row_hi,
                                                 const uint64_t* __restrict__ ord, uint8_t* __restrict__ al,
                                                 const unsigned long long* __restrict__ bloom, const unsigned long long* __restrict__ blk_lo, uint32_t n_rows,
                                                 const uint32_t* __restrict__ p_rowtab, const uint32_t* __restrict__ p_end) {
    const int low = kl.cbits + kl.ubits;
    // unit u = records [256 * CL_U * (u / NSHARD), + 256 * CL_U) of shard slice u % NSHARD: the slices are filled round-robin by consecutive
    // join tiles, so the blocks in flight together hold ONE narrow position range of the file (and one window of the filter table).
    // The kernel is a chain of dependent loads per record (record -> block bounds -> filter word, then for the few survivors cell bounds
    // -> ~9 bisection probes -> ordinal), and nearly every wave holds a survivor: a thread therefore takes CL_U records - all their filter
    // probes are in flight together (a gap covers at most 32 SNPs: two blocks), and the survivors of all of them are looked up in ONE
    // loop whose trip count is the most survivors any lane has, not a loop per record.
    for (unsigned long long u = blockIdx.x; u < n_units; u += gridDim.x) {
        const int sh = (int)(u % NSHARD);
        const unsigned long long n_sh = sp.start[sh + 1] - sp.start[sh], idx0 = (u / NSHARD) * (256 * CL_U) + threadIdx.x;
        K cu[CL_U]; uint64_t ordn[CL_U]; uint32_t k1[CL_U], m[CL_U][2];
#pragma unroll
        for (int q = 0; q < CL_U; q++) {
            const unsigned long long idx = idx0 + (unsigned long long)q * 256;
            cu[q] = 0; ordn[q] = 0; k1[q] = 0; m[q][0] = m[q][1] = 0;
            if (idx >= n_sh) continue;
            const unsigned long long j = (unsigned long long)sh * cap + idx;
            const K rec = __builtin_nontemporal_load(&nk[j]);           // (read once: keep the L2 for the table window and the key look-ups)
            const uint64_t v = __builtin_nontemporal_load(&nv[j]);
            ordn[q] = v >> ALLELE_BITS;
            k1[q] = kl.row(rec);
            const uint32_t k2 = min(k1[q] + (uint32_t)(v & ((1u << ALLELE_BITS) - 1)) + 1u, n_rows);
            cu[q] = rec & ((K(1) << low) - 1);
#pragma unroll
            for (int b = 0; b < 2; b++) {                               // (at most 32 SNPs from k1: the block of k1 and the next)
                const uint32_t blk = (k1[q] >> 5) + b;
                if (blk > (k2 - 1) >> 5) continue;
                const unsigned long long b_lo = blk_lo[blk], b_len = blk_lo[blk + 1] - b_lo;
                if (!b_len) continue;                                   // no read shows a base anywhere in this block of SNPs
                unsigned long long word; uint32_t r1, r2;
                bloom_slot((unsigned long long)cu[q], blk, b_lo, b_len, word, r1, r2);
                const uint32_t o0 = max(k1[q], blk << 5) & 31u, o1 = (min(k2, (blk + 1) << 5) - 1u) & 31u;     // the gap's SNPs inside this block: offsets o0 .. o1
                const unsigned long long w = bloom[word];
                const uint32_t h1 = (uint32_t)w, h2 = (uint32_t)(w >> 32);
                m[q][b] = ((h1 >> r1) | (h1 << ((32u - r1) & 31u))) & ((h2 >> r2) | (h2 << ((32u - r2) & 31u))) & ((0xffffffffu >> (31u - o1)) & (0xffffffffu << o0));
            }
        }
        // SNPs at which one of this thread's molecules (or one that shares its filter word) shows a base
        for (;;) {
            int sq = -1, sb = 0;
#pragma unroll
            for (int q = CL_U - 1; q >= 0; q--) { if (m[q][1]) { sq = q; sb = 1; } if (m[q][0]) { sq = q; sb = 0; } }
            if (sq < 0) break;
            K cellumi = 0; uint64_t my_ord = 0; uint32_t my_k1 = 0, mm = 0;
#pragma unroll
            for (int q = 0; q < CL_U; q++) if (q == sq) { cellumi = cu[q]; my_ord = ordn[q]; my_k1 = k1[q]; mm = m[q][sb]; }
            const uint32_t srow = (((my_k1 >> 5) + (uint32_t)sb) << 5) + (uint32_t)__builtin_ctz(mm);
            mm &= mm - 1;
#pragma unroll
            for (int q = 0; q < CL_U; q++) if (q == sq) m[q][sb] = mm;
            // the key's place: inside its (SNP, cell group) cell of the partition sort where there was one (<= 2048 entries: ~9 probes
            // on a few lines; a hot SNP's whole range is 100 k entries), else inside the SNP's range
            unsigned long long lo, hi;
            if (p_rowtab) {
                const uint32_t t = p_rowtab[srow], z = (t >> 5) + ((uint32_t)((unsigned long long)cellumi >> kl.ubits) >> (t & 31u));
                lo = z ? p_end[z - 1] : 0u; hi = p_end[z];
            } else { lo = row_lo[srow]; hi = row_hi[srow]; }
            if (lo >= hi) continue;
            const unsigned long long end = hi;
            const K key = (K(srow) << low) | cellumi;
            while (lo < hi) { const unsigned long long mid = (lo + hi) >> 1; if (keys[mid] < key) lo = mid + 1; else hi = mid; }
            if (lo < end && keys[lo] == key && my_ord < ord[lo]) al[lo] = 0;   // benign race: every writer stores 0
        }
    }
}
// per-SNP allele tallies (baf/fc/mcount.py:140-150) of the keys that kept an allele.  The keys are sorted by SNP and k_first_base has
// left every SNP's index range, so nothing is searched or added atomically: EIGHT LANES PER SNP walk the SNP's slice of `al` (one
// byte per key; neighbouring SNPs are neighbouring slices, so a wave reads one contiguous stretch), three shuffles put the five
// counts together and lanes 0..4 of the group store them.  (The first form - one thread per key, ballots per wave segment, atomics
// on the SNP's counters - spent 0.77 ms at configs[2] queueing on counter lines shared by neighbouring SNPs.)
// A SNP deeper than TALLY_LONG keys (a hot gene: 100 k keys at configs[2]; bulk input) is queued in pieces of TALLY_PIECE keys, one
// block per piece (k_tally_long), which add their counts to the SNP's (zeroed) counters.
constexpr unsigned long long TALLY_LONG = 2048, TALLY_PIECE = 4096;
__device__ __forceinline__ void tally_code(uint32_t code, uint32_t (&c)[5]) {
    const int b = code ? nib_bucket(int(code) - 1) : -1;
#pragma unroll
    for (int q = 0; q < 5; q++) c[q] += (b == q) ? 1u : 0u;
}
__global__ void __launch_bounds__(256) k_tally_rows(const uint8_t* __restrict__ al, const unsigned long long* __restrict__ row_lo, const unsigned long long* __restrict__ row_hi,
                                                    uint32_t n_rows, uint32_t* __restrict__ tally, unsigned long long* __restrict__ long_rows) {
    const unsigned long long gid = (unsigned long long)blockIdx.x * 256 + threadIdx.x, s = gid >> 3; const uint32_t sub = (uint32_t)gid & 7u;
    unsigned long long lo = 0, len = 0;
    if (s < n_rows) { lo = row_lo[s]; const unsigned long long hi = row_hi[s]; len = hi > lo ? hi - lo : 0; }
    const bool is_long = len > TALLY_LONG;
    if (is_long) {
        if (sub == 0) {
            const unsigned long long np = (len + TALLY_PIECE - 1) / TALLY_PIECE, at = atomicAdd(&long_rows[0], np);
            for (unsigned long long q = 0; q < np; q++) long_rows[1 + at + q] = (s << 32) | q;      // (at most len / 2048 entries per SNP: the list holds n / 64)
        }
        len = 0;
    }
    uint32_t c[5] = {0, 0, 0, 0, 0};
    for (unsigned long long t = sub; t < len; t += 8) tally_code(al[lo + t], c);
#pragma unroll
    for (int q = 0; q < 5; q++) { c[q] += __shfl_xor(c[q], 1); c[q] += __shfl_xor(c[q], 2); c[q] += __shfl_xor(c[q], 4); }
    if (s < n_rows && !is_long && sub < 5) tally[(size_t)s * 5 + sub] = sub == 0 ? c[0] : sub == 1 ? c[1] : sub == 2 ? c[2] : sub == 3 ? c[3] : c[4];
}
__global__ void __launch_bounds__(256) k_tally_long(const uint8_t* __restrict__ al, const unsigned long long* __restrict__ row_lo, const unsigned long long* __restrict__ row_hi,
                                                    const unsigned long long* __restrict__ long_rows, uint32_t* __restrict__ tally) {
    __shared__ uint32_t s_c[4][5];
    const unsigned long long n_long = long_rows[0];
    for (unsigned long long r = blockIdx.x; r < n_long; r += gridDim.x) {
        const unsigned long long ent = long_rows[1 + r];
        const size_t s = (size_t)(ent >> 32);
        const unsigned long long lo = row_lo[s] + (ent & 0xffffffffull) * TALLY_PIECE, hi = min(row_hi[s], lo + TALLY_PIECE);
        uint32_t c[5] = {0, 0, 0, 0, 0};
        for (unsigned long long t = lo + threadIdx.x; t < hi; t += 256) tally_code(al[t], c);
#pragma unroll
        for (int q = 0; q < 5; q++) { for (int d = 32; d; d >>= 1) c[q] += __shfl_xor(c[q], d); }
        if ((threadIdx.x & 63) == 0) { for (int q = 0; q < 5; q++) s_c[threadIdx.x >> 6][q] = c[q]; }
        __syncthreads();
        if (threadIdx.x < 5) { const uint32_t v = s_c[0][threadIdx.x] + s_c[1][threadIdx.x] + s_c[2][threadIdx.x] + s_c[3][threadIdx.x]; if (v) atomicAdd(&tally[s * 5 + threadIdx.x], v); }
        __syncthreads();
    }
}

struct SnpFilter { int32_t min_count; double min_maf; };

// plp_snp() filters, baf/fc/core.py:238-246.  info: ref nibble | alt nibble << 4 | ref_hap << 8 | alt_hap << 9
__device__ __forceinline__ bool snp_passes(const uint32_t* tally, const uint32_t* info, uint32_t s, SnpFilter f) {
    const uint32_t* t = tally + (size_t)s * 5;
    uint32_t tot = t[0] + t[1] + t[2] + t[3] + t[4];
    if ((int64_t)tot < (int64_t)f.min_count) return false;
    uint32_t inf = info[s];
    uint32_t a = t[nib_bucket(inf & 15)], b = t[nib_bucket((inf >> 4) & 15)];
    uint32_t minor = a < b ? a : b;
    if ((double)minor < (double)tot * f.min_maf) return false;
    return true;
}

// BAF step 2: expand each surviving (snp, cell, umi, allele) to the regions that contain the SNP
// (baf/fc/main.py:92-101, core.py:156-166).  COUNT pass sums the fan-out, EMIT pass writes.
template <class K, bool EMIT, class V>
__global__ __launch_bounds__(JOIN_BLOCK) void k_expand(const K* __restrict__ k, const uint8_t* __restrict__ al, long long n,
                                  KeyLayout<K> kl, const uint32_t* __restrict__ tally, const uint32_t* __restrict__ info,
                                  SnpFilter f, const int32_t* __restrict__ csr_off, const int32_t* __restrict__ csr_reg,
                                  K* __restrict__ k2, V* __restrict__ v2, unsigned long long* ctl, XBases xb, int pack_shift) {
    // pack_shift >= 0 (emit pass, 64-bit keys): no values are written - the haplotype class goes into two UMI-field bits that no UMI code
    // uses (k_join ORs the UMI codes of the reads it accepts into ctl_umi_or; the host finds the free bits)
    __shared__ uint32_t s_wave[JOIN_BLOCK / 64];
    __shared__ unsigned long long s_base;
    long long i = (long long)blockIdx.x * JOIN_BLOCK + threadIdx.x;
    uint32_t cnt = 0, s = 0, code = 0; K me = 0;
    int32_t c0 = 0, c1 = 0;
    if (i < n) {
        // (the kernel waits for memory 83 % of its wave cycles: key and allele code are loaded together, and the SNP's region list bounds
        // together with its tallies - three dependent stages instead of five)
        code = al[i]; me = k[i]; s = kl.row(me);
        c0 = csr_off[s]; c1 = csr_off[s + 1];
        if (code && snp_passes(tally, info, s, f)) cnt = uint32_t(c1 - c0);
    }
    uint32_t total;
    uint32_t excl = block_excl_scan(cnt, s_wave, total);
    if (total == 0) return;
    // count pass: per-shard totals in word 0 of the shard's line; emit pass: per-shard cursor in word 1, inside the
    // shard's slice [xb.base[sh], xb.base[sh] + total_sh) of the output (the order of k2 is irrelevant: it is sorted next)
    const int sh = blockIdx.x & (XSHARD - 1);
    if (threadIdx.x == 0) s_base = xb.base[sh] + atomicAdd(&ctl[CTL_X0 + sh * CTL_STRIDE + (EMIT ? 1 : 0)], (unsigned long long)total);
    if (!EMIT) return;
    __syncthreads();
    if (!cnt) return;
    unsigned long long dst = s_base + excl;
    uint32_t inf = info[s];
    int nib = int(code) - 1;
    int idx = -1;                                        // snp.gt = {ref: ref_idx, alt: alt_idx}: alt wins if equal
    if (nib == int(inf & 15)) idx = int((inf >> 8) & 1);
    if (nib == int((inf >> 4) & 15)) idx = int((inf >> 9) & 1);
    const V bits = idx == 0 ? 1 : idx == 1 ? 2 : 4;
    uint32_t cell = kl.cell(me); uint64_t umi = kl.umi(me);
    if (pack_shift >= 0) umi |= (uint64_t)(idx == 0 ? 0u : idx == 1 ? 1u : 2u) << pack_shift;
    for (int32_t c = c0; c < c1; c++, dst++) {
        k2[dst] = kl.make((uint32_t)csr_reg[c], cell, umi);
        if (pack_shift < 0) v2[dst] = bits;
    }
}

// BAF step 3: haplotype set algebra per (row, cell) run, baf/fc/core.py:173-192, without any thread walking a (row, cell) run
// (a SMART-seq cell holds hundreds of thousands of read names in its most expressed gene; one run = one thread would serialise).
//   k_hap_class: head of every (row, cell, UMI) run: OR of the run's haplotype bits (one entry per SNP the molecule meets: a few)
//   k_hap_sum  : per 2048-key tile, ONE block scan of four packed counters (REF-hap, ALT-hap, either, other-only keys); a run
//                that starts in the tile gets its counts up to the tile end, what later tiles hold of it arrives by atomicAdd
//   k_hap_count / k_hap_scatter: the no_dup_hap arithmetic on the per-run sums -> AD / DP / OTH, compacted into COO
template <class K, class V>
__global__ void k_hap_class(const K* __restrict__ k, const V* __restrict__ v, long long n, uint8_t* __restrict__ cls, unsigned long long* __restrict__ long_runs) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const K me = k[i];
    if (i > 0 && k[i - 1] == me) { cls[i] = 0; return; }
    uint32_t bits = (uint32_t)v[i];
    long long j = i + 1;
    for (; j < n && j <= i + RUN_WALK && k[j] == me; j++) bits |= (uint32_t)v[j];
    // a molecule that meets more than RUN_WALK SNPs of one region (constant UMI tags, bulk input: thousands) is not walked by its
    // head lane: the head queues the run and one block per run finishes it (k_hap_class_long)
    if (j < n && j > i + RUN_WALK && k[j] == me) { cls[i] = 4; long_runs[1 + atomicAdd(&long_runs[0], 1ull)] = (unsigned long long)i; return; }
    cls[i] = (uint8_t)bits;                                               // 1 REF haplotype, 2 ALT haplotype, 4 other allele; never 0 at a head
}
template <class K, class V>
__global__ void __launch_bounds__(256) k_hap_class_long(const K* __restrict__ k, const V* __restrict__ v, long long n, const unsigned long long* __restrict__ long_runs, uint8_t* __restrict__ cls) {
    __shared__ uint32_t s_or[4];
    const unsigned long long n_long = long_runs[0];
    for (unsigned long long r = blockIdx.x; r < n_long; r += gridDim.x) {
        const long long h = (long long)long_runs[1 + r];
        const K me = k[h];
        long long lo = h + 1, hi = n;                                     // first index past the run
        while (lo < hi) { const long long mid = lo + ((hi - lo) >> 1); if (k[mid] == me) lo = mid + 1; else hi = mid; }
        uint32_t bits = 0;
        for (long long j = h + threadIdx.x; j < lo; j += blockDim.x) bits |= (uint32_t)v[j];
        for (int d = 32; d; d >>= 1) bits |= __shfl_xor(bits, d);
        if ((threadIdx.x & 63) == 0) s_or[threadIdx.x >> 6] = bits;
        __syncthreads();
        if (threadIdx.x == 0) cls[h] = (uint8_t)(s_or[0] | s_or[1] | s_or[2] | s_or[3]);
        __syncthreads();
    }
}

__device__ __forceinline__ unsigned long long block_excl_scan64(unsigned long long v, unsigned long long* s_wave, unsigned long long& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const unsigned long long t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    unsigned long long base = 0; total = 0;
#pragma unroll
    for (int w = 0; w < FD_BLOCK / 64; w++) { const unsigned long long t = s_wave[w]; if (w < wave) base += t; total += t; }
    __syncthreads();
    return base + inc - v;
}

template <class K>
__global__ __launch_bounds__(FD_BLOCK) void k_hap_sum(const K* __restrict__ k, const uint8_t* __restrict__ cls, long long n, KeyLayout<K> kl,
                                                      const unsigned long long* __restrict__ off, K* __restrict__ run_key, uint32_t* __restrict__ acc, long long stride) {
    // acc[f * stride + run]: f = 0 REF-hap keys, 1 ALT-hap keys, 2 keys on either haplotype, 3 keys with only another allele; zero on entry
    __shared__ uint32_t s_wave[FD_BLOCK / 64];
    __shared__ unsigned long long s_wave64[FD_BLOCK / 64];
    __shared__ K tile[FD_TILE + 1];                                       // tile[0] = key before the tile (halo)
    __shared__ unsigned long long s_x[FD_TILE + 1];                       // per head: packed counters before it (4 x 16 bits: a tile holds 2048 keys)
    __shared__ uint16_t s_he[FD_TILE];
    const long long base = (long long)blockIdx.x * FD_TILE;
    const int n_loc = (int)min((long long)FD_TILE, n - base);
#pragma unroll
    for (int t = 0; t < FD_ITEMS; t++) { const int e = t * FD_BLOCK + threadIdx.x; if (e < n_loc) tile[1 + e] = k[base + e]; }
    if (threadIdx.x == 0) tile[0] = base > 0 ? k[base - 1] : K(0);
    __syncthreads();
    const int e0 = threadIdx.x * FD_ITEMS;                                // blocked: thread t owns elements [t * FD_ITEMS, (t + 1) * FD_ITEMS)
    uint32_t hmask = 0; unsigned long long c[FD_ITEMS], sum = 0;
    K prev = tile[e0];
#pragma unroll
    for (int q = 0; q < FD_ITEMS; q++) {
        const int e = e0 + q; c[q] = 0;
        if (e < n_loc) {
            const K me = tile[1 + e];
            if (base + e == 0 || kl.rc(me) != kl.rc(prev)) hmask |= 1u << q;
            const uint32_t bits = cls[base + e];
            c[q] = (unsigned long long)(bits & 1u) | ((unsigned long long)((bits >> 1) & 1u) << 16) | ((unsigned long long)((bits & 3u) ? 1u : 0u) << 32)
                 | ((unsigned long long)((!(bits & 3u) && (bits & 4u)) ? 1u : 0u) << 48);
            sum += c[q];
            prev = me;
        }
    }
    uint32_t n_heads; unsigned long long tot;
    uint32_t r = block_excl_scan((uint32_t)__popc(hmask), s_wave, n_heads);
    unsigned long long xd = block_excl_scan64(sum, s_wave64, tot);
#pragma unroll
    for (int q = 0; q < FD_ITEMS; q++) {
        if (hmask & (1u << q)) { s_x[r] = xd; s_he[r] = (uint16_t)(e0 + q); r++; }
        xd += c[q];
    }
    if (threadIdx.x == 0) s_x[n_heads] = tot;
    __syncthreads();
    const unsigned long long out = off[blockIdx.x];
    auto add = [&](unsigned long long d, unsigned long long x, bool atomic) {
#pragma unroll
        for (int f = 0; f < 4; f++) { const uint32_t val = (uint32_t)((x >> (16 * f)) & 0xffffu); if (!val) continue;
            if (atomic) atomicAdd(&acc[(size_t)f * stride + d], val); else acc[(size_t)f * stride + d] = val; }
    };
    for (uint32_t i = threadIdx.x; i < n_heads; i += FD_BLOCK) {
        const unsigned long long d = out + i;
        run_key[d] = tile[1 + s_he[i]];
        add(d, s_x[i + 1] - s_x[i], i + 1 == n_heads);                    // monotone fields: the packed difference needs no borrow; the last run may continue
    }
    if (threadIdx.x == 0) {
        const unsigned long long lead = n_heads ? s_x[0] : tot;          // keys of the run that the previous tiles started
        if (lead && out > 0) add(out - 1, lead, true);
    }
}

// Control words go to the host through MAPPED pinned memory written by a tiny kernel, never through the
// DMA engines: a 4 KB hipMemcpy D2H would queue behind a 170 MB result copy-out of another engine.
__global__ void k_publish(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ host_alias, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) host_alias[i] = src[i];
}

__global__ void k_copy_words(const int32_t* __restrict__ src, int32_t* __restrict__ host_alias, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) host_alias[i] = src[i];
}
struct CopySeg3 { const int32_t* src[3]; int32_t* dst[3]; size_t n[3]; };
__global__ void k_copy_words3(CopySeg3 sg) {                              // blockIdx.y = segment
    const int32_t* __restrict__ src = sg.src[blockIdx.y]; int32_t* __restrict__ dst = sg.dst[blockIdx.y]; const size_t n = sg.n[blockIdx.y];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// shard slices of (key, value) pairs -> contiguous arrays, one launch (the shard count would be that many copy commands)
template <class K>
__global__ void __launch_bounds__(256) k_pack_pairs(const K* __restrict__ sk, const uint64_t* __restrict__ sv, unsigned long long cap, ShardSpan sp,
                                                      K* __restrict__ dk, uint64_t* __restrict__ dv) {
    const unsigned long long n = sp.start[NSHARD];
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) {
        int sh = 0;
#pragma unroll
        for (int q = 1; q < NSHARD; q++) sh += (i >= sp.start[q]) ? 1 : 0;
        const unsigned long long j = (unsigned long long)sh * cap + (i - sp.start[sh]);
        dk[i] = sk[j]; dv[i] = sv[j];
    }
}

// ordered compaction of the haplotype matrices into COO -------------------------------------------
constexpr int CP_BLOCK = 256, CP_ITEMS = 8, CP_TILE = CP_BLOCK * CP_ITEMS;

// single block: exclusive scan of nb block counts (64-bit running sum), total -> out_total
__global__ __launch_bounds__(1024) void k_cp_scan(const uint32_t* __restrict__ blk, long long nb, unsigned long long* __restrict__ off,
                                                  unsigned long long* out_total) {
    __shared__ unsigned long long s_w[16];
    __shared__ unsigned long long s_carry;
    blk += (long long)blockIdx.x * nb; off += (long long)blockIdx.x * nb; out_total += blockIdx.x;   // one block per matrix
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (long long b0 = 0; b0 < nb; b0 += 1024) {
        long long i = b0 + threadIdx.x;
        unsigned long long v = i < nb ? blk[i] : 0, inc = v;
        int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        for (int d = 1; d < 64; d <<= 1) { unsigned long long t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
        if (lane == 63) s_w[w] = inc;
        __syncthreads();
        unsigned long long wb = 0, tot = 0;
        for (int x = 0; x < 16; x++) { if (x < w) wb += s_w[x]; tot += s_w[x]; }
        unsigned long long carry = s_carry;
        if (i < nb) off[i] = carry + wb + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) s_carry = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *out_total = s_carry;
}

struct CooOut3 { int32_t* o[3]; unsigned long long total[3]; };       // per matrix: [row | col | val] block and its nnz
// The haplotype matrices straight from the per-run sums: the no_dup_hap arithmetic (baf/fc/core.py:173-192) is done by the count pass
// and by the scatter pass instead of going through three dense arrays (written once, read twice, 2 x the runs long because the arrays
// were sized for the keys).  Eight consecutive runs per thread (two 16-byte loads per sum array; `stride` is a multiple of 8 and the
// arrays are zero beyond the runs, so nothing is bounds-checked per element); tiles beyond the runs leave at once.  One block scan
// of the three counts packed into one 64-bit word.  Output order = run order = (row, cell) order.
struct HapSrc { const uint32_t* acc; long long stride; const unsigned long long* n_runs; long long n_fixed; int no_dup_hap; const unsigned long long* packed; };
// runs: *n_runs, or n_fixed (staging area with holes) when n_runs is null.  packed != null: the four sums of a run as 4 x 16 bits of one
// word (k_hap_items: an item holds at most 2048 keys), instead of the four 32-bit arrays `acc` (k_hap_sum: a run can be millions of keys)
__device__ __forceinline__ void hap_load8(const HapSrc& h, long long i0, int32_t (&ad)[CP_ITEMS], int32_t (&dp)[CP_ITEMS], int32_t (&oth)[CP_ITEMS]) {
    static_assert(CP_ITEMS == 8, "two uint4 per array");
    uint32_t a[4][CP_ITEMS];
    if (h.packed) {
        const uint4* p = reinterpret_cast<const uint4*>(h.packed + i0);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint4 x = p[q];
            a[0][2 * q] = x.x & 0xffffu; a[1][2 * q] = x.x >> 16; a[2][2 * q] = x.y & 0xffffu; a[3][2 * q] = x.y >> 16;
            a[0][2 * q + 1] = x.z & 0xffffu; a[1][2 * q + 1] = x.z >> 16; a[2][2 * q + 1] = x.w & 0xffffu; a[3][2 * q + 1] = x.w >> 16;
        }
    } else {
#pragma unroll
        for (int f = 0; f < 4; f++) {
            const uint4* p = reinterpret_cast<const uint4*>(h.acc + (size_t)f * h.stride + i0);
            const uint4 x = p[0], y = p[1];
            a[f][0] = x.x; a[f][1] = x.y; a[f][2] = x.z; a[f][3] = x.w; a[f][4] = y.x; a[f][5] = y.y; a[f][6] = y.z; a[f][7] = y.w;
        }
    }
#pragma unroll
    for (int t = 0; t < CP_ITEMS; t++) {
        int32_t ref = (int32_t)a[0][t], alt = (int32_t)a[1][t], d = (int32_t)a[2][t]; const int32_t ot = (int32_t)a[3][t];
        if (ref + alt != d) {
            if (h.no_dup_hap) { const int32_t share = ref + alt - d; ref -= share; alt -= share; }
            d = ref + alt;
        }
        const bool keep = d + ot > 0;
        ad[t] = keep && alt > 0 ? alt : 0; dp[t] = keep && d > 0 ? d : 0; oth[t] = keep && ot > 0 ? ot : 0;
    }
}
__global__ __launch_bounds__(CP_BLOCK) void k_hap_count(HapSrc h, uint32_t* __restrict__ blk) {
    __shared__ unsigned long long s_w[CP_BLOCK / 64];
    const long long n = h.n_runs ? (long long)*h.n_runs : h.n_fixed, i0 = (long long)blockIdx.x * CP_TILE + (long long)threadIdx.x * CP_ITEMS;
    if ((long long)blockIdx.x * CP_TILE >= n) { if (threadIdx.x < 3) blk[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = 0; return; }
    unsigned long long c = 0;
    if (i0 < n) {
        int32_t ad[CP_ITEMS], dp[CP_ITEMS], oth[CP_ITEMS];
        hap_load8(h, i0, ad, dp, oth);
#pragma unroll
        for (int t = 0; t < CP_ITEMS; t++) c += (unsigned long long)(ad[t] > 0) | ((unsigned long long)(dp[t] > 0) << 21) | ((unsigned long long)(oth[t] > 0) << 42);
    }
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x < 3) { const unsigned long long t = s_w[0] + s_w[1] + s_w[2] + s_w[3]; blk[(size_t)threadIdx.x * gridDim.x + blockIdx.x] = (uint32_t)(t >> (21 * threadIdx.x)) & 0x1fffffu; }
}
template <class K>
__global__ __launch_bounds__(CP_BLOCK) void k_hap_scatter(HapSrc h, const K* __restrict__ k, KeyLayout<K> kl, const unsigned long long* __restrict__ off, CooOut3 out) {
    __shared__ unsigned long long s_w[FD_BLOCK / 64];
    __shared__ int32_t s_row[CP_TILE], s_col[CP_TILE], s_val[CP_TILE];    // one matrix's entries of this tile, compacted: they leave with coalesced stores
    static_assert(FD_BLOCK == CP_BLOCK, "block_excl_scan64 is written for FD_BLOCK threads");
    const long long n = h.n_runs ? (long long)*h.n_runs : h.n_fixed, i0 = (long long)blockIdx.x * CP_TILE + (long long)threadIdx.x * CP_ITEMS;
    if ((long long)blockIdx.x * CP_TILE >= n) return;
    int32_t v[3][CP_ITEMS];
    unsigned long long c = 0;
    if (i0 < n) {
        hap_load8(h, i0, v[0], v[1], v[2]);
#pragma unroll
        for (int t = 0; t < CP_ITEMS; t++) c += (unsigned long long)(v[0][t] > 0) | ((unsigned long long)(v[1][t] > 0) << 21) | ((unsigned long long)(v[2][t] > 0) << 42);
    } else {
#pragma unroll
        for (int t = 0; t < CP_ITEMS; t++) v[0][t] = v[1][t] = v[2][t] = 0;
    }
    unsigned long long total;
    const unsigned long long excl = block_excl_scan64(c, s_w, total);
    int32_t krow[CP_ITEMS], kcol[CP_ITEMS];
#pragma unroll
    for (int t = 0; t < CP_ITEMS; t++) {                                 // (`k` holds one key per staging entry and is at least stride long; only entries with a value are used)
        krow[t] = 0; kcol[t] = 0;
        if (c && (v[0][t] > 0 || v[1][t] > 0 || v[2][t] > 0)) { const K key = k[i0 + t]; krow[t] = (int32_t)kl.row(key); kcol[t] = (int32_t)kl.cell(key); }
    }
    // (a thread's entries sit next to each other, eight threads' worth apart: stored straight from the registers they kept the address
    // unit busy 62 % of the wave cycles)
#pragma unroll
    for (int y = 0; y < 3; y++) {
        uint32_t d = (uint32_t)((excl >> (21 * y)) & 0x1fffffull);
        const uint32_t tot = (uint32_t)((total >> (21 * y)) & 0x1fffffull);
#pragma unroll
        for (int t = 0; t < CP_ITEMS; t++)
            if (v[y][t] > 0) { s_row[d] = krow[t]; s_col[d] = kcol[t]; s_val[d] = v[y][t]; d++; }
        __syncthreads();
        int32_t* __restrict__ row = out.o[y]; int32_t* __restrict__ col = row + out.total[y]; int32_t* __restrict__ val = col + out.total[y];
        const unsigned long long base = off[(size_t)y * gridDim.x + blockIdx.x];
        for (uint32_t j = threadIdx.x; j < tot; j += CP_BLOCK) { row[base + j] = s_row[j]; col[base + j] = s_col[j]; val[base + j] = s_val[j]; }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct ContigTab { int32_t reg_base = 0, n_reg = 0, snp_base = 0, n_snp = 0, swin_base = 0, n_swin = 0; };

struct BatchSlot {
    int32_t* pos = nullptr; uint16_t* flag = nullptr; uint8_t* mapq = nullptr; int32_t* cell = nullptr;
    uint64_t* umi = nullptr; uint32_t* cig_off = nullptr; uint32_t* cigar = nullptr; uint32_t* seq_off = nullptr; uint8_t* seq = nullptr;
    size_t cap_reads = 0, cap_cig = 0, cap_seq = 0;
    bool busy = false;
};

// grow-only device workspace: finish() sub-allocates from it instead of hipMalloc/hipFree per call
struct Arena {
    char* base = nullptr; size_t cap = 0, off = 0;
    template <class T> T* get(size_t n) {
        off = (off + 255) & ~size_t(255);
        T* p = reinterpret_cast<T*>(base + off);
        off += std::max<size_t>(n, 1) * sizeof(T);
        return off <= cap ? p : nullptr;
    }
};

// Device staging of decoded chunks for the host ingest (engine_push_block): the decoder hands over one pinned block per
// chunk, it is copied with ONE hipMemcpyAsync into one of three slots, and every pipeline of the handle (basefc and pileup of a
// fused handle) reads the same copy.  A slot is reused once every launch that reads it has been confirmed (its hits fitted).
struct Stager {
    int device = 0; hipStream_t s_copy = nullptr;
    struct Slot { char* buf = nullptr; size_t cap = 0; hipEvent_t t0 = nullptr, copied = nullptr; int users = 0; bool timed = false; } slot[3];
    int next = 0;
};

struct EngineImpl {
    xck_engine* eng = nullptr;
    int mode = 0, device = 0;
    int key_bits = 64, ubits = 0, cbits = 0, rbits = 0;
    ReadFilter rf{};
    SnpFilter sf{};
    int no_dup_hap = 1;
    int n_cells = 0, n_regions = 0, n_snps_sorted = 0;
    std::vector<ContigTab> ctab;
    // device tables
    int32_t *d_reg_s0 = nullptr, *d_reg_e0 = nullptr, *d_reg_row = nullptr, *d_reg_pmax = nullptr;   // basefc: regions per contig sorted by start
    int32_t *d_snp_p0 = nullptr, *d_snp_win = nullptr, *d_csr_off = nullptr, *d_csr_reg = nullptr;
    uint32_t *d_snp_info = nullptr, *d_tally = nullptr;
    hipStream_t s_copy = nullptr, s_comp = nullptr;
    BatchSlot slot[2];
    int next_slot = 0;
    int64_t max_batch_reads = 0;
    // hit accumulators (ping-pong pair so that sort results can stay where they land)
    void* d_keys = nullptr; uint64_t* d_vals = nullptr; size_t hit_cap = 0;
    void* d_nkeys = nullptr; uint64_t* d_nvals = nullptr;   // pileup split mode: hits without a base (same per-shard capacity)
    unsigned long long ncur[NSHARD] = {0}, ncur_before[NSHARD] = {0}, ncursor = 0;
    unsigned long long* d_ctl = nullptr;       // CTL_WORDS control words (overflow flag, scratch, sharded cursors)
    unsigned long long* h_ctl = nullptr;       // pinned + mapped mirror
    unsigned long long* d_hctl = nullptr;      // device alias of h_ctl (written by k_publish)
    unsigned long long cur[NSHARD] = {0};      // host view of the shard cursors after the last completed launch
    unsigned long long cur_before[NSHARD] = {0}, acc_before[NSHARD] = {0};
    unsigned long long cursor = 0;             // sum of cur[]; hit_cap is the capacity of ONE shard
    int fold_extra_digits = 0;                 // basefc hash fold: extra radix digits that earlier finishes needed (giant runs)
    int fold_path = 0, fold_fallbacks = 0;     // xck_stats: which basefc fold ran last (1 partition, 2 radix sort), hand-overs so far
    int pileup_sort_path = 0;                  // pileup hits of the last finish: 1 sorted by partition + LDS sort, 2 by the radix sort
    int pileup_sort2_path = 0;                 // ... and its region-level hits
    int fold_refinements = 0;                  // partition fold of the last finish: refinements of the level-2 geometry
    // fused launch queue
    std::vector<BatchDesc> queue;              // not yet launched (device-resident pushes are deferred)
    std::vector<BatchDesc> inflight;           // launched, not yet confirmed (kept for overflow replay)
    int inflight_slot = -1;
    int inflight_shared = -1;                  // staging slot (Stager) the launch in flight reads, -1 = none
    int64_t queued_reads = 0, inflight_reads = 0;
    TileMeta* d_meta = nullptr; size_t meta_cap = 0;
    // timing
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_res = nullptr, ev_c0 = nullptr, ev_c1 = nullptr;
    hipEvent_t ev_f1 = nullptr, ev_f2 = nullptr;   // partition fold: level-1 bucket pass on the copy stream (fold_partition.h)
    bool copy_timed = false, copy_pending = false;
    xck_stats st{};
    int64_t n_join_launches = 0;
    bool fold_failed = false;             // xck_finish returned an error from inside a fold: only xck_reset makes the handle usable again
    unsigned long long stamp_sum[12] = {0}; int stamp_tiles = 0; float stamp_ms = 0;   // XCK_STAMPS builds: phase cycles of the last join launch
    // workspace + results
    Arena ws1, ws2;
    int32_t* h_res[4] = {nullptr, nullptr, nullptr, nullptr}; size_t h_res_cap[4] = {0, 0, 0, 0}; size_t res_nnz[4] = {0, 0, 0, 0};
    int32_t* d_res[4] = {nullptr, nullptr, nullptr, nullptr};   // device copies [row | col | val] inside the workspace, valid until the next finish / reset
    bool finished = false;
};

static size_t key_bytes(const EngineImpl* im) { return im->key_bits == 64 ? 8 : 16; }

template <class T> static int dev_upload(EngineImpl* im, T** dptr, const std::vector<T>& h) {
    size_t n = std::max<size_t>(h.size(), 1);
    HIP_TRY(hipMalloc((void**)dptr, n * sizeof(T)));
    if (!h.empty()) HIP_TRY(hipMemcpy(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

static uint32_t nib_of(uint8_t ch) {
    switch (ch) { case 'A': return 1; case 'C': return 2; case 'G': return 4; case 'T': return 8; default: return 15; }
}

static int build_tables(EngineImpl* im, const xck_config* cfg) {
    const int nc = cfg->n_contigs;
    im->ctab.assign(std::max(nc, 1), ContigTab());
    std::vector<int32_t> reg_s0, reg_e0, reg_row, reg_pmax;
    std::vector<int32_t> snp_p0, snp_win, csr_off, csr_reg;
    std::vector<uint32_t> snp_info;
    if (im->mode == XCK_MODE_BASEFC) {
        // regions valid for fetch(): pysam raises (-> region silently gets 0, utils/sam.py:105-118)
        // when start-1 < 0 or start-1 > end.
        std::vector<std::vector<int32_t>> by_c(nc);
        for (int g = 0; g < cfg->n_regions; g++) {
            const xck_region& r = cfg->regions[g];
            if (r.contig < 0 || r.contig >= nc) continue;
            if (r.start < 1 || (int64_t)r.start - 1 > (int64_t)r.end) continue;
            by_c[r.contig].push_back(g);
        }
        for (int c = 0; c < nc; c++) {
            auto& v = by_c[c];
            std::sort(v.begin(), v.end(), [&](int32_t a, int32_t b) {
                const xck_region &x = cfg->regions[a], &y = cfg->regions[b];
                if (x.start != y.start) return x.start < y.start;
                if (x.end != y.end) return x.end < y.end;
                return a < b; });
            ContigTab& t = im->ctab[c];
            t.reg_base = (int32_t)reg_s0.size(); t.n_reg = (int32_t)v.size();
            int32_t max_e = 0;
            for (int32_t g : v) {
                const xck_region& r = cfg->regions[g];
                reg_s0.push_back(r.start - 1); reg_e0.push_back(r.end); reg_row.push_back(g);
                max_e = std::max(max_e, r.end); reg_pmax.push_back(max_e);          // running maximum of the ends: first candidate of a position by binary search
            }
            (void)max_e;
        }
    } else {
        std::vector<std::vector<int32_t>> by_c(nc);
        for (int s = 0; s < cfg->n_snps; s++) {
            const xck_snp& x = cfg->snps[s];
            if (x.contig < 0 || x.contig >= nc || x.pos < 1) continue;   // fetch(pos-1 < 0) raises -> no reads
            by_c[x.contig].push_back(s);
        }
        // regions per contig sorted by start for the SNP -> region join (baf/fc/main.py:92-101)
        std::vector<std::vector<int32_t>> reg_c(nc);
        for (int g = 0; g < cfg->n_regions; g++) { const xck_region& r = cfg->regions[g]; if (r.contig >= 0 && r.contig < nc) reg_c[r.contig].push_back(g); }
        std::vector<uint64_t> excl;                                       // (snp index << 32 | region index), sorted
        for (int i = 0; i < cfg->n_excl_pairs; i++) excl.push_back(((uint64_t)(uint32_t)cfg->excl_snp[i] << 32) | (uint32_t)cfg->excl_region[i]);
        std::sort(excl.begin(), excl.end());
        csr_off.push_back(0);
        for (int c = 0; c < nc; c++) {
            auto& v = by_c[c];
            std::sort(v.begin(), v.end(), [&](int32_t a, int32_t b) {
                if (cfg->snps[a].pos != cfg->snps[b].pos) return cfg->snps[a].pos < cfg->snps[b].pos; return a < b; });
            ContigTab& t = im->ctab[c];
            t.snp_base = (int32_t)snp_p0.size(); t.n_snp = (int32_t)v.size();
            for (int32_t s : v) {
                const xck_snp& x = cfg->snps[s];
                snp_p0.push_back(x.pos - 1);
                snp_info.push_back(nib_of(x.ref) | (nib_of(x.alt) << 4) | ((uint32_t)(x.ref_hap & 1) << 8) | ((uint32_t)(x.alt_hap & 1) << 9));
            }
            int32_t max_p = v.empty() ? 0 : cfg->snps[v.back()].pos;
            t.n_swin = v.empty() ? 0 : (max_p >> WSS) + 1;
            t.swin_base = (int32_t)snp_win.size();
            { int32_t k = 0; for (int32_t w = 0; w < t.n_swin; w++) { while (k < t.n_snp && snp_p0[t.snp_base + k] < (w << WSS)) k++; snp_win.push_back(t.snp_base + k); } }
            // SNP -> regions: start <= pos <= end_incl; rows ascending so keys stay deterministic
            // (minus the caller's exclusion pairs: SNPs that local phasing removed from one region's list)
            std::vector<std::vector<int32_t>> hits(t.n_snp);
            for (int32_t g : reg_c[c]) {
                const xck_region& r = cfg->regions[g];
                if (r.end < r.start) continue;
                auto lo = std::lower_bound(snp_p0.begin() + t.snp_base, snp_p0.begin() + t.snp_base + t.n_snp, r.start - 1);
                for (auto it = lo; it != snp_p0.begin() + t.snp_base + t.n_snp && *it <= r.end - 1; ++it) {
                    const size_t k = (size_t)((it - snp_p0.begin()) - t.snp_base);
                    if (!excl.empty() && std::binary_search(excl.begin(), excl.end(), ((uint64_t)(uint32_t)v[k] << 32) | (uint32_t)g)) continue;
                    hits[k].push_back(g);
                }
            }
            for (int32_t k = 0; k < t.n_snp; k++) { std::sort(hits[k].begin(), hits[k].end()); for (int32_t g : hits[k]) csr_reg.push_back(g); csr_off.push_back((int32_t)csr_reg.size()); }
        }
        im->n_snps_sorted = (int)snp_p0.size();
    }
    int rc;
    if ((rc = dev_upload(im, &im->d_reg_s0, reg_s0))) return rc;
    if ((rc = dev_upload(im, &im->d_reg_e0, reg_e0))) return rc;
    if ((rc = dev_upload(im, &im->d_reg_row, reg_row))) return rc;
    if ((rc = dev_upload(im, &im->d_reg_pmax, reg_pmax))) return rc;
    if ((rc = dev_upload(im, &im->d_snp_p0, snp_p0))) return rc;
    if ((rc = dev_upload(im, &im->d_snp_win, snp_win))) return rc;
    if ((rc = dev_upload(im, &im->d_snp_info, snp_info))) return rc;
    if ((rc = dev_upload(im, &im->d_csr_off, csr_off))) return rc;
    if ((rc = dev_upload(im, &im->d_csr_reg, csr_reg))) return rc;
    if (im->mode == XCK_MODE_BAF) {
        size_t n = std::max<size_t>((size_t)im->n_snps_sorted * 5, 1);
        HIP_TRY(hipMalloc((void**)&im->d_tally, n * sizeof(uint32_t)));
    }
    return 0;
}

static int arena_begin(EngineImpl* im, Arena& a, size_t need) {
    a.off = 0;
    if (need > a.cap) {
        if (a.base) HIP_TRY(hipFree(a.base));
        a.base = nullptr; a.cap = 0;
        size_t c = need + need / 4 + (1 << 20);
        HIP_TRY(hipMalloc((void**)&a.base, c));
        a.cap = c;
    }
    return 0;
}

static int res_reserve(EngineImpl* im, int m, size_t nnz) {
    if (nnz * 3 > im->h_res_cap[m]) {
        if (im->h_res[m]) HIP_TRY(hipHostFree(im->h_res[m]));
        im->h_res[m] = nullptr; im->h_res_cap[m] = 0;
        size_t c = nnz * 3 + nnz / 2 + 1024;
        HIP_TRY(hipHostMalloc((void**)&im->h_res[m], c * sizeof(int32_t), hipHostMallocMapped));
        im->h_res_cap[m] = c;
    }
    return 0;
}

// per-shard head room added to every capacity guess (XCK_HIT_SLACK: test knob that makes the overflow / replay path easy to reach)
static inline size_t hit_slack(const EngineImpl* im) { return (size_t)im->eng->knobs.hit_slack; }
static inline bool split_mode(const EngineImpl* im) { return XCK_BAF_SPLIT && im->mode == XCK_MODE_BAF && im->key_bits == 64; }

static int ensure_hits(EngineImpl* im, size_t need) {           // need = elements per shard
    if (need <= im->hit_cap) return 0;
    size_t ncap = std::max<size_t>(need, im->hit_cap * 2);
    void* nk = nullptr; uint64_t* nv = nullptr;
    HIP_TRY(hipMalloc(&nk, ncap * NSHARD * key_bytes(im)));
    if (im->mode == XCK_MODE_BAF) HIP_TRY(hipMalloc((void**)&nv, ncap * NSHARD * sizeof(uint64_t)));
    for (int sh = 0; sh < NSHARD; sh++) if (im->cur[sh]) {
        HIP_TRY(hipMemcpyAsync((char*)nk + (size_t)sh * ncap * key_bytes(im), (char*)im->d_keys + (size_t)sh * im->hit_cap * key_bytes(im),
                               im->cur[sh] * key_bytes(im), hipMemcpyDeviceToDevice, im->s_comp));
        if (nv) HIP_TRY(hipMemcpyAsync(nv + (size_t)sh * ncap, im->d_vals + (size_t)sh * im->hit_cap, im->cur[sh] * sizeof(uint64_t), hipMemcpyDeviceToDevice, im->s_comp));
    }
    void* nnk = nullptr; uint64_t* nnv = nullptr;
    if (split_mode(im)) {
        HIP_TRY(hipMalloc(&nnk, ncap * NSHARD * sizeof(uint64_t))); HIP_TRY(hipMalloc((void**)&nnv, ncap * NSHARD * sizeof(uint64_t)));
        for (int sh = 0; sh < NSHARD; sh++) if (im->ncur[sh]) {
            HIP_TRY(hipMemcpyAsync((uint64_t*)nnk + (size_t)sh * ncap, (uint64_t*)im->d_nkeys + (size_t)sh * im->hit_cap, im->ncur[sh] * sizeof(uint64_t), hipMemcpyDeviceToDevice, im->s_comp));
            HIP_TRY(hipMemcpyAsync(nnv + (size_t)sh * ncap, im->d_nvals + (size_t)sh * im->hit_cap, im->ncur[sh] * sizeof(uint64_t), hipMemcpyDeviceToDevice, im->s_comp));
        }
    }
    HIP_TRY(hipStreamSynchronize(im->s_comp));
    if (im->d_keys) HIP_TRY(hipFree(im->d_keys));
    if (im->d_vals) HIP_TRY(hipFree(im->d_vals));
    if (im->d_nkeys) HIP_TRY(hipFree(im->d_nkeys));
    if (im->d_nvals) HIP_TRY(hipFree(im->d_nvals));
    im->d_keys = nk; im->d_vals = nv; im->d_nkeys = nnk; im->d_nvals = nnv; im->hit_cap = ncap;
    return 0;
}

static int slot_reserve(EngineImpl* im, BatchSlot& s, size_t n_reads, size_t n_cig, size_t n_seq) {
    if (n_reads > s.cap_reads) {
        size_t c = std::max(n_reads, s.cap_reads * 2);
        if (s.pos) { hipFree(s.pos); hipFree(s.flag); hipFree(s.mapq); hipFree(s.cell); hipFree(s.umi); hipFree(s.cig_off); hipFree(s.seq_off); }
        HIP_TRY(hipMalloc((void**)&s.pos, c * 4)); HIP_TRY(hipMalloc((void**)&s.flag, c * 2)); HIP_TRY(hipMalloc((void**)&s.mapq, c));
        HIP_TRY(hipMalloc((void**)&s.cell, c * 4)); HIP_TRY(hipMalloc((void**)&s.umi, c * 8));
        HIP_TRY(hipMalloc((void**)&s.cig_off, (c + 1) * 4)); HIP_TRY(hipMalloc((void**)&s.seq_off, (c + 1) * 4));
        s.cap_reads = c;
    }
    if (n_cig > s.cap_cig) { size_t c = std::max(n_cig, s.cap_cig * 2); if (s.cigar) hipFree(s.cigar); HIP_TRY(hipMalloc((void**)&s.cigar, c * 4)); s.cap_cig = c; }
    if (n_seq > s.cap_seq) { size_t c = std::max(n_seq, s.cap_seq * 2); if (s.seq) hipFree(s.seq); HIP_TRY(hipMalloc((void**)&s.seq, c)); s.cap_seq = c; }
    return 0;
}

// launch ONE fused join kernel over every batch in im->inflight
template <class K>
static int launch_join_t(EngineImpl* im) {
    const int nb = (int)im->inflight.size();
    JoinArgs<K> a;
    int32_t tiles = 0;
    for (int i = 0; i < nb; i++) { im->inflight[i].tile0 = tiles; tiles += (im->inflight[i].n + TILE - 1) / TILE; a.bt.desc[i] = im->inflight[i]; }
    a.bt.n_batches = nb; a.bt.n_tiles = tiles;
    if ((size_t)tiles > im->meta_cap) {
        if (im->d_meta) HIP_TRY(hipFree(im->d_meta));
        im->d_meta = nullptr; im->meta_cap = 0;
        size_t c = (size_t)tiles + tiles / 2 + 1024;
        HIP_TRY(hipMalloc((void**)&im->d_meta, c * sizeof(TileMeta)));
        im->meta_cap = c;
    }
    a.meta = im->d_meta; a.f = im->rf;
    a.reg_s0 = im->d_reg_s0; a.reg_e0 = im->d_reg_e0; a.reg_row = im->d_reg_row; a.reg_pmax = im->d_reg_pmax;
    a.snp_p0 = im->d_snp_p0;
    a.kl.ubits = im->ubits; a.kl.cbits = im->cbits;
    a.keys = (K*)im->d_keys; a.vals = im->d_vals; a.cap = im->hit_cap; a.ctl = im->d_ctl;
    a.nkeys = (K*)im->d_nkeys; a.nvals = im->d_nvals;
    dim3 grid(tiles), block(JOIN_BLOCK);
    clear_stale_error("launch_join", im->eng->knobs.debug_timing);
    HIP_TRY(hipEventRecord(im->ev0, im->s_comp));
    const dim3 mgrid((tiles + 255) / 256), mblock(256);
    if (im->mode == XCK_MODE_BASEFC) {
        hipLaunchKernelGGL((k_tile_meta<XCK_MODE_BASEFC>), mgrid, mblock, 0, im->s_comp, a.bt, im->d_meta, (const int32_t*)im->d_reg_pmax);
        hipLaunchKernelGGL((k_join<K, XCK_MODE_BASEFC>), grid, block, 0, im->s_comp, a);
    } else {
        hipLaunchKernelGGL((k_tile_meta<XCK_MODE_BAF>), mgrid, mblock, 0, im->s_comp, a.bt, im->d_meta, (const int32_t*)nullptr);
        hipLaunchKernelGGL((k_join<K, XCK_MODE_BAF>), grid, block, 0, im->s_comp, a);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(im->ev1, im->s_comp));
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(256), 0, im->s_comp, (const unsigned long long*)im->d_ctl, im->d_hctl, CTL_WORDS);
    HIP_TRY(hipGetLastError());
    return 0;
}
static int launch_join(EngineImpl* im) { return im->key_bits == 64 ? launch_join_t<uint64_t>(im) : launch_join_t<u128>(im); }

// wait for the launch in flight, collect cursor / timing; if its fragments did not fit, grow, rewind and replay
static int complete_pending(EngineImpl* im) {
    while (!im->inflight.empty()) {
        HIP_TRY(hipStreamSynchronize(im->s_comp));
        float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, im->ev0, im->ev1));
        im->st.ms_join += ms; im->st.ms_device += ms; im->n_join_launches++;
#if XCK_STAMPS
        { int32_t tiles = 0; for (auto& b : im->inflight) tiles += (b.n + TILE - 1) / TILE;
          std::vector<uint32_t> h((size_t)tiles * 12);
          HIP_TRY(hipMemcpy(h.data(), im->d_meta, h.size() * 4, hipMemcpyDeviceToHost));
          for (int q = 0; q < 12; q++) im->stamp_sum[q] = 0;
          for (size_t t = 0; t < (size_t)tiles; t++) for (int q = 0; q < 12; q++) im->stamp_sum[q] += h[t * 12 + q];
          im->stamp_tiles = tiles; im->stamp_ms = ms; }
#endif
        if (im->h_ctl[CTL_OVERFLOW]) {                       // some fragment did not fit: grow, rewind, replay
            unsigned long long mx = 0;
            for (int sh = 0; sh < NSHARD; sh++) mx = std::max(mx, std::max(im->h_ctl[ctl_cursor(sh)], im->h_ctl[ctl_ncursor(sh)]));
            int rc = ensure_hits(im, std::max<size_t>(mx + mx / 4 + 65536, im->hit_cap * 2)); if (rc) return rc;
            for (int sh = 0; sh < NSHARD; sh++) { im->h_ctl[ctl_cursor(sh)] = im->cur_before[sh]; im->h_ctl[ctl_ncursor(sh)] = im->ncur_before[sh];
                                                  im->h_ctl[ctl_accepted(sh)] = im->acc_before[sh]; }
            im->h_ctl[CTL_OVERFLOW] = 0;
            HIP_TRY(hipMemcpyAsync(im->d_ctl, im->h_ctl, CTL_WORDS * sizeof(unsigned long long), hipMemcpyHostToDevice, im->s_comp));
            HIP_TRY(hipStreamSynchronize(im->s_comp));
            rc = launch_join(im); if (rc) return rc;
            continue;
        }
        im->cursor = 0; im->ncursor = 0;
        for (int sh = 0; sh < NSHARD; sh++) { im->cur[sh] = im->h_ctl[ctl_cursor(sh)]; im->cursor += im->cur[sh];
                                              im->ncur[sh] = split_mode(im) ? im->h_ctl[ctl_ncursor(sh)] : 0; im->ncursor += im->ncur[sh]; }
        if (im->inflight_slot >= 0) im->slot[im->inflight_slot].busy = false;
        if (im->inflight_shared >= 0 && im->eng->stager) { ((Stager*)im->eng->stager)->slot[im->inflight_shared].users--; im->inflight_shared = -1; }
        im->inflight.clear(); im->inflight_slot = -1; im->inflight_reads = 0;
    }
    return 0;
}

// launch whatever is queued (after the previous launch has been confirmed)
static int launch_queue(EngineImpl* im, int slot_idx, int shared_slot = -1) {
    if (im->queue.empty()) return 0;
    int rc = complete_pending(im); if (rc) return rc;
    { unsigned long long mx = 0;
      for (int sh = 0; sh < NSHARD; sh++) mx = std::max(mx, std::max(im->cur[sh], im->ncur[sh]));
      // first guess: 1.25 keys per queued read (after the LDS de-duplication a 10x run leaves ~0.8); a launch that needs
      // more sets the overflow flag and is replayed into grown buffers, and the capacity is kept for the next pass
      rc = ensure_hits(im, mx + (size_t)im->queued_reads * 5 / 4 / NSHARD + hit_slack(im)); if (rc) return rc; }
    im->inflight.swap(im->queue); im->queue.clear();
    im->inflight_reads = im->queued_reads; im->queued_reads = 0;
    im->inflight_slot = slot_idx;
    im->inflight_shared = shared_slot;
    if (shared_slot >= 0) ((Stager*)im->eng->stager)->slot[shared_slot].users++;
    for (int sh = 0; sh < NSHARD; sh++) { im->cur_before[sh] = im->cur[sh]; im->ncur_before[sh] = im->ncur[sh]; im->acc_before[sh] = im->h_ctl[ctl_accepted(sh)]; }
    if (slot_idx >= 0) im->slot[slot_idx].busy = true;
    return launch_join(im);
}

int engine_push(xck_engine* e, const xck_batch* b, bool device_resident) {
    EngineImpl* im = (EngineImpl*)e->impl;
    if (!im) { e->err = "decode-only handle: no GPU engine behind it"; return XCK_E_STATE; }
    if (im->finished) { e->err = "push after finish (call xck_reset)"; return XCK_E_STATE; }
    im->st.n_batches++; im->st.n_reads += b->n_reads;
    if (b->n_reads <= 0 || b->contig < 0) return 0;
    if (b->contig >= (int)im->ctab.size()) { e->err = "batch contig out of range"; return XCK_E_ARG; }
    // (host batches: a null cigar / seq pointer is fine when its offset range is empty - the same rule as xck_push_batch's check and
    // its packed form; device-resident batches: the offsets live in HBM, so the pointers must be there)
    if (!b->pos || !b->flag || !b->mapq || !b->cell || !b->umi || !b->cig_off) { e->err = "null batch array"; return XCK_E_ARG; }
    if (!b->cigar && (device_resident || b->cig_off[b->n_reads] != b->cig_off[0])) { e->err = "null batch array"; return XCK_E_ARG; }
    if (im->mode == XCK_MODE_BAF && (!b->seq_off || (!b->seq && (device_resident || b->seq_off[b->n_reads] != b->seq_off[0])))) { e->err = "BAF mode needs seq arrays"; return XCK_E_ARG; }
    HIP_TRY(hipSetDevice(im->device));
    const ContigTab& t = im->ctab[b->contig];
    bool has_targets = im->mode == XCK_MODE_BASEFC ? t.n_reg > 0 : t.n_snp > 0;
    if (!has_targets) return 0;
    BatchDesc d;
    memset(&d, 0, sizeof d);
    d.n = b->n_reads; d.ordinal_base = b->ordinal_base;
    d.reg_lo = t.reg_base; d.reg_hi = t.reg_base + t.n_reg;
    d.snp_win = im->d_snp_win + t.swin_base; d.n_swin = t.n_swin; d.snp_end = t.snp_base + t.n_snp;
    if (device_resident) {
        // deferred: consecutive device-resident batches are fused into one launch (>> 256 workgroups)
        d.pos = b->pos; d.flag = b->flag; d.mapq = b->mapq; d.cell = b->cell; d.umi = b->umi;
        d.cig_off = b->cig_off; d.cigar = b->cigar; d.seq_off = b->seq_off; d.seq = b->seq;
        im->queue.push_back(d); im->queued_reads += b->n_reads;
        if ((int)im->queue.size() >= MAX_FUSE) return launch_queue(im, -1);
        return 0;
    }
    int rc = launch_queue(im, -1); if (rc) return rc;          // keep launch order == push order
    size_t n = (size_t)b->n_reads;
    // offsets need not start at 0 (a batch may be a window into a larger decode buffer)
    const uint32_t c_lo = b->cig_off[0], s_lo = im->mode == XCK_MODE_BAF ? b->seq_off[0] : 0;
    if (b->cig_off[n] < c_lo || (im->mode == XCK_MODE_BAF && b->seq_off[n] < s_lo)) { e->err = "batch offsets are not monotonic"; return XCK_E_ARG; }
    uint32_t n_cig = b->cig_off[n] - c_lo, n_seq = im->mode == XCK_MODE_BAF ? b->seq_off[n] - s_lo : 0;
    int slot_idx = im->next_slot; im->next_slot ^= 1;
    BatchSlot& s = im->slot[slot_idx];
    if (s.busy) { rc = complete_pending(im); if (rc) return rc; }   // slot still feeds the launch in flight
    rc = slot_reserve(im, s, n, std::max<uint32_t>(n_cig, 1), std::max<uint32_t>(n_seq, 1)); if (rc) return rc;
    auto t0 = std::chrono::steady_clock::now();
    HIP_TRY(hipMemcpyAsync(s.pos, b->pos, n * 4, hipMemcpyHostToDevice, im->s_copy));
    HIP_TRY(hipMemcpyAsync(s.flag, b->flag, n * 2, hipMemcpyHostToDevice, im->s_copy));
    HIP_TRY(hipMemcpyAsync(s.mapq, b->mapq, n, hipMemcpyHostToDevice, im->s_copy));
    HIP_TRY(hipMemcpyAsync(s.cell, b->cell, n * 4, hipMemcpyHostToDevice, im->s_copy));
    HIP_TRY(hipMemcpyAsync(s.umi, b->umi, n * 8, hipMemcpyHostToDevice, im->s_copy));
    HIP_TRY(hipMemcpyAsync(s.cig_off, b->cig_off, (n + 1) * 4, hipMemcpyHostToDevice, im->s_copy));
    if (n_cig) HIP_TRY(hipMemcpyAsync(s.cigar, b->cigar + c_lo, (size_t)n_cig * 4, hipMemcpyHostToDevice, im->s_copy));
    if (im->mode == XCK_MODE_BAF) {
        HIP_TRY(hipMemcpyAsync(s.seq_off, b->seq_off, (n + 1) * 4, hipMemcpyHostToDevice, im->s_copy));
        if (n_seq) HIP_TRY(hipMemcpyAsync(s.seq, b->seq + s_lo, n_seq, hipMemcpyHostToDevice, im->s_copy));
    }
    HIP_TRY(hipStreamSynchronize(im->s_copy));                 // caller may reuse its arrays now
    im->st.ms_h2d += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    im->st.algo_bytes_join += (int64_t)n * 20 + (int64_t)n_cig * 4 + (int64_t)(n_seq / 2);
    d.pos = s.pos; d.flag = s.flag; d.mapq = s.mapq; d.cell = s.cell; d.umi = s.umi;
    d.cig_off = s.cig_off; d.cigar = s.cigar - c_lo; d.seq_off = s.seq_off; d.seq = s.seq - s_lo;   // rebased, never read below *_lo
    im->queue.push_back(d); im->queued_reads += b->n_reads;
    return launch_queue(im, slot_idx);                         // the kernel overlaps the caller's next decode + copy
}

// ---- host ingest: one decoded chunk = one H2D copy, shared by every pipeline of the handle ----
#define HIP_TRY_E(e_, expr)                                                                 \
    do { hipError_t e__ = (expr); if (e__ != hipSuccess) {                                  \
        char b_[512]; snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #expr,               \
                               hipGetErrorString(e__), __FILE__, __LINE__);                 \
        (e_)->err = b_; return XCK_E_DEVICE; } } while (0)

int engine_push_block(xck_engine* e, const void* host_base, size_t bytes, const xck_batch* batches, int n, void** fence) {
    if (e->n_impl <= 0) { e->err = "decode-only handle: no GPU engine behind it"; return XCK_E_STATE; }
    EngineImpl* im0 = (EngineImpl*)e->impls[0];
    HIP_TRY_E(e, hipSetDevice(im0->device));
    Stager* st = (Stager*)e->stager;
    if (!st) {
        st = new Stager(); st->device = im0->device; e->stager = st;
        HIP_TRY_E(e, hipStreamCreateWithFlags(&st->s_copy, hipStreamNonBlocking));
        for (auto& sl : st->slot) { HIP_TRY_E(e, hipEventCreate(&sl.t0)); HIP_TRY_E(e, hipEventCreate(&sl.copied)); }
    }
    const int si = st->next; st->next = (st->next + 1) % 3;
    Stager::Slot& sl = st->slot[si];
    for (int k = 0; k < e->n_impl && sl.users > 0; k++) {                // launches that still read this slot: confirm them
        EngineImpl* im = (EngineImpl*)e->impls[k];
        if (im->inflight_shared == si) { int rc = complete_pending(im); if (rc) return rc; }
    }
    if (sl.users != 0) { e->err = "internal: staging slot still in use"; return XCK_E_STATE; }
    if (sl.timed) { float ms = 0; if (hipEventElapsedTime(&ms, sl.t0, sl.copied) == hipSuccess) im0->st.ms_h2d += ms; sl.timed = false; }
    if (bytes > sl.cap) {
        if (sl.buf) HIP_TRY_E(e, hipFree(sl.buf));
        sl.buf = nullptr; sl.cap = 0;
        const size_t c = bytes + bytes / 4 + (1 << 20);
        HIP_TRY_E(e, hipMalloc((void**)&sl.buf, c));
        sl.cap = c;
    }
    if (!*fence) { hipEvent_t ev; HIP_TRY_E(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming)); *fence = (void*)ev; }
    HIP_TRY_E(e, hipEventRecord(sl.t0, st->s_copy));
    HIP_TRY_E(e, hipMemcpyAsync(sl.buf, host_base, bytes, hipMemcpyHostToDevice, st->s_copy));
    HIP_TRY_E(e, hipEventRecord(sl.copied, st->s_copy));
    HIP_TRY_E(e, hipEventRecord((hipEvent_t)*fence, st->s_copy));
    sl.timed = true;
    const char* hb = (const char*)host_base;
    auto dev = [&](const void* hp) { return (void*)(sl.buf + ((const char*)hp - hb)); };
    for (int k = 0; k < e->n_impl; k++) {
        EngineImpl* im = (EngineImpl*)e->impls[k];
        if (im->finished) { e->err = "push after finish (call xck_reset)"; return XCK_E_STATE; }
        int rc = launch_queue(im, -1); if (rc) return rc;                 // earlier device-resident pushes keep their order
        HIP_TRY_E(e, hipStreamWaitEvent(im->s_comp, sl.copied, 0));
        for (int i = 0; i < n; i++) {
            const xck_batch* b = &batches[i];
            im->st.n_batches++; im->st.n_reads += b->n_reads;
            if (b->n_reads <= 0 || b->contig < 0) continue;
            if (b->contig >= (int)im->ctab.size()) { e->err = "batch contig out of range"; return XCK_E_ARG; }
            const ContigTab& t = im->ctab[b->contig];
            if (!(im->mode == XCK_MODE_BASEFC ? t.n_reg > 0 : t.n_snp > 0)) continue;
            if (im->mode == XCK_MODE_BAF && (!b->seq_off || !b->seq)) { e->err = "BAF mode needs seq arrays"; return XCK_E_ARG; }
            BatchDesc d; memset(&d, 0, sizeof d);
            d.n = b->n_reads; d.ordinal_base = b->ordinal_base;
            d.reg_lo = t.reg_base; d.reg_hi = t.reg_base + t.n_reg;
            d.snp_win = im->d_snp_win + t.swin_base; d.n_swin = t.n_swin; d.snp_end = t.snp_base + t.n_snp;
            d.pos = (const int32_t*)dev(b->pos); d.flag = (const uint16_t*)dev(b->flag); d.mapq = (const uint8_t*)dev(b->mapq);
            d.cell = (const int32_t*)dev(b->cell); d.umi = (const uint64_t*)dev(b->umi);
            d.cig_off = (const uint32_t*)dev(b->cig_off); d.cigar = (const uint32_t*)dev(b->cigar);
            if (im->mode == XCK_MODE_BAF) { d.seq_off = (const uint32_t*)dev(b->seq_off); d.seq = (const uint8_t*)dev(b->seq); }
            const size_t nr = (size_t)b->n_reads;
            im->st.algo_bytes_join += (int64_t)nr * 20 + (int64_t)(b->cig_off[nr] - b->cig_off[0]) * 4 + (im->mode == XCK_MODE_BAF ? (int64_t)((b->seq_off[nr] - b->seq_off[0]) / 2) : 0);
            im->queue.push_back(d); im->queued_reads += b->n_reads;
            if ((int)im->queue.size() >= MAX_FUSE) { rc = launch_queue(im, -1, si); if (rc) return rc; }
        }
        rc = launch_queue(im, -1, si); if (rc) return rc;                  // one fused launch per chunk and pipeline
    }
    return XCK_OK;
}

void engine_release_staging(xck_engine* e) {
    Stager* st = (Stager*)e->stager;
    if (!st) return;
    hipSetDevice(st->device);
    if (st->s_copy) hipStreamSynchronize(st->s_copy);
    for (auto& sl : st->slot) { if (sl.buf) hipFree(sl.buf); if (sl.t0) hipEventDestroy(sl.t0); if (sl.copied) hipEventDestroy(sl.copied); }
    if (st->s_copy) hipStreamDestroy(st->s_copy);
    delete st; e->stager = nullptr;
}

void fence_wait(void* f) { if (f) hipEventSynchronize((hipEvent_t)f); }
void fence_destroy(void* f) { if (f) hipEventDestroy((hipEvent_t)f); }

int engine_flush(xck_engine* e) {
    EngineImpl* im = (EngineImpl*)e->impl;
    if (!im) { e->err = "decode-only handle: no GPU engine behind it"; return XCK_E_STATE; }
    HIP_TRY(hipSetDevice(im->device));
    int rc = launch_queue(im, -1); if (rc) return rc;
    return complete_pending(im);
}

// ---- finish -------------------------------------------------------------------------------
struct Timer {
    EngineImpl* im; hipEvent_t a, b;
    int start() { HIP_TRY(hipEventRecord(a, im->s_comp)); return 0; }
    int stop(double* acc) { HIP_TRY(hipEventRecord(b, im->s_comp)); HIP_TRY(hipEventSynchronize(b)); float ms; HIP_TRY(hipEventElapsedTime(&ms, a, b)); *acc += ms; im->st.ms_device += ms; return 0; }
};

// One radix sort over key bits [0, top).  (rocPRIM's mid-size merge path is not stable, so the classic
// "sort the low range, then the high range" trick to skip the all-zero bits between the used UMI bits and
// the cell field is NOT safe with it - measured on gfx950, profiles/experiments/sorttest.hip.)
// Keys-only sort of 64-bit keys: sort kernel at 1024 threads x 8 keys and histogram kernel at 512 x 32 instead of rocPRIM
// 4.2's tuned default for gfx950 (512 x 12 for both): 7.90 ms instead of 9.42 ms for 380 M keys over 32 bits
// (profiles/experiments/sortcfg.hip, profiles/r01_g_sort_configs.log; 10-bit digits, which would need one pass fewer, are
// slower: 5.3 ms vs 4.5 ms at 200 M keys).  Everything else keeps the defaults.
template <class K>
using KeySortConfig = typename std::conditional<sizeof(K) == 8,
    rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                               rocprim::radix_sort_onesweep_config<rocprim::kernel_config<512, 32>, rocprim::kernel_config<1024, 8>, 8,
                                                                   rocprim::block_radix_rank_algorithm::match>>,
    rocprim::default_config>::type;
// pairs with 64-bit keys: 1024 x 6 with 8-byte values (4.29 ms vs 4.56 ms for 72 M pairs over 56 bits), 1024 x 8 with 1-byte values
// (1.31 ms vs 1.45 ms for 25 M pairs)
template <class K, class V>
using PairSortConfig = typename std::conditional<sizeof(K) == 8 && (sizeof(V) == 8 || sizeof(V) == 1),
    rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                               rocprim::radix_sort_onesweep_config<rocprim::kernel_config<512, 32>, rocprim::kernel_config<1024, sizeof(V) == 8 ? 6 : 8>, 8,
                                                                   rocprim::block_radix_rank_algorithm::match>>,
    rocprim::default_config>::type;

template <class K, class V>
static size_t sort_tmp_bytes(size_t n, int top) {
    size_t tb = 0; K* k = nullptr; V* v = nullptr;
    if constexpr (std::is_same<V, rocprim::empty_type>::value) (void)rocprim::radix_sort_keys<KeySortConfig<K>>(nullptr, tb, k, k, n, 0u, (unsigned)top, (hipStream_t)0);
    else (void)rocprim::radix_sort_pairs<PairSortConfig<K, V>>(nullptr, tb, k, k, v, v, n, 0u, (unsigned)top, (hipStream_t)0);
    return tb + 256;
}
template <class K, class V>
static int sort_run(EngineImpl* im, void* tmp, size_t tmp_bytes, K* kin, K* kout, V* vin, V* vout, size_t n, int top, int begin = 0) {
    hipError_t er;
    if constexpr (std::is_same<V, rocprim::empty_type>::value) er = rocprim::radix_sort_keys<KeySortConfig<K>>(tmp, tmp_bytes, kin, kout, n, (unsigned)begin, (unsigned)top, im->s_comp);
    else er = rocprim::radix_sort_pairs<PairSortConfig<K, V>>(tmp, tmp_bytes, kin, kout, vin, vout, n, 0u, (unsigned)top, im->s_comp);
    HIP_TRY(er);
    return 0;
}

static int copy_out(EngineImpl* im, int m, int32_t* d_o, size_t total);

// ordered compaction of the runs' AD / DP / OTH (hap: per-run sums, n staging entries) into the COO blocks of matrices m0..m0+2:
// counts and scans of all three first, ONE read-back of the totals, then the scatter and the copy-out
template <class K>
static int compact_coo(EngineImpl* im, Arena& ws, const HapSrc& hap, const K* keys, size_t n, KeyLayout<K> kl, int m0) {
    const int nm = 3;
    size_t nb = (n + CP_TILE - 1) / CP_TILE;
    uint32_t* d_blk = ws.get<uint32_t>(nb * nm); unsigned long long* d_off = ws.get<unsigned long long>(nb * nm);
    if (!d_blk || !d_off) { im->eng->err = "workspace exhausted (compaction)"; return XCK_E_NOMEM; }
    unsigned long long* d_tot = im->d_ctl + CTL_X0;                       // the k_expand words are free again at this point
    hipLaunchKernelGGL(k_hap_count, dim3(nb), dim3(CP_BLOCK), 0, im->s_comp, hap, d_blk);
    hipLaunchKernelGGL(k_cp_scan, dim3(nm), dim3(1024), 0, im->s_comp, d_blk, (long long)nb, d_off, d_tot);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, im->s_comp, (const unsigned long long*)d_tot, im->d_hctl + CTL_X0, nm);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(im->s_comp));
    CooOut3 out; memset(&out, 0, sizeof out);
    bool any = false;
    for (int y = 0; y < nm; y++) {
        const size_t total = im->h_ctl[CTL_X0 + y];
        im->res_nnz[m0 + y] = total; im->d_res[m0 + y] = nullptr; out.total[y] = total;
        if (!total) continue;
        int rc = res_reserve(im, m0 + y, total); if (rc) return rc;
        out.o[y] = ws.get<int32_t>(total * 3);
        if (!out.o[y]) { im->eng->err = "workspace exhausted (COO)"; return XCK_E_NOMEM; }
        any = true;
    }
    if (!any) return 0;
    hipLaunchKernelGGL((k_hap_scatter<K>), dim3(nb), dim3(CP_BLOCK), 0, im->s_comp, hap, keys, kl, (const unsigned long long*)d_off, out);
    HIP_TRY(hipGetLastError());
    // copy-out: large blocks go through copy_out() (copy stream); the small ones share one store kernel into mapped pinned memory
    CopySeg3 sg; memset(&sg, 0, sizeof sg); size_t mx = 0;
    for (int y = 0; y < nm; y++) {
        const size_t total = out.total[y];
        if (!total) continue;
        if (total * 3 * sizeof(int32_t) >= (size_t(8) << 20)) { int rc = copy_out(im, m0 + y, out.o[y], total); if (rc) return rc; continue; }
        im->d_res[m0 + y] = out.o[y];
        int32_t* alias = nullptr;
        HIP_TRY(hipHostGetDevicePointer((void**)&alias, im->h_res[m0 + y], 0));
        sg.src[y] = out.o[y]; sg.dst[y] = alias; sg.n[y] = total * 3; mx = std::max(mx, total * 3);
    }
    if (mx) {
        hipLaunchKernelGGL(k_copy_words3, dim3((unsigned)std::min<size_t>((mx + 255) / 256, 1024), nm), dim3(256), 0, im->s_comp, sg);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

// hand matrix m ([row|col|val] at d_o) to the host
static int copy_out(EngineImpl* im, int m, int32_t* d_o, size_t total) {
    im->d_res[m] = d_o;
    // copy-out on the copy stream, ordered behind the scatter: the compute stream (and other engines) keep
    // the CUs busy while the matrix crosses PCIe; xck_finish() waits for it, xck_finish_async() does not
    if (total * 3 * sizeof(int32_t) >= (size_t(8) << 20)) {
        HIP_TRY(hipEventRecord(im->ev_res, im->s_comp));
        HIP_TRY(hipStreamWaitEvent(im->s_copy, im->ev_res, 0));
        if (!im->copy_timed) { HIP_TRY(hipEventRecord(im->ev_c0, im->s_copy)); im->copy_timed = true; }
        HIP_TRY(hipMemcpyAsync(im->h_res[m], d_o, total * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, im->s_copy));
        HIP_TRY(hipEventRecord(im->ev_c1, im->s_copy));
    } else {                                                  // small matrix: CUs store it straight into mapped pinned memory
        int32_t* alias = nullptr;                             // (no DMA queue shared with another engine's bulk copy)
        HIP_TRY(hipHostGetDevicePointer((void**)&alias, im->h_res[m], 0));
        const unsigned g = (unsigned)std::min<size_t>((total * 3 + 255) / 256, 1024);
        hipLaunchKernelGGL(k_copy_words, dim3(g), dim3(256), 0, im->s_comp, (const int32_t*)d_o, alias, total * 3);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

// basefc: sorted keys -> COO (row, col, count of distinct keys) without a dense intermediate
constexpr int FOLD_GIANT = 1;              // fold_coo(): a (row, cell) run too long for the hash fold - redo on fully sorted keys
template <class K>
static int fold_coo(EngineImpl* im, Arena& ws, const K* keys, size_t n, KeyLayout<K> kl, int m, int sorted_from = 0) {   // sorted_from: lowest key bit the sort covered
    size_t nb = (n + FD_TILE - 1) / FD_TILE;
    uint32_t* d_blk = ws.get<uint32_t>(nb); unsigned long long* d_off = ws.get<unsigned long long>(nb);
    if (!d_blk || !d_off) { im->eng->err = "workspace exhausted (fold)"; return XCK_E_NOMEM; }
    hipLaunchKernelGGL((k_fold_heads<K>), dim3(nb), dim3(FD_BLOCK), 0, im->s_comp, keys, (long long)n, kl, d_blk);
    hipLaunchKernelGGL(k_cp_scan, dim3(1), dim3(1024), 0, im->s_comp, d_blk, (long long)nb, d_off, im->d_ctl + CTL_SCRATCH);
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, im->s_comp, (const unsigned long long*)(im->d_ctl + CTL_SCRATCH), im->d_hctl + CTL_SCRATCH, 1);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(im->s_comp));
    size_t total = im->h_ctl[CTL_SCRATCH];
    im->res_nnz[m] = total; im->d_res[m] = nullptr;
    if (!total) return 0;
    int rc = res_reserve(im, m, total); if (rc) return rc;
    int32_t* d_o = ws.get<int32_t>(total * 3);
    if (!d_o) { im->eng->err = "workspace exhausted (COO)"; return XCK_E_NOMEM; }
    HIP_TRY(hipMemsetAsync(d_o + 2 * total, 0, total * sizeof(int32_t), im->s_comp));       // k_fold_emit accumulates run pieces into val[]
    if constexpr (sizeof(K) == 8) {
        if (sorted_from > 0) {
            HIP_TRY(hipMemsetAsync(im->d_ctl + CTL_GIANT, 0, sizeof(unsigned long long), im->s_comp));
            KeyLayout<unsigned long long> kl8; kl8.ubits = kl.ubits; kl8.cbits = kl.cbits;
            hipLaunchKernelGGL(k_fold_emit_unsorted, dim3(nb), dim3(FU_BLOCK), FU_SLOTS * 8, im->s_comp, (const unsigned long long*)keys, (long long)n, kl8, d_off,
                               d_o, d_o + total, d_o + 2 * total, im->d_ctl + CTL_GIANT, sorted_from);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, im->s_comp, (const unsigned long long*)(im->d_ctl + CTL_GIANT), im->d_hctl + CTL_GIANT, 1);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(im->s_comp));
            if (im->eng->knobs.debug_timing) fprintf(stderr, "[xck] hash fold: n=%zu nnz=%zu ubits=%d cbits=%d giant=%llu\n", n, total, kl.ubits, kl.cbits, im->h_ctl[CTL_GIANT]);
            if (im->h_ctl[CTL_GIANT]) return FOLD_GIANT;
            return copy_out(im, m, d_o, total);
        }
    }
    hipLaunchKernelGGL((k_fold_emit<K>), dim3(nb), dim3(FD_BLOCK), 0, im->s_comp, keys, (long long)n, kl, d_off, d_o, d_o + total, d_o + 2 * total);
    HIP_TRY(hipGetLastError());
    return copy_out(im, m, d_o, total);
}

// shard slices -> one contiguous array, with the UMI field narrowed from `ubits` to `used` bits (the row and cell
// fields move down): every dead bit removed is one bit the radix sort does not have to pass over
__global__ void __launch_bounds__(256) k_pack_squeeze(const unsigned long long* __restrict__ src, unsigned long long cap, ShardSpan sp,
                                                        int ubits, int used, unsigned long long* __restrict__ dst) {
    const unsigned long long n = sp.start[NSHARD];
    const unsigned long long lowmask = (1ull << used) - 1;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) {
        int sh = 0;
#pragma unroll
        for (int q = 1; q < NSHARD; q++) sh += (i >= sp.start[q]) ? 1 : 0;
        const unsigned long long k = src[(unsigned long long)sh * cap + (i - sp.start[sh])];
        dst[i] = ((k >> ubits) << used) | (k & lowmask);
    }
}

// copy the used prefix of every shard slice into one contiguous array (sort input)
template <class K>
static int pack_shards(EngineImpl* im, K* dst_keys, uint64_t* dst_vals) {
    size_t off = 0;
    for (int sh = 0; sh < NSHARD; sh++) {
        const size_t c = im->cur[sh];
        if (!c) continue;
        HIP_TRY(hipMemcpyAsync(dst_keys + off, (K*)im->d_keys + (size_t)sh * im->hit_cap, c * sizeof(K), hipMemcpyDeviceToDevice, im->s_comp));
        if (dst_vals) HIP_TRY(hipMemcpyAsync(dst_vals + off, im->d_vals + (size_t)sh * im->hit_cap, c * sizeof(uint64_t), hipMemcpyDeviceToDevice, im->s_comp));
        off += c;
    }
    return 0;
}

#include "fold_partition.h"

template <class K>
static int finish_t(EngineImpl* im) {
    clear_stale_error("finish", im->eng->knobs.debug_timing);
    KeyLayout<K> kl; kl.ubits = im->ubits; kl.cbits = im->cbits;
    const size_t n = im->cursor;
    for (int m = 0; m < 4; m++) { im->res_nnz[m] = 0; im->d_res[m] = nullptr; }
    { int64_t acc = 0; for (int sh = 0; sh < NSHARD; sh++) acc += (int64_t)im->h_ctl[ctl_accepted(sh)];
      im->st.n_hits = acc; }                          // accepted pairs (before the LDS de-duplication)
    im->st.n_hits_unique = (int64_t)(n + im->ncursor);   // keys that reached HBM
#if XCK_STAMPS
    { static const char* nm[12] = {"record", "prologue", "stage_barrier", "sweep0", "sweep1", "sweep2", "sweep3", "flush_barrier", "flush_count", "flush_cursor", "flush_stores", "tail"};
      unsigned long long tot = 0; for (int q = 0; q < 12; q++) tot += im->stamp_sum[q];
      fprintf(stderr, "[stamps mode=%d] last join launch: %d tiles, %.3f ms; cycles of wave 0 per tile (share of the block's life):", im->mode, im->stamp_tiles, im->stamp_ms);
      for (int q = 0; q < 12; q++) fprintf(stderr, " %s=%.0f (%.1f%%)", nm[q], (double)im->stamp_sum[q] / std::max(1, im->stamp_tiles), 100.0 * im->stamp_sum[q] / std::max(1ull, tot));
      fprintf(stderr, " | total=%.0f\n", (double)tot / std::max(1, im->stamp_tiles)); }
#endif
    if (n == 0) return 0;
    Timer tm{im, im->ev0, im->ev1};
    int rc;
    const int top = im->ubits + im->cbits + im->rbits;
    const unsigned gs = (unsigned)((n + 255) / 256);
    const size_t nb = (n + CP_TILE - 1) / CP_TILE;
    K* keys = (K*)im->d_keys;
    if (im->mode == XCK_MODE_BASEFC) {
        // 64-bit keys: the partition fold (fold_partition.h - no sort); it hands back PF_FALLBACK for the inputs it cannot place
        // (one (row, cell) with more keys than a work item holds, ...), and the radix-sort fold below then takes over
        if constexpr (sizeof(K) == 8) {
            const bool want_sort = im->eng->knobs.fold_sort;
            if (!want_sort) {
                if ((rc = tm.start())) return rc;
                KeyLayout<unsigned long long> kl8; kl8.ubits = im->ubits; kl8.cbits = im->cbits;
                rc = fold_partition(im, kl8, n);
                if (rc == 0) {
                    im->fold_path = 1;
                    if ((rc = tm.stop(&im->st.ms_sort))) return rc;
                    HIP_TRY(hipStreamSynchronize(im->s_comp));
                    return 0;
                }
                if (rc != PF_FALLBACK) return rc;
                im->fold_fallbacks++;
                if (im->eng->knobs.debug_timing) fprintf(stderr, "[xck] partition fold: handing over to the radix-sort fold\n");
            }
        }
        im->fold_path = 2;
        const size_t tmpb = sort_tmp_bytes<K, rocprim::empty_type>(n, top);
        if ((rc = arena_begin(im, im->ws1, n * sizeof(K) + tmpb + nb * 12 + n * 12 + (1 << 20)))) return rc;
        K* alt = im->ws1.get<K>(n); void* tmp = im->ws1.get<char>(tmpb);
        if ((rc = tm.start())) return rc;
        int used = im->ubits;                                                       // UMI bits actually in use
        if (sizeof(K) == 8) { unsigned long long uor = 0; for (int sh = 0; sh < NSHARD; sh++) uor |= im->h_ctl[ctl_umi_or(sh)]; used = uor ? 64 - __builtin_clzll(uor) : 1; if (used > im->ubits) used = im->ubits; }
        if (used < im->ubits) {
            ShardSpan sp; sp.start[0] = 0; for (int sh = 0; sh < NSHARD; sh++) sp.start[sh + 1] = sp.start[sh] + im->cur[sh];
            hipLaunchKernelGGL(k_pack_squeeze, dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, im->s_comp,
                               (const unsigned long long*)im->d_keys, (unsigned long long)im->hit_cap, sp, im->ubits, used, (unsigned long long*)alt);
            HIP_TRY(hipGetLastError());
            kl.ubits = used;
        } else
        if ((rc = pack_shards(im, alt, (uint64_t*)nullptr))) return rc;             // shard slices -> contiguous
        const int top_fc = kl.ubits + im->cbits + im->rbits;
        // 64-bit keys: the radix sort only orders (row, cell) - 4 passes instead of 7 - and the fold tells the UMIs of a run
        // apart with an LDS hash set; a run too long for that (FOLD_GIANT) is redone on fully sorted keys
        const bool full_sort = im->eng->knobs.full_sort;
        // (rocPRIM 4.2 returns garbage for begin_bit > 0 with end_bit = 64 - profiles/experiments/sortpart.hip - so keys that could not be
        // squeezed below 63 bits take the classic path)
        const bool partial = sizeof(K) == 8 && !full_sort && top_fc <= 62;
        // whole 8-bit digits from the top: the (row, cell) bits plus whatever UMI bits the last digit reaches for free.  A run
        // that is still too long for the fold's lead-in window (FOLD_GIANT: one gene holding a large share of a cell's reads)
        // is split further - one more digit per attempt, remembered for the next finish() of this handle.
        auto begin_for = [&](int extra) { const int passes = (top_fc - kl.ubits + 7) / 8 + extra; return std::max(0, top_fc - 8 * passes); };
        int begin = partial ? begin_for(im->fold_extra_digits) : 0;
        K* src = alt; K* dst = keys;
        if ((rc = sort_run<K, rocprim::empty_type>(im, tmp, tmpb, src, dst, nullptr, nullptr, n, top_fc, begin))) return rc;
        const size_t ws_mark = im->ws1.off;                              // a discarded attempt gives its workspace back
        rc = fold_coo<K>(im, im->ws1, dst, n, kl, 0, begin);
        while (rc == FOLD_GIANT) {
            im->ws1.off = ws_mark;
            im->fold_extra_digits++;
            begin = begin_for(im->fold_extra_digits);
            std::swap(src, dst);                                          // any order of the same keys is a valid sort input
            if ((rc = sort_run<K, rocprim::empty_type>(im, tmp, tmpb, src, dst, nullptr, nullptr, n, top_fc, begin))) return rc;
            rc = fold_coo<K>(im, im->ws1, dst, n, kl, 0, begin);      // begin == 0: fully sorted, classic fold, cannot be giant
        }
        if (rc) return rc;
        if ((rc = tm.stop(&im->st.ms_sort))) return rc;
    } else {
        const size_t tmpb = sort_tmp_bytes<K, uint64_t>(n, top);
        if ((rc = arena_begin(im, im->ws1, n * sizeof(K) + n * 8 + n + tmpb + n * 8 + (size_t)std::max(im->n_snps_sorted, 1) * 17 + std::max<size_t>(n * 8, 8192) + n / 8 + (1 << 16)))) return rc;
        K* alt = im->ws1.get<K>(n); uint64_t* valt = im->ws1.get<uint64_t>(n); void* tmp = im->ws1.get<char>(tmpb); uint8_t* al = im->ws1.get<uint8_t>(n);
        if ((rc = tm.start())) return rc;
        // the hits sorted by (key, value): by row partition + one LDS sort per item (fold_partition.h); a SNP deeper than an item, or
        // 128-bit keys, take the radix sort
        bool sorted = false;
        PartIndex pidx{nullptr, nullptr};                                    // the partition's cells, for k_claim's look-ups (stays null after the radix sort)
        if constexpr (sizeof(K) == 8) {
            // (default since the items are sorted by an LDS radix sort: 1.7 ms at configs[2] against 2.3 ms for pack + rocPRIM's eight passes;
            // XCK_PILEUP_SORT=radix forces the library sort)
            const bool want_part = !im->eng->knobs.pileup_radix;
            if (want_part) {
                KeyLayout<unsigned long long> kl8; kl8.ubits = im->ubits; kl8.cbits = im->cbits;
                rc = pileup_partition_sort(im, im->ws2, true, kl8, (const unsigned long long*)im->d_keys, (const uint64_t*)im->d_vals, im->hit_cap, im->cur,
                                           (uint32_t)std::max(im->n_snps_sorted, 1), n, (unsigned long long*)alt, valt, nullptr, &pidx);
                if (rc == 0) { sorted = true; im->pileup_sort_path = 1; }
                else if (rc != PF_FALLBACK) return rc;
            }
        }
        if (!sorted) {
        im->pileup_sort_path = 2;
        { ShardSpan sp; sp.start[0] = 0; for (int sh = 0; sh < NSHARD; sh++) sp.start[sh + 1] = sp.start[sh] + im->cur[sh];
          hipLaunchKernelGGL((k_pack_pairs<K>), dim3((unsigned)std::min<size_t>((n + 255) / 256, 8192)), dim3(256), 0, im->s_comp,
                             (const K*)im->d_keys, (const uint64_t*)im->d_vals, (unsigned long long)im->hit_cap, sp, alt, valt);
          HIP_TRY(hipGetLastError()); }
        if ((rc = sort_run<K, uint64_t>(im, tmp, tmpb, alt, keys, valt, im->d_vals, n, top))) return rc;
        std::swap(alt, keys); { uint64_t* t_ = valt; valt = im->d_vals; (void)t_; }   // sorted data now lives in d_keys / d_vals
        }
        HIP_TRY(hipMemsetAsync(im->d_tally, 0, std::max<size_t>((size_t)im->n_snps_sorted * 5, 1) * sizeof(uint32_t), im->s_comp));
        unsigned long long* long_runs = im->ws1.get<unsigned long long>(n / (size_t)RUN_WALK + 2);      // [0] = count, then the heads of the runs longer than RUN_WALK
        if (!long_runs) { im->eng->err = "workspace exhausted (pileup fold)"; return XCK_E_NOMEM; }
        HIP_TRY(hipMemsetAsync(long_runs, 0, sizeof(unsigned long long), im->s_comp));
        if (sizeof(K) == 8 && split_mode(im)) {
            const size_t ns = std::max<size_t>((size_t)im->n_snps_sorted, 1);
            uint64_t* ordv = im->ws1.get<uint64_t>(n); unsigned long long* row_lo = im->ws1.get<unsigned long long>(2 * ns); unsigned long long* row_hi = row_lo + ns;
            if (n >> 32) { im->eng->err = "pileup fold: more than 2^32 hits with a base in one finish"; return XCK_E_NOMEM; }
            const uint32_t n_blk = (uint32_t)((ns + 31) >> 5);
            unsigned long long* bloom = im->ws1.get<unsigned long long>(n);   // one word per key, laid out along the sorted stream (bloom_slot)
            unsigned long long* blk_lo = im->ws1.get<unsigned long long>((size_t)n_blk + 1);
            if (!ordv || !row_lo || !bloom || !blk_lo) { im->eng->err = "workspace exhausted (split pileup)"; return XCK_E_NOMEM; }
            HIP_TRY(hipMemsetAsync(row_lo, 0, 2 * ns * sizeof(unsigned long long), im->s_comp));
            HIP_TRY(hipMemsetAsync(bloom, 0, n * sizeof(unsigned long long), im->s_comp));
            hipLaunchKernelGGL((k_blk_bounds<K>), dim3((n_blk + 256) / 256), dim3(256), 0, im->s_comp, (const K*)alt, (long long)n, kl, n_blk, blk_lo);
            hipLaunchKernelGGL((k_first_base<K>), dim3(gs), dim3(256), 0, im->s_comp, alt, valt, (long long)n, kl, al, ordv, row_lo, row_hi,
                               bloom, (const unsigned long long*)blk_lo, long_runs);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL((k_first_long<K, true>), dim3(1024), dim3(256), 0, im->s_comp, alt, valt, (long long)n, kl, (const unsigned long long*)long_runs, al, ordv, im->d_tally);
            HIP_TRY(hipGetLastError());
            if (im->ncursor) {
                ShardSpan nsp; nsp.start[0] = 0; unsigned long long mx = 0;
                for (int sh = 0; sh < NSHARD; sh++) { nsp.start[sh + 1] = nsp.start[sh] + im->ncur[sh]; mx = std::max(mx, im->ncur[sh]); }
                const unsigned long long n_units = ((mx + 256 * CL_U - 1) / (256 * CL_U)) * NSHARD;
                // (one block per unit: blocks start in index order, so the resident ones hold consecutive units = ONE window of the table; a grid-stride
                // loop over 16 k blocks mixed up to 11 windows and every probe went to HBM: 5.5 GB read for 0.74 GB of records)
                hipLaunchKernelGGL((k_claim<K>), dim3((unsigned)std::min<unsigned long long>(n_units, 1ull << 30)), dim3(256), 0, im->s_comp,
                                   (const K*)im->d_nkeys, (const uint64_t*)im->d_nvals, (unsigned long long)im->hit_cap, nsp, n_units,
                                   (const K*)alt, kl, (const unsigned long long*)row_lo, (const unsigned long long*)row_hi, (const uint64_t*)ordv, al,
                                   (const unsigned long long*)bloom, (const unsigned long long*)blk_lo, (uint32_t)ns, pidx.rowtab, pidx.end);
                HIP_TRY(hipGetLastError());
            }
            HIP_TRY(hipMemsetAsync(long_runs, 0, sizeof(unsigned long long), im->s_comp));   // (k_first_long is done with the list: now the SNPs deeper than TALLY_LONG)
            hipLaunchKernelGGL(k_tally_rows, dim3((unsigned)((ns * 8 + 255) / 256)), dim3(256), 0, im->s_comp, (const uint8_t*)al, (const unsigned long long*)row_lo, (const unsigned long long*)row_hi,
                               (uint32_t)ns, im->d_tally, long_runs);
            hipLaunchKernelGGL(k_tally_long, dim3(1024), dim3(256), 0, im->s_comp, (const uint8_t*)al, (const unsigned long long*)row_lo, (const unsigned long long*)row_hi,
                               (const unsigned long long*)long_runs, im->d_tally);
            HIP_TRY(hipGetLastError());
        } else {
        hipLaunchKernelGGL((k_first_read<K>), dim3(gs), dim3(256), 0, im->s_comp, alt, valt, (long long)n, kl, al, im->d_tally, long_runs);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL((k_first_long<K, false>), dim3(1024), dim3(256), 0, im->s_comp, alt, valt, (long long)n, kl, (const unsigned long long*)long_runs, al, (uint64_t*)nullptr, im->d_tally);
        HIP_TRY(hipGetLastError());
        }
        // region-level values: 64-bit words beside 64-bit keys (the partition sort carries 64-bit values), bytes beside 128-bit keys
        typedef typename std::conditional<sizeof(K) == 8, uint64_t, uint8_t>::type V2;
        XBases xb; memset(&xb, 0, sizeof xb);
        HIP_TRY(hipMemsetAsync(im->d_ctl + CTL_X0, 0, XSHARD * CTL_STRIDE * sizeof(unsigned long long), im->s_comp));
        hipLaunchKernelGGL((k_expand<K, false, V2>), dim3(gs), dim3(JOIN_BLOCK), 0, im->s_comp, alt, al, (long long)n, kl, im->d_tally, im->d_snp_info,
                           im->sf, im->d_csr_off, im->d_csr_reg, (K*)nullptr, (V2*)nullptr, im->d_ctl, xb, -1);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(256), 0, im->s_comp, (const unsigned long long*)(im->d_ctl + CTL_X0), im->d_hctl + CTL_X0, XSHARD * CTL_STRIDE);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(im->s_comp));
        size_t n2 = 0; unsigned long long tot2[XSHARD], cap2 = 0;
        unsigned long long uor2 = 0;
        for (int sh = 0; sh < XSHARD; sh++) { tot2[sh] = im->h_ctl[CTL_X0 + sh * CTL_STRIDE]; n2 += tot2[sh]; cap2 = std::max(cap2, tot2[sh]); }
        for (int sh = 0; sh < NSHARD; sh++) uor2 |= im->h_ctl[ctl_umi_or(sh)];
        const int used2 = uor2 ? 64 - __builtin_clzll(uor2) : 0;            // UMI-field bits in use by the reads the join accepted (a superset of the region-level keys')
        cap2 = (cap2 + 63) & ~63ull;
        if (n2) {
            // 64-bit keys: k_expand writes its 16 slices at a fixed stride and the partition sort (fold_partition.h) orders them; when it hands
            // back PF_FALLBACK (a (region, cell group) deeper than an item), or with 128-bit keys, k_expand writes the slices back to back and
            // the library radix sort orders them.  XCK_PILEUP_SORT=radix forces the latter.
            const bool try_part = sizeof(K) == 8 && !im->eng->knobs.pileup_radix;
            const size_t n2s = try_part ? std::max<size_t>((size_t)XSHARD * cap2, n2) : n2;      // entries of the unsorted buffers
            const size_t tmpb2 = sort_tmp_bytes<K, V2>(n2, top);
            const size_t nb2 = (n2 + CP_TILE - 1) / CP_TILE;
            const size_t part_bytes = try_part ? partition_sort_scratch(im, n2, (size_t)std::max(im->n_regions, 1)) : 0;
            if ((rc = arena_begin(im, im->ws2, (n2s + n2 + 8) * (sizeof(K) + sizeof(V2)) + n2 + 4 * (n2 + 8) * 4 + std::max(tmpb2, part_bytes) + 3 * (nb2 * 12 + n2 * 12) + ((n2 + FD_TILE - 1) / FD_TILE) * 12 + (n2 / RUN_WALK + 2) * 8 + (1 << 16)))) return rc;
            K* k2 = im->ws2.get<K>(n2s + 8); K* k2b = im->ws2.get<K>(n2); V2* v2 = im->ws2.get<V2>(n2s); V2* v2b = im->ws2.get<V2>(n2);
            uint8_t* cls = im->ws2.get<uint8_t>(n2);
            const long long stride2 = (long long)((n2 + 7) & ~size_t(7));  // (k_hap_count / k_hap_scatter read eight runs with two 16-byte loads)
            uint32_t* acc = im->ws2.get<uint32_t>(4 * (size_t)stride2);      // per run: REF-hap, ALT-hap, either, other-only keys
            if (!k2 || !k2b || !v2 || !v2b || !cls || !acc) { im->eng->err = "workspace exhausted (region-level hits)"; return XCK_E_NOMEM; }
            HIP_TRY(hipMemsetAsync(acc, 0, 2 * (size_t)stride2 * sizeof(uint32_t), im->s_comp));   // (the half that k_hap_items' packed sums use; the rest below, if k_hap_sum runs)
            K* run_key = k2;                                                // (the unsorted keys are dead once they are partitioned / sorted)
            const size_t ws2_mark = im->ws2.off;
            bool sorted2 = false, summed2 = false;
            if constexpr (sizeof(K) == 8) {
                if (try_part) {
                    for (int sh = 0; sh < XSHARD; sh++) xb.base[sh] = (unsigned long long)sh * cap2;
                    // (XCK_PILEUP_HAP=sorted: sort the items completely and run k_hap_class / k_hap_sum on them, as after the radix sort;
                    //  XCK_PILEUP_HAP=values: keep the haplotype class in a value word beside the key, as when the UMI field has no two free bits)
                    const bool hap_items = im->eng->knobs.pileup_hap != 1;
                    const int pack_shift = hap_items && im->eng->knobs.pileup_hap != 2 && used2 + 2 <= im->ubits ? used2 : -1;
                    hipLaunchKernelGGL((k_expand<K, true, V2>), dim3(gs), dim3(JOIN_BLOCK), 0, im->s_comp, alt, al, (long long)n, kl, im->d_tally, im->d_snp_info,
                                       im->sf, im->d_csr_off, im->d_csr_reg, k2, v2, im->d_ctl, xb, pack_shift);
                    HIP_TRY(hipGetLastError());
                    KeyLayout<unsigned long long> kl8; kl8.ubits = im->ubits; kl8.cbits = im->cbits;
                    const HapItemsOut ho{(unsigned long long*)acc, (unsigned long long*)run_key, pack_shift};   // (the packed sums use the first half of acc)
                    rc = pileup_partition_sort(im, im->ws2, false, kl8, (const unsigned long long*)k2, pack_shift >= 0 ? (const uint64_t*)nullptr : (const uint64_t*)v2, (size_t)cap2, tot2,
                                               (uint32_t)std::max(im->n_regions, 1), n2, (unsigned long long*)k2b, pack_shift >= 0 ? (uint64_t*)nullptr : (uint64_t*)v2b, hap_items ? &ho : nullptr);
                    im->ws2.off = ws2_mark;                                    // (its scratch is free again; the kernels that used it are ordered before the next ones)
                    if (rc == 0) { sorted2 = true; summed2 = hap_items; im->pileup_sort2_path = hap_items ? 1 : 3; }
                    else if (rc != PF_FALLBACK) return rc;
                    else {                                                 // the cursors of the emit pass start again
                        for (int sh = 0; sh < XSHARD; sh++) HIP_TRY(hipMemsetAsync(im->d_ctl + CTL_X0 + sh * CTL_STRIDE + 1, 0, sizeof(unsigned long long), im->s_comp));
                    }
                }
            }
            if (!sorted2) {
                im->pileup_sort2_path = 2;
                void* tmp2 = im->ws2.get<char>(tmpb2);
                { unsigned long long at = 0; for (int sh = 0; sh < XSHARD; sh++) { xb.base[sh] = at; at += tot2[sh]; } }
                hipLaunchKernelGGL((k_expand<K, true, V2>), dim3(gs), dim3(JOIN_BLOCK), 0, im->s_comp, alt, al, (long long)n, kl, im->d_tally, im->d_snp_info,
                                   im->sf, im->d_csr_off, im->d_csr_reg, k2, v2, im->d_ctl, xb, -1);
                HIP_TRY(hipGetLastError());
                if ((rc = sort_run<K, V2>(im, tmp2, tmpb2, k2, k2b, v2, v2b, n2, top))) return rc;
            }
            HapSrc hs{(const uint32_t*)acc, stride2, (const unsigned long long*)nullptr, (long long)n2, im->no_dup_hap, (const unsigned long long*)acc};   // k_hap_items: runs staged (packed) at their items' offsets
            if (!summed2) {                                                 // sorted keys: classes per (row, cell, UMI) run, sums per (row, cell) run
                HIP_TRY(hipMemsetAsync(acc + 2 * (size_t)stride2, 0, 2 * (size_t)stride2 * sizeof(uint32_t), im->s_comp));
                const unsigned gs2 = (unsigned)((n2 + 255) / 256);
                const size_t nt2 = (n2 + FD_TILE - 1) / FD_TILE;
                uint32_t* d_blk2 = im->ws2.get<uint32_t>(nt2); unsigned long long* d_off2 = im->ws2.get<unsigned long long>(nt2);
                unsigned long long* long2 = im->ws2.get<unsigned long long>(n2 / (size_t)RUN_WALK + 2);   // [0] = count, then the heads of the (row, cell, UMI) runs longer than RUN_WALK
                if (!d_blk2 || !d_off2 || !long2) { im->eng->err = "workspace exhausted (haplotype classes)"; return XCK_E_NOMEM; }
                HIP_TRY(hipMemsetAsync(long2, 0, sizeof(unsigned long long), im->s_comp));
                hipLaunchKernelGGL((k_hap_class<K, V2>), dim3(gs2), dim3(256), 0, im->s_comp, (const K*)k2b, (const V2*)v2b, (long long)n2, cls, long2);
                hipLaunchKernelGGL((k_hap_class_long<K, V2>), dim3(256), dim3(256), 0, im->s_comp, (const K*)k2b, (const V2*)v2b, (long long)n2, (const unsigned long long*)long2, cls);
                hipLaunchKernelGGL((k_fold_heads<K>), dim3((unsigned)nt2), dim3(FD_BLOCK), 0, im->s_comp, (const K*)k2b, (long long)n2, kl, d_blk2);
                hipLaunchKernelGGL(k_cp_scan, dim3(1), dim3(1024), 0, im->s_comp, d_blk2, (long long)nt2, d_off2, im->d_ctl + CTL_SCRATCH);
                hipLaunchKernelGGL((k_hap_sum<K>), dim3((unsigned)nt2), dim3(FD_BLOCK), 0, im->s_comp, (const K*)k2b, (const uint8_t*)cls, (long long)n2, kl,
                                   (const unsigned long long*)d_off2, run_key, acc, stride2);
                HIP_TRY(hipGetLastError());
                hs.n_runs = (const unsigned long long*)(im->d_ctl + CTL_SCRATCH); hs.packed = nullptr;
            }
            if ((rc = compact_coo<K>(im, im->ws2, hs, run_key, n2, kl, 1))) return rc;   // AD, DP, OTH together, from the per-run sums
        }
        if ((rc = tm.stop(&im->st.ms_sort))) return rc;
    }
    HIP_TRY(hipStreamSynchronize(im->s_comp));
    return 0;
}

// fold everything on the GPU and ENQUEUE the copy-out of the matrices; does not wait for the copy
int engine_finish_async(xck_engine* e) {
    EngineImpl* im = (EngineImpl*)e->impl;
    if (!im) { e->err = "decode-only handle: no GPU engine behind it"; return XCK_E_STATE; }
    HIP_TRY(hipSetDevice(im->device));
    int rc = launch_queue(im, -1); if (rc) return rc;
    rc = complete_pending(im); if (rc) return rc;
    if (!im->finished) {
        // a finish that failed half-way may have overwritten the accumulated keys (the folds reuse the shard slices as scratch): it
        // cannot be tried again on them
        if (im->fold_failed) { e->err = "an earlier xck_finish failed inside the fold: the accumulated hits are gone (call xck_reset)"; return XCK_E_STATE; }
        im->copy_timed = false;
        rc = im->key_bits == 64 ? finish_t<uint64_t>(im) : finish_t<u128>(im);
        if (rc) { im->fold_failed = true; hipStreamSynchronize(im->s_comp); hipStreamSynchronize(im->s_copy); return rc; }   // (nothing of the failed fold is still running when the arenas are reused)
        im->finished = true; im->copy_pending = true;
    }
    return 0;
}

int engine_finish(xck_engine* e, xck_result* out) {
    EngineImpl* im = (EngineImpl*)e->impl;
    if (!im) { e->err = "decode-only handle: no GPU engine behind it"; return XCK_E_STATE; }
    int rc = engine_finish_async(e); if (rc) return rc;
    if (im->copy_pending) {
        const bool dbg = im->eng->knobs.debug_timing;
        const auto t0_ = std::chrono::steady_clock::now();
        const hipError_t q_ = dbg ? hipStreamQuery(im->s_copy) : hipSuccess;
        HIP_TRY(hipStreamSynchronize(im->s_copy));
        if (dbg) fprintf(stderr, "[xck] finish: copy stream %s at entry, waited %.3f ms\n", q_ == hipSuccess ? "idle" : "busy",
                         std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0_).count());
        if (im->copy_timed) { float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, im->ev_c0, im->ev_c1)); im->st.ms_d2h += ms; }
        im->copy_pending = false;
    }
    memset(out, 0, sizeof *out);
    xck_coo* dst[4] = { &out->count, &out->ad, &out->dp, &out->oth };
    for (int m = 0; m < 4; m++) {
        const size_t z = im->res_nnz[m];
        dst[m]->nnz = (int64_t)z;
        dst[m]->row = im->h_res[m]; dst[m]->col = im->h_res[m] ? im->h_res[m] + z : nullptr; dst[m]->val = im->h_res[m] ? im->h_res[m] + 2 * z : nullptr;
    }
    return 0;
}

// device-resident copy of the last finish() result (for device-to-device exchanges such as the RCCL gather)
int engine_result_device(xck_engine* e, xck_result* out) {
    EngineImpl* im = (EngineImpl*)e->impl;
    if (!im) { e->err = "decode-only handle: no GPU engine behind it"; return XCK_E_STATE; }
    if (!im->finished) { e->err = "xck_get_result_device before xck_finish"; return XCK_E_STATE; }
    memset(out, 0, sizeof *out);
    xck_coo* dst[4] = { &out->count, &out->ad, &out->dp, &out->oth };
    for (int m = 0; m < 4; m++) {
        const size_t z = im->res_nnz[m];
        dst[m]->nnz = (int64_t)z;
        if (z && im->d_res[m]) { dst[m]->row = im->d_res[m]; dst[m]->col = im->d_res[m] + z; dst[m]->val = im->d_res[m] + 2 * z; }
    }
    return 0;
}

int engine_reset(xck_engine* e) {
    EngineImpl* im = (EngineImpl*)e->impl;
    if (!im) { e->err = "decode-only handle: no GPU engine behind it"; return XCK_E_STATE; }
    HIP_TRY(hipSetDevice(im->device));
    int rc = launch_queue(im, -1); if (rc) return rc;
    rc = complete_pending(im); if (rc) return rc;
    if (im->copy_pending) { HIP_TRY(hipStreamSynchronize(im->s_copy)); im->copy_pending = false; }
    HIP_TRY(hipMemsetAsync(im->d_ctl, 0, CTL_WORDS * sizeof(unsigned long long), im->s_comp));
    HIP_TRY(hipStreamSynchronize(im->s_comp));
    im->cursor = 0; im->ncursor = 0; im->finished = false; im->fold_failed = false;
    for (int i = 0; i < CTL_WORDS; i++) im->h_ctl[i] = 0;
    for (int sh = 0; sh < NSHARD; sh++) { im->cur[sh] = 0; im->ncur[sh] = 0; }
    int kb = im->key_bits, ub = im->ubits;
    memset(&im->st, 0, sizeof im->st);
    im->st.key_bits = kb; im->st.umi_bits = ub; im->n_join_launches = 0;
    return 0;
}

int engine_device(const xck_engine* e) { return e->n_impl > 0 && e->impls[0] ? ((const EngineImpl*)e->impls[0])->device : -1; }

int engine_stats(const xck_engine* e, xck_stats* out) {
    const EngineImpl* im = (const EngineImpl*)e->impl;
    if (!im) return XCK_E_STATE;
    *out = im->st; out->key_bits = im->key_bits; out->umi_bits = im->ubits;
    out->n_join_launches = im->n_join_launches;
    out->fold_path = im->fold_path; out->fold_fallbacks = im->fold_fallbacks; out->pileup_sort_path = im->pileup_sort_path; out->fold_refinements = im->fold_refinements;
    out->pileup_sort2_path = im->pileup_sort2_path; out->gpu_inflate_chunks = (int32_t)std::min<int64_t>(e->gpu_inflate_chunks.load(), INT32_MAX);
    return 0;
}

int engine_numa_node(const xck_engine* e) {
    const EngineImpl* im = e && e->n_impl > 0 ? (const EngineImpl*)e->impls[0] : nullptr;
    if (!im) return -1;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, im->device) != hipSuccess) { (void)hipGetLastError(); return -1; }
    for (char* p = bus; *p; p++) if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');
    char path[160]; snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE* f = fopen(path, "r"); if (!f) return -1;
    int node = -1; if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f);
    return node;
}
int engine_umi_bits(const xck_engine* e) { const EngineImpl* im = (const EngineImpl*)e->impl; return im ? im->ubits : 0; }

int engine_create(const xck_config* cfg, xck_engine* e) {
    EngineImpl* im = new EngineImpl();
    im->eng = e; e->impl = im;
    im->mode = cfg->mode; im->device = cfg->device;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { e->err = "no HIP device available (the engine has no CPU fallback)"; return XCK_E_DEVICE; }
    if (cfg->device < 0 || cfg->device >= ndev) { e->err = "device ordinal out of range"; return XCK_E_ARG; }
    HIP_TRY(hipSetDevice(im->device));
    // filters (double-typed thresholds become exact integer thresholds: x < v  <=>  x < ceil(v) for integer x)
    im->rf.min_mapq = (int32_t)std::min(256.0, std::max(0.0, std::ceil(cfg->min_mapq)));
    im->rf.min_len = cfg->min_len;
    im->rf.incl_flag = cfg->incl_flag; im->rf.excl_flag = cfg->excl_flag; im->rf.no_orphan = cfg->no_orphan;
    im->rf.frac_mode = (cfg->min_include > 0.0 && cfg->min_include < 1.0) ? 1 : 0;
    im->rf.min_inc_frac = cfg->min_include;
    im->rf.min_inc_len = cfg->min_include <= 0.0 ? 0 : (int32_t)std::min(2147483647.0, std::ceil(cfg->min_include));
    im->sf.min_count = cfg->min_count <= 0.0 ? 0 : (int32_t)std::min(2147483647.0, std::ceil(cfg->min_count));
    im->sf.min_maf = cfg->min_maf;
    im->no_dup_hap = cfg->no_dup_hap;
    im->n_cells = cfg->n_cells; im->n_regions = cfg->n_regions;
    { KeyBits kb = key_layout(cfg); im->key_bits = kb.key_bits; im->ubits = kb.ubits; im->cbits = kb.cbits; im->rbits = kb.rbits; }
    im->st.key_bits = im->key_bits; im->st.umi_bits = im->ubits;
    im->max_batch_reads = cfg->max_batch_reads > 0 ? cfg->max_batch_reads : (int64_t)1 << 21;
    int rc = build_tables(im, cfg); if (rc) return rc;
    // the hash fold needs 64 KB of dynamic LDS: raise the limit on THIS engine's device (a per-process flag would leave every
    // device but the first at the 64 KB default and race between engines created from different threads)
    HIP_TRY(hipFuncSetAttribute((const void*)k_fold_emit_unsorted, hipFuncAttributeMaxDynamicSharedMemorySize, FU_SLOTS * 8));
    HIP_TRY(hipFuncSetAttribute((const void*)k_pf_bucket, hipFuncAttributeMaxDynamicSharedMemorySize, pf_bucket_lds(PF_SB_MAX)));
    HIP_TRY(hipStreamCreateWithFlags(&im->s_copy, hipStreamNonBlocking));
    { int lo = 0, hi = 0;                                   // numerically lowest value = highest priority
      HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
      HIP_TRY(hipStreamCreateWithPriority(&im->s_comp, hipStreamNonBlocking, (cfg->flags & XCK_F_LOW_PRIORITY) ? lo : hi)); }
    HIP_TRY(hipEventCreate(&im->ev0)); HIP_TRY(hipEventCreate(&im->ev1));
    HIP_TRY(hipEventCreate(&im->ev_res)); HIP_TRY(hipEventCreate(&im->ev_c0)); HIP_TRY(hipEventCreate(&im->ev_c1));
    HIP_TRY(hipMalloc((void**)&im->d_ctl, CTL_WORDS * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(im->d_ctl, 0, CTL_WORDS * sizeof(unsigned long long)));
    HIP_TRY(hipHostMalloc((void**)&im->h_ctl, CTL_WORDS * sizeof(unsigned long long), hipHostMallocMapped));
    memset(im->h_ctl, 0, CTL_WORDS * sizeof(unsigned long long));
    HIP_TRY(hipHostGetDevicePointer((void**)&im->d_hctl, im->h_ctl, 0));
    rc = ensure_hits(im, (size_t)e->knobs.hit_cap0); if (rc) return rc;   // (XCK_HIT_CAP0: test knob)
    return 0;
}

void engine_destroy(xck_engine* e) {
    EngineImpl* im = (EngineImpl*)e->impl;
    if (!im) return;
    hipSetDevice(im->device);
    if (im->s_comp) hipStreamSynchronize(im->s_comp);
    void* ptrs[] = { im->d_reg_s0, im->d_reg_e0, im->d_reg_row, im->d_reg_pmax, im->d_snp_p0, im->d_snp_win,
                     im->d_csr_off, im->d_csr_reg, im->d_snp_info, im->d_tally, im->d_keys, im->d_vals, im->d_nkeys, im->d_nvals, im->d_ctl, im->d_meta,
                     im->ws1.base, im->ws2.base };
    for (void* p : ptrs) if (p) hipFree(p);
    for (auto& s : im->slot) { void* q[] = { s.pos, s.flag, s.mapq, s.cell, s.umi, s.cig_off, s.cigar, s.seq_off, s.seq }; for (void* p : q) if (p) hipFree(p); }
    for (int m = 0; m < 4; m++) if (im->h_res[m]) hipHostFree(im->h_res[m]);
    if (im->h_ctl) hipHostFree(im->h_ctl);
    if (im->ev0) hipEventDestroy(im->ev0);
    if (im->ev1) hipEventDestroy(im->ev1);
    if (im->ev_res) hipEventDestroy(im->ev_res);
    if (im->ev_c0) hipEventDestroy(im->ev_c0);
    if (im->ev_c1) hipEventDestroy(im->ev_c1);
    if (im->ev_f1) hipEventDestroy(im->ev_f1);
    if (im->ev_f2) hipEventDestroy(im->ev_f2);
    if (im->s_copy) hipStreamDestroy(im->s_copy);
    if (im->s_comp) hipStreamDestroy(im->s_comp);
    delete im; e->impl = nullptr;
}

void* pinned_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess) return p;
    return nullptr;
}
void pinned_free(void* p) { if (p) hipHostFree(p); }

}  // namespace xck
