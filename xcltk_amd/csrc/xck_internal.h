// xck_internal.h - shared declarations between the HIP engine, the BAM decoder and the C-ABI.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include <unordered_map>
#include <mutex>
#include <atomic>
#include "../../include/xck.h"

#define XCK_VERSION_STR "0.1.0 (gfx950)"

namespace xck {

// BAM flag bits used by check_read (reference xcltk/utils/sam.py:120-121)
constexpr uint32_t BAM_FPAIRED = 1, BAM_FPROPER_PAIR = 2, BAM_FUNMAP = 4;

// ordinal layout: (bam_index << 40) | record number ; value word of a pileup hit:
// (ordinal << 5) | (allele nibble + 1), 0 in the low 5 bits = "read has no base at the SNP"
constexpr int ORD_REC_BITS = 40;
constexpr int ALLELE_BITS = 5;

// Host-side string -> dense id table for keys that cannot be 2-bit coded (IUPAC UMIs,
// read names in UMI-less mode).  Sharded so decoder threads can intern concurrently.
class InternTable {
public:
    static constexpr int NSHARD = 64;
    uint64_t intern(const char* s, size_t n);
    // many strings at once: grouped by shard, so that a shard's lock is taken once and its slots stay in the cache
    struct Item { uint64_t h; const char* s; uint32_t n; uint32_t pad; uint64_t* out; int umi_bits; };
    void intern_batch(std::vector<Item>& items, bool* overflow);
    static uint64_t hash(const char* s, size_t n);
    uint64_t size() const;
    void clear();
private:
    // open addressing over (hash, bytes in a per-shard arena): no allocation per lookup (a std::unordered_map<std::string>
    // cost ~4 us per read name in UMI-less mode and made the well-based ingest 10x slower than the 10x one)
    struct Slot { uint64_t h; uint32_t off; uint32_t len1; uint32_t idx; uint32_t epoch; };   // live iff epoch == the shard's (clear() is O(1))
    struct Shard {
        std::mutex mu; std::vector<Slot> slots; std::vector<char> arena; uint32_t count = 0, epoch = 1;
        void grow();
    };
    Shard shards_[NSHARD];
    // id = (local index << 6) | shard  -> unique without a global counter
};

// Decoder settings derived from xck_config at xck_create()
struct DecodeCfg {
    bool use_barcodes = false;
    bool use_umi = false;
    char cell_tag[2] = {0, 0};
    char umi_tag[2] = {0, 0};
    int  umi_bits = 64;
    bool want_seq = false;
    bool verify_crc = false;
    int  n_threads = 0;
    int64_t max_batch_reads = 0;
    // barcode hash (open addressing over the barcode strings)
    struct BcSlot { uint64_t h; int32_t idx; uint16_t len; char key[18]; };   // 32 bytes: hash, column (-1 = empty), the barcode itself
    std::vector<std::string> barcodes;
    std::vector<BcSlot> bc_slots;        // size pow2
    uint64_t bc_mask = 0;
    int32_t lookup_cell(const char* s, size_t n) const;
    void build_barcodes(const char* const* names, int n);
};

uint64_t hash_bytes(const char* s, size_t n);
// 2-bit / interned key code, see include/xck.h xck_umi_bits()
uint64_t encode_key(const char* s, size_t n, int umi_bits, InternTable& tab, bool* overflow);

constexpr int XCK_F_LAYOUT_BOTH = 1 << 16;   // internal: size the row field for regions AND SNPs (fused handle)
// key layout rule shared by engine and decoder: 64-bit keys (row | cell | umi) when the UMI code
// gets >= 26 bits, otherwise 128-bit keys with a 64-bit UMI code.
struct KeyBits { int key_bits, ubits, cbits, rbits; };
inline int bits_for_count(int64_t n) { int b = 1; while ((int64_t(1) << b) < n) b++; return b; }
inline KeyBits key_layout(const xck_config* cfg) {
    KeyBits k;
    k.cbits = bits_for_count(cfg->n_cells > 2 ? cfg->n_cells : 2);
    // (rows are sized for one more than there are: the row field is then never all ones, so no key equals ~0 - the "empty" word of the
    // LDS hash sets of the folds - whatever the cell and UMI fields hold)
    k.rbits = bits_for_count((int64_t)(cfg->n_regions > 2 ? cfg->n_regions : 2) + 1);
    if ((cfg->mode & XCK_MODE_BAF) || (cfg->flags & XCK_F_LAYOUT_BOTH)) { int sb = bits_for_count((int64_t)(cfg->n_snps > 2 ? cfg->n_snps : 2) + 1); if (sb > k.rbits) k.rbits = sb; }
    int ub = 64 - k.rbits - k.cbits;
    if ((cfg->flags & XCK_F_FORCE_KEY128) || ub < 26) { k.key_bits = 128; k.ubits = 64; }
    else { k.key_bits = 64; k.ubits = ub; }
    return k;
}

// Tuning / test knobs of a handle, read from the environment ONCE, at xck_create (include/xck.h lists them with their meaning):
// the host path of push / finish never calls getenv().
struct Knobs {
    bool debug_timing = false;           // XCK_DEBUG_TIMING
    bool fold_sort = false;              // XCK_FOLD=sort
    int  fold_c = 0;                     // XCK_FOLD_C (0 = the built-in page size)
    int  fold_lgg = -1;                  // XCK_FOLD_LGG (-1 = by the number of cells)
    int  fold_copies_lg = 4;             // XCK_FOLD_COPIES_LG
    int  fold_bucket_blocks = 2048;      // XCK_FOLD_BUCKET_BLOCKS
    int  fold_overlap = 1;               // XCK_FOLD_OVERLAP
    int  fold_overlap_blocks = 512;      // XCK_FOLD_OVERLAP_BLOCKS
    bool full_sort = false;              // XCK_FULL_SORT
    bool pileup_radix = false;           // XCK_PILEUP_SORT=radix
    int  pileup_hap = 0;                 // XCK_PILEUP_HAP: 0 = packed class bits, 1 = sorted, 2 = values
    bool pileup_bitonic = false;         // XCK_PILEUP_ITEM_SORT=bitonic
    int  pileup_lgg = 10;                // XCK_PILEUP_LGG
    long long hit_slack = 65536;         // XCK_HIT_SLACK
    long long hit_cap0 = 1 << 20;        // XCK_HIT_CAP0
    int  push_stage = -1;                // XCK_PUSH_STAGE (-1 = by size)
    long long push_stage_bytes = 2 << 20;   // XCK_PUSH_STAGE_BYTES
    int  gpu_inflate_pct = -1;           // XCK_GPU_INFLATE: share (percent) of the BGZF chunks inflated on the handle's GPU; -1 = auto (keep gpu_inflate_depth chunks on the device)
    int  gpu_inflate_depth = 10;         // XCK_GPU_INFLATE_DEPTH
    int  gpu_inflate_ring = 12;          // XCK_GPU_INFLATE_RING: chunks in flight (host + device) while the GPU share is on
    int  gpu_inflate_min_mb = 96;        // XCK_GPU_INFLATE_MIN_MB: auto mode only for files (index ranges) of at least this many compressed MB
    int  gpu_inflate_free_cus = 32;      // XCK_GPU_INFLATE_FREE_CUS: CUs the inflate streams never use (they stay free for the join kernels)
    static Knobs from_env();             // api.cpp
};

// the decoder's own batches are valid by construction and skip the O(n) check of xck_push_batch()
int push_trusted(xck_engine* e, const xck_batch* b);
// host threads this process may really use: min(hardware threads, CPU affinity, cgroup CPU quota) - a container with a
// 16-CPU quota on a 256-thread host must not start 256 decoder threads
int  default_threads();
void set_thread_error(const std::string& s);
const char* get_thread_error();

}  // namespace xck

// The engine object behind the opaque C handle.
struct xck_engine {
    std::string err;
    xck::DecodeCfg dec;
    xck::InternTable intern;
    void* impl = nullptr;                // xck::EngineImpl being addressed (engine.hip); null for decode-only handles
    void* impls[2] = {nullptr, nullptr}; // fused handle (XCK_MODE_BOTH): [0] basefc pipeline, [1] pileup pipeline
    int n_impl = 0;
    void* stager = nullptr;              // xck::Stager (engine.hip): device staging slots of engine_push_block
    struct PushRing { void* blk[3] = {nullptr, nullptr, nullptr}; size_t cap[3] = {0, 0, 0}; void* fence[3] = {nullptr, nullptr, nullptr}; int next = 0; } push_ring;   // pinned blocks of xck_push_batch's one-copy form (api.cpp)
    xck::Knobs knobs;                    // the environment, read once at xck_create
    std::atomic<int64_t> gpu_inflate_chunks{0};   // chunks inflated on the device by the readers that fed this handle (xck_stats)
    int mode = 0;
    int umi_bits = 64;
    int32_t n_cells = 0, n_contigs = 0;  // bounds that caller-supplied batches are checked against (xck_push_batch)
    // host pinned batch staging used by xck_ingest_bam lives in the xck_bam
};

// implemented in engine.hip
namespace xck {
int  engine_create(const xck_config* cfg, xck_engine* e);
void engine_destroy(xck_engine* e);
int  engine_push(xck_engine* e, const xck_batch* b, bool device_resident);
int  engine_flush(xck_engine* e);
int  engine_finish(xck_engine* e, xck_result* out);
int  engine_finish_async(xck_engine* e);
int  engine_result_device(xck_engine* e, xck_result* out);
int  engine_reset(xck_engine* e);
int  engine_stats(const xck_engine* e, xck_stats* out);
int  engine_umi_bits(const xck_engine* e);
int  engine_device(const xck_engine* e);      // HIP device of the handle (-1: decode-only handle)
int  engine_numa_node(const xck_engine* e);   // NUMA node of the handle's GPU (sysfs, by PCI bus id); -1 = unknown
// One decoded chunk (all SoA columns in one pinned host block of `bytes` bytes at host_base; the batches point into it): ONE
// asynchronous H2D copy into a device staging slot shared by the handle's pipelines, then the join kernel(s) on the batches.
// *fence (created on first use) is recorded behind the copy: fence_wait() before the host block is overwritten.
int  engine_push_block(xck_engine* e, const void* host_base, size_t bytes, const xck_batch* batches, int n, void** fence);
void engine_release_staging(xck_engine* e);   // frees the staging slots (xck_destroy)
void fence_wait(void* fence);
void fence_destroy(void* fence);
void* pinned_alloc(size_t bytes);        // hipHostMalloc, falls back to malloc when no device
void  pinned_free(void* p);
}
